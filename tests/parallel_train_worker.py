"""Rank process of the view-parallel TRAINING-LOOP test (tests/test_gpu_parallel_render.py): a few iterations of what train.py:212-417
does, on `world` views per step -- ViewParallelRender step, densification statistics, Adam step of the nine parameter groups and of
the two decoders, one densify-and-prune in the middle (same torch seed on every rank: scene/gaussian_model.py:526-530 samples) --
and stores the final model so that the test can check that the replicas are still bit-identical.

  python -m tests.parallel_train_worker <out_prefix> <P> <V> <W> <H> <iters> <densify_at>
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    out, P, V, W, H, iters, densify_at = sys.argv[1], *[int(v) for v in sys.argv[2:8]]
    from mygauhuman_amd import densify, human_synth, parallel
    from tests.parallel_render_worker import image_weights, loss_of, pipe
    rank, world, local = parallel.init_distributed("cuda")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.manual_seed(0)
    model, body = human_synth.build(P, V, dev, seed=0, motion=True)
    densify.training_setup(model, dict(xyz=1.6e-4, f_dc=2.5e-3, f_rest=1.25e-4, opacity=0.05, scaling=5e-3, rotation=1e-3, normal=1e-3,
                                       albedo=0.02, roughness=0.02))
    dec_params = list(model.pose_decoder.parameters()) + list(model.lweight_offset_decoder.parameters())
    dec_opt = torch.optim.Adam(dec_params, lr=1e-4)
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    verts = torch.from_numpy(body["v_template"]).to(dev)
    step = parallel.ViewParallelRender(model, pipe(), bg)
    counts, losses = [], []
    for it in range(1, iters + 1):
        view = parallel.view_for_step(it - 1, rank, world) % 8
        cam = human_synth.view_camera(body, W, H, view, n_views=8, device=dev)
        weights = image_weights(W, H, view, dev)
        _, loss = step(it, cam, lambda o: loss_of(o, weights))
        with torch.no_grad():
            step.accumulate_densification_stats()
        model.optimizer.step()
        dec_opt.step()
        losses.append(float(loss.detach()))
        if it == densify_at:
            step.check()
            with torch.no_grad():
                densify.densify_and_prune(model, 2e-6, 0.005, 2.0, 20, t_vertices=verts)
            step = parallel.ViewParallelRender(model, pipe(), bg)   # the Gaussian count changed: new bucket, same protocol
        counts.append(int(model.get_xyz.shape[0]))
    step.check()
    res = {n: getattr(model, n).detach().cpu().numpy() for n in parallel.ViewParallelRender.MODEL_LEAVES}
    for k, p_ in enumerate(dec_params):
        res[f"dec{k}"] = p_.detach().cpu().numpy()
    res["xyz_gradient_accum"] = model.xyz_gradient_accum.cpu().numpy()
    res["denom"] = model.denom.cpu().numpy()
    res["max_radii2D"] = model.max_radii2D.cpu().numpy()
    res["counts"] = np.array(counts)
    res["losses"] = np.array(losses)
    np.savez(f"{out}_rank{rank}.npz", **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
