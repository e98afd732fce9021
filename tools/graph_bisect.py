"""Which kind of graph node replays wrong under the HIP runtime's AQL graph packet capture (ROCm 7.2 default)?  

Round 2 established (profiles/r2_graph_packet_capture.txt): a hipGraph recorded over tensors torch allocates INSIDE the capture
and libgsr calls replays right back to back and wrong once other GPU work ran between two replays -- unless
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0; a graph over pre-allocated buffers is right either way.  This script narrows it down: every
variant below is captured, replayed twice back to back, then once after unrelated eager GPU work (a fill, a sum with a
device-to-host read, a new allocation), and compared with its eager result.  Run it in BOTH environments:

    python tools/graph_bisect.py                                   # runtime default (packet capture on)
    DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 python tools/graph_bisect.py  # packet capture off

It must be started in a fresh process (the flag is read when the HIP runtime starts) and does not import mygauhuman_amd before
the environment is final.
"""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
FLAG = os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE")  # as given by the caller (the package import does not touch the environment)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mygauhuman_amd.diff_gaussian_rasterization import _C  # noqa: E402
from tests import util  # noqa: E402

dev = torch.device("cuda", 0)
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]


def disturb(k):
    junk = torch.empty(1 << 22, device=dev)
    junk.fill_(float(k))
    float(junk.sum())
    junk2 = torch.empty(1 << 24, device=dev)
    junk2.fill_(1.0)
    torch.cuda.synchronize()


def run_variant(name, make):
    """make() -> (step_fn, outputs_fn): step_fn runs the work; outputs_fn returns the list of tensors to compare."""
    step, outputs = make()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    ref = [t.detach().clone() for t in outputs()]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    out = outputs()   # tensors of the graph's pool (or the same pre-allocated ones)

    def rel():
        torch.cuda.synchronize()
        return [float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-30)) for a, b in zip(out, ref)]
    g.replay()
    g.replay()
    r0 = rel()
    disturb(2)
    g.replay()
    r1 = rel()
    disturb(3)
    g.replay()
    r2 = rel()
    res = dict(back_to_back=max(r0), after_eager_work=max(r1), again=max(r2), per_output_after=[float(f"{v:.2e}") for v in r1])
    print(json.dumps({name: res}), flush=True)
    del g
    return res


P, W, H = 4000, 160, 112
cam, g_np = util.make_scene(P, W, H, 3, 3, 0.04)
d = util.to_dev
T = {k: d(g_np[k]) for k in ("means3D", "opacities", "scales", "rotations", "shs")}
cm = {k: d(cam[k]) for k in ("viewmatrix", "projmatrix", "campos")}
bg = d(np.zeros(3, np.float32))
e = torch.empty(0)
rng = np.random.default_rng(0)
dc, dd, da = (d(rng.normal(0, 1, s).astype(np.float32)) for s in ((3, H, W), (1, H, W), (1, H, W)))
x0 = torch.rand(1 << 20, device=dev)


def v_torch_only():
    box = {}

    def step():
        y = x0 * 2.0
        z = torch.empty_like(y)
        z.copy_(y)
        w = torch.zeros_like(z)
        w += z
        box["o"] = [w, w.sum().reshape(1)]
    return step, lambda: box["o"]


def v_memset_node_on_pool():
    box = {}

    def step():
        t = torch.empty(1 << 20, device=dev)
        hip.hipMemsetAsync(t.data_ptr(), 0, t.numel() * 4, torch.cuda.current_stream().cuda_stream)
        box["o"] = [t + 1.0]
    return step, lambda: box["o"]


def v_memcpy_node_on_pool():
    box = {}

    def step():
        t = torch.empty(1 << 20, device=dev)
        hip.hipMemcpyAsync(t.data_ptr(), x0.data_ptr(), t.numel() * 4, 3, torch.cuda.current_stream().cuda_stream)  # D2D
        box["o"] = [t * 1.0]
    return step, lambda: box["o"]


def v_forward(prealloc):
    box = {}
    from mygauhuman_amd.fastpath import RasterSession
    sess = RasterSession(P, W, H, 16, dev, 1 << 20) if prealloc else None
    params = dict(means3D=T["means3D"], shs=T["shs"], opacities=T["opacities"], scales=T["scales"], rotations=T["rotations"])
    camd = dict(viewmatrix=cm["viewmatrix"], projmatrix=cm["projmatrix"], campos=cm["campos"], tanfovx=cam["tanfovx"],
                tanfovy=cam["tanfovy"], W=W, H=H)

    def step():
        if prealloc:
            c, dpt, a, r = sess.forward(params, camd, bg, 3)
            box["o"] = [c, a]
        else:
            o = _C.rasterize_gaussians_async(bg, T["means3D"], e, T["opacities"], T["scales"], T["rotations"], 1.0, e, cm["viewmatrix"],
                                             cm["projmatrix"], cam["tanfovx"], cam["tanfovy"], H, W, T["shs"], 3, cm["campos"], False, False)
            box["o"] = [o[1], o[3]]
            box["f"] = o
    return step, lambda: box["o"]


def v_forward_backward():
    box = {}

    def step():
        o = _C.rasterize_gaussians_async(bg, T["means3D"], e, T["opacities"], T["scales"], T["rotations"], 1.0, e, cm["viewmatrix"],
                                         cm["projmatrix"], cam["tanfovx"], cam["tanfovy"], H, W, T["shs"], 3, cm["campos"], False, False)
        gr = _C.rasterize_gaussians_backward(bg, T["means3D"], o[4], e, T["scales"], T["rotations"], 1.0, e, cm["viewmatrix"],
                                             cm["projmatrix"], cam["tanfovx"], cam["tanfovy"], dc, dd, da, T["shs"], 3, cm["campos"],
                                             o[5], o[0], o[6], o[7], o[3], False)
        box["o"] = [o[1], o[3]] + [t for t in gr if t is not None]
    return step, lambda: box["o"]


def main():
    print(json.dumps(dict(env_flag=FLAG, torch=torch.__version__, hip=torch.version.hip)), flush=True)
    out = {}
    out["torch_only_pool_allocs"] = run_variant("torch_only_pool_allocs", v_torch_only)
    out["memset_node_on_pool_tensor"] = run_variant("memset_node_on_pool_tensor", v_memset_node_on_pool)
    out["memcpy_node_on_pool_tensor"] = run_variant("memcpy_node_on_pool_tensor", v_memcpy_node_on_pool)
    out["libgsr_forward_preallocated"] = run_variant("libgsr_forward_preallocated", lambda: v_forward(True))
    out["libgsr_forward_pool_allocs"] = run_variant("libgsr_forward_pool_allocs", lambda: v_forward(False))
    out["libgsr_forward_backward_pool_allocs"] = run_variant("libgsr_forward_backward_pool_allocs", v_forward_backward)
    _C.AsyncCapacity.graph_status.clear()
    print(json.dumps(dict(env_flag=FLAG, summary={k: v["after_eager_work"] for k, v in out.items()})), flush=True)


if __name__ == "__main__":
    main()
