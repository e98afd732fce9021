"""Densify / prune / optimizer-state surgery of the Gaussian set -- the step either side of the
render hot path, with the reference's method names and semantics (scene/gaussian_model.py):

    training_setup :250-287 (Adam groups by name)      add_densification_stats :764-766    reset_opacity :348-351
    prune_points :443-461                              densify_and_clone :544-565          densify_and_split :514-542
    densify_and_prune :697-726 (clone, split, then prune by opacity / screen size / world size / distance to the SMPL surface)
    kl_div :740-762, kl_densify_and_clone :566-606, kl_densify_and_split :608-666, kl_merge :668-708 (pairs from the k = 2 self k-NN)

What changes underneath: the reference rebuilds 9 parameters x (value, exp_avg, exp_avg_sq) + 3 statistics with boolean-mask
indexing and torch.cat -- about 90 kernels and 30 host synchronisations per operation.  Here every operation is ONE row plan
(an int32 source-row index; bit 30 marks rows of newly created Gaussians whose Adam moments start at zero) applied to all
30 arrays by ONE HIP kernel (gsr_gather_rows, csrc/rows.hip); the nearest-SMPL-vertex distance of the prune test comes from
the grid k-NN (gsr_knn_nearest).  The plan builders are plain tensor code (also runs on CPU, tests/test_densify_cpu.py);
applying a plan needs the GPU library.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import covariance
from ._lib import check, lib

GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "normal", "albedo", "roughness")
ATTR = dict(xyz="_xyz", f_dc="_features_dc", f_rest="_features_rest", opacity="_opacity", scaling="_scaling",
            rotation="_rotation", normal="_normal", albedo="_albedo", roughness="_roughness")
NEW_ROW = 1 << 30


# ---------------------------------------------------------------- row plans (device-agnostic)
def plan_prune(mask):
    """prune_points(mask): keep the rows where mask is False, in order."""
    return torch.nonzero(~mask, as_tuple=False).squeeze(1).to(torch.int32)


def plan_clone(selected):
    """densify_and_clone: all rows, then a copy of every selected row (new)."""
    P = selected.shape[0]
    src = torch.nonzero(selected, as_tuple=False).squeeze(1).to(torch.int32)
    return torch.cat([torch.arange(P, dtype=torch.int32, device=selected.device), src | NEW_ROW])


def plan_split(selected, N=2):
    """densify_and_split: the rows that are not split, then N blocks of copies of the split rows (new).
    Returns (plan, source rows of the children [N * n_sel], first child row in the output)."""
    keep = torch.nonzero(~selected, as_tuple=False).squeeze(1).to(torch.int32)
    src = torch.nonzero(selected, as_tuple=False).squeeze(1).to(torch.int32)
    children = src.repeat(N)
    return torch.cat([keep, children | NEW_ROW]), children, int(keep.shape[0])


def clone_mask(grads, scaling, grad_threshold, scene_extent, percent_dense):
    sel = torch.norm(grads, dim=-1) >= grad_threshold
    return torch.logical_and(sel, torch.max(scaling, dim=1).values <= percent_dense * scene_extent)


def split_mask(grads, n_points, scaling, grad_threshold, scene_extent, percent_dense):
    padded = torch.zeros((n_points,), device=grads.device, dtype=grads.dtype)
    padded[:grads.shape[0]] = grads.squeeze()
    sel = padded >= grad_threshold
    return torch.logical_and(sel, torch.max(scaling, dim=1).values > percent_dense * scene_extent)


# ---------------------------------------------------------------- applying a plan to model + optimizer
def _arrays(model):
    """[(kind, name, tensor [P, w], zero_new)] of everything that has one row per Gaussian."""
    out = []
    for g in GROUPS:
        p = getattr(model, ATTR[g])
        out.append(("param", g, p, 0))
        st = model.optimizer.state.get(p, None) if model.optimizer is not None else None
        if st is not None and "exp_avg" in st:
            out.append(("exp_avg", g, st["exp_avg"], 1))
            out.append(("exp_avg_sq", g, st["exp_avg_sq"], 1))
    for s in ("xyz_gradient_accum", "denom", "max_radii2D"):
        out.append(("stat", s, getattr(model, s), 1))
    return out


def apply_plan(model, plan, reset_stats):
    """Move every per-Gaussian array of `model` (parameters, Adam moments, statistics) to the row set `plan` describes and
    re-register the parameters with the optimizer the way _prune_optimizer / cat_tensors_to_optimizer do."""
    dev = model._xyz.device
    if dev.type != "cuda":
        raise RuntimeError("densify: applying a row plan needs the HIP library (tensors must live on the GPU)")
    arrays = _arrays(model)
    n_out = int(plan.shape[0])
    srcs, dsts, widths, zero_new, outs = [], [], [], [], []
    for kind, name, t, zn in arrays:
        t2 = t.detach().contiguous().float()
        w = 1
        for s in t2.shape[1:]:
            w *= int(s)
        o = torch.empty((n_out,) + tuple(t2.shape[1:]), dtype=torch.float32, device=dev)
        srcs.append(t2)
        outs.append(o)
        widths.append(w)
        zero_new.append(1 if (zn and not (kind == "stat" and not reset_stats)) else 0)
    if reset_stats:  # densification_postfix: statistics restart from zero for the whole new set
        for i, (kind, _, _, _) in enumerate(arrays):
            if kind == "stat":
                outs[i].zero_()
    # zero-width arrays (f_rest at SH degree 0) have nothing to move
    move = [i for i, (kind, _, _, _) in enumerate(arrays) if not (kind == "stat" and reset_stats) and widths[i] > 0]
    if n_out and move:
        n = len(move)
        plan_c = plan.to(device=dev, dtype=torch.int32).contiguous()
        src_p = (C.c_void_p * n)(*[srcs[i].data_ptr() for i in move])
        dst_p = (C.c_void_p * n)(*[outs[i].data_ptr() for i in move])
        w_p = (C.c_int * n)(*[widths[i] for i in move])
        z_p = (C.c_int * n)(*[zero_new[i] for i in move])
        with torch.cuda.device(dev):
            check(lib.gsr_gather_rows(n, src_p, dst_p, w_p, z_p, n_out, plan_c.data_ptr(),
                                      torch.cuda.current_stream(dev).cuda_stream), "gsr_gather_rows")
    # re-register (scene/gaussian_model.py:421-441,463-486)
    by = {(kind, name): o for (kind, name, _, _), o in zip(arrays, outs)}
    for g in GROUPS:
        old = getattr(model, ATTR[g])
        new = nn.Parameter(by[("param", g)].requires_grad_(old.requires_grad))
        if model.optimizer is not None:
            for group in model.optimizer.param_groups:
                if group.get("name") == g:
                    st = model.optimizer.state.pop(old, None)
                    group["params"][0] = new
                    if st is not None:
                        if ("exp_avg", g) in by:
                            st["exp_avg"], st["exp_avg_sq"] = by[("exp_avg", g)], by[("exp_avg_sq", g)]
                        model.optimizer.state[new] = st
        setattr(model, ATTR[g], new)
    model.xyz_gradient_accum = by[("stat", "xyz_gradient_accum")]
    model.denom = by[("stat", "denom")]
    model.max_radii2D = by[("stat", "max_radii2D")]


# ---------------------------------------------------------------- the reference's operations
def training_setup(model, lrs, percent_dense=0.01):
    """lrs: dict group name -> learning rate (position_lr_init * spatial_lr_scale etc., :256-281).  Adam(eps=1e-15)."""
    P, dev = model._xyz.shape[0], model._xyz.device
    model.percent_dense = percent_dense
    model.xyz_gradient_accum = torch.zeros((P, 1), device=dev)
    model.denom = torch.zeros((P, 1), device=dev)
    model.max_radii2D = torch.zeros((P,), device=dev)
    groups = [{"params": [getattr(model, ATTR[g])], "lr": float(lrs.get(g, 0.0)), "name": g} for g in GROUPS]
    # same update rule as the reference's torch.optim.Adam(l, lr=0.0, eps=1e-15) (:283); on the GPU the fused implementation
    # runs one kernel per group instead of the foreach path's ~8
    model.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15, fused=True if dev.type == "cuda" else None)
    return model.optimizer


def add_densification_stats(model, viewspace_point_tensor, update_filter):
    """scene/gaussian_model.py:764-766.  Same values, written without boolean-mask indexing (which costs a nonzero kernel and
    a host synchronisation per statement): rows outside the filter get + 0."""
    g = viewspace_point_tensor.grad
    if g is None:  # the reference fails here as well (NoneType has no [:, :2]): retain_grad() was lost or backward() did not run
        raise RuntimeError("add_densification_stats: viewspace_point_tensor.grad is None (call backward() first and keep "
                           "retain_grad() on the screen-space points)")
    f = update_filter.to(model.denom.dtype).unsqueeze(-1)
    model.xyz_gradient_accum += torch.norm(g[:, :2], dim=-1, keepdim=True) * f
    model.denom += f


def update_max_radii(model, radii, visibility_filter):
    """max_radii2D[vis] = max(max_radii2D[vis], radii[vis]) (train.py:403) without the boolean-mask gather / scatter."""
    r = radii.to(model.max_radii2D.dtype)
    model.max_radii2D = torch.where(visibility_filter, torch.maximum(model.max_radii2D, r), model.max_radii2D)


def prune_points(model, mask):
    apply_plan(model, plan_prune(mask), reset_stats=False)


def densify_and_clone(model, grads, grad_threshold, scene_extent):
    sel = clone_mask(grads, model.get_scaling, grad_threshold, scene_extent, model.percent_dense)
    apply_plan(model, plan_clone(sel), reset_stats=True)
    return sel


def _split_selected(model, sel, N, unit_samples):
    """The body shared by densify_and_split (:527-542) and kl_densify_and_split (:649-666): every selected Gaussian is replaced
    by N children drawn from it, 1 / (0.8 N) of its size."""
    scal = model.get_scaling.detach()
    plan, children, first = plan_split(sel, N)
    n_child = int(children.shape[0])
    if n_child:
        ch = children.long()
        stds = scal[ch]
        unit = torch.randn((n_child, 3), device=stds.device) if unit_samples is None else unit_samples[:n_child].to(stds.device)
        samples = stds * unit                                     # torch.normal(mean=0, std=stds)
        rots = covariance.build_rotation(model._rotation.detach()[ch])
        new_xyz = covariance.bmm3(rots, samples.unsqueeze(-1)).squeeze(-1) + model._xyz.detach()[ch]
        new_scaling = torch.log(stds / (0.8 * N))
    apply_plan(model, plan, reset_stats=True)
    if n_child:
        with torch.no_grad():
            model._xyz[first:] = new_xyz
            model._scaling[first:] = new_scaling


def densify_and_split(model, grads, grad_threshold, scene_extent, N=2, unit_samples=None):
    """unit_samples: optional N(0,1) draws [N * n_selected, 3] (tests); default torch.randn like the reference's torch.normal."""
    sel = split_mask(grads, model._xyz.shape[0], model.get_scaling.detach(), grad_threshold, scene_extent, model.percent_dense)
    _split_selected(model, sel, N, unit_samples)
    return sel


# ---------------------------------------------------------------- KL-divergence variants (scene/gaussian_model.py:566-708,740-762)
# Not called by the reference's own training loop (its densify_and_prune has them commented out, :702-704), kept here with the
# reference's names and thresholds for callers that enable them.  The pair of every Gaussian is (itself, its nearest other
# Gaussian) from the self k-NN with k = 2 (gsr_knn_self instead of KNN_CUDA).
def kl_div(mu_0, rotation_0_q, scaling_0_diag, mu_1, rotation_1_q, scaling_1_diag):
    """KL( N(mu_0, S_0) || N(mu_1, S_1) ) = 1/2 [ tr(S_1^-1 S_0) + d^T S_1^-1 d + ln(det S_1 / det S_0) - 3 ],  S = R diag(s^2) R^T."""
    R0, R1 = covariance.build_rotation(rotation_0_q), covariance.build_rotation(rotation_1_q)
    cov_0 = covariance.bmm3(R0 * (scaling_0_diag * scaling_0_diag).unsqueeze(1), R0.transpose(1, 2))
    inv_1 = covariance.bmm3(R1 * (1.0 / (scaling_1_diag * scaling_1_diag)).unsqueeze(1), R1.transpose(1, 2))
    d = mu_1 - mu_0
    trace = (inv_1 * cov_0).sum(dim=(1, 2))                       # both symmetric: tr(A B) = sum_ij A_ij B_ij
    maha = (covariance.bmm3(inv_1, d.unsqueeze(2)).squeeze(2) * d).sum(1)
    logdet = torch.log(torch.prod((scaling_1_diag / scaling_0_diag) ** 2, dim=1))
    return 0.5 * (trace + maha + logdet - 3.0)


def kl_to_nearest(model):
    """(kl [P], ids [P, 2] int64): divergence between every Gaussian's k-NN pair (:573-589)."""
    from .knn_cuda import knn_self
    xyz = model._xyz.detach()
    _, ids = knn_self(xyz, 2)
    ids = ids.long()
    rot, scal = model._rotation.detach(), model.get_scaling.detach()
    i0, i1 = ids[:, 0], ids[:, 1]
    return kl_div(xyz[i0], rot[i0], scal[i0], xyz[i1], rot[i1], scal[i1]), ids


def kl_densify_and_clone(model, grads, grad_threshold, scene_extent, kl_threshold=0.4, unit_samples=None):
    """:566-606.  Unlike densify_and_clone the copy is displaced by a draw from the Gaussian itself."""
    sel = clone_mask(grads, model.get_scaling.detach(), grad_threshold, scene_extent, model.percent_dense)
    kl, _ = kl_to_nearest(model)
    model.kl_selected_pts_mask = kl > kl_threshold
    sel = sel & model.kl_selected_pts_mask
    src = torch.nonzero(sel, as_tuple=False).squeeze(1)
    n_new, first = int(src.shape[0]), model._xyz.shape[0]
    if n_new:
        stds = model.get_scaling.detach()[src]
        unit = torch.randn((n_new, 3), device=stds.device) if unit_samples is None else unit_samples[:n_new].to(stds.device)
        rots = covariance.build_rotation(model._rotation.detach()[src])
        new_xyz = covariance.bmm3(rots, (stds * unit).unsqueeze(-1)).squeeze(-1) + model._xyz.detach()[src]
        new_scaling = torch.log(stds)                             # scaling_inverse_activation(get_scaling)
    apply_plan(model, plan_clone(sel), reset_stats=True)
    if n_new:
        with torch.no_grad():
            model._xyz[first:] = new_xyz
            model._scaling[first:] = new_scaling
    return sel


def kl_densify_and_split(model, grads, grad_threshold, scene_extent, kl_threshold=0.4, N=2, unit_samples=None):
    """:608-666."""
    sel = split_mask(grads, model._xyz.shape[0], model.get_scaling.detach(), grad_threshold, scene_extent, model.percent_dense)
    kl, _ = kl_to_nearest(model)
    model.kl_selected_pts_mask = kl > kl_threshold
    sel = sel & model.kl_selected_pts_mask
    _split_selected(model, sel, N, unit_samples)
    return sel


def kl_merge(model, grads, grad_threshold, scene_extent, kl_threshold=0.1):
    """:668-708: a selected small Gaussian whose pair is closer than kl_threshold is replaced, together with its partner, by one
    Gaussian at their mean (position, SH, opacity averaged; rotation of the first, its scale / 0.8).

    The reference's body cannot run as written: it averages `_normal[selected_pts_mask]` over the wrong axis and calls
    densification_postfix with 7 of its 9 arguments (:694-698, a TypeError).  normal / albedo / roughness follow the rule of the
    other averaged attributes here (mean over the pair)."""
    P = model._xyz.shape[0]
    padded = torch.zeros((P,), device=grads.device, dtype=grads.dtype)
    padded[:grads.shape[0]] = grads.squeeze()
    sel = torch.logical_and(padded >= grad_threshold,
                            torch.max(model.get_scaling.detach(), dim=1).values <= model.percent_dense * scene_extent)
    kl, ids = kl_to_nearest(model)
    model.kl_selected_pts_mask = kl < kl_threshold
    sel = sel & model.kl_selected_pts_mask
    pair = ids[sel]                                               # [n, 2]
    n_new = int(pair.shape[0])
    if n_new == 0:
        return sel
    with torch.no_grad():
        mean = {g: getattr(model, ATTR[g]).detach()[pair].mean(1) for g in GROUPS if g not in ("scaling", "rotation")}
        new_scaling = torch.log(model.get_scaling.detach()[pair][:, 0] / 0.8)
    gone = sel.clone()
    gone[pair[:, 1]] = True
    keep = torch.nonzero(~gone, as_tuple=False).squeeze(1).to(torch.int32)
    first = int(keep.shape[0])
    apply_plan(model, torch.cat([keep, pair[:, 0].to(torch.int32) | NEW_ROW]), reset_stats=True)   # rotation: copied from the first
    with torch.no_grad():
        for g, v in mean.items():
            getattr(model, ATTR[g])[first:] = v
        model._scaling[first:] = new_scaling
    return sel


def reset_opacity(model):
    op = torch.min(model.get_opacity, torch.ones_like(model.get_opacity) * 0.01).detach()
    new = nn.Parameter(torch.log(op / (1 - op)).requires_grad_(True))
    for group in model.optimizer.param_groups:
        if group.get("name") == "opacity":
            st = model.optimizer.state.pop(group["params"][0], None)
            if st is not None:
                st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(new), torch.zeros_like(new)
            group["params"][0] = new
            if st is not None:
                model.optimizer.state[new] = st
    model._opacity = new


def densify_and_prune(model, max_grad, min_opacity, extent, max_screen_size, t_vertices=None, unit_samples=None):
    """scene/gaussian_model.py:697-726 (the KL variants are commented out in the reference)."""
    from .knn_cuda import knn_nearest
    grads = model.xyz_gradient_accum / model.denom
    grads[grads.isnan()] = 0.0
    densify_and_clone(model, grads, max_grad, extent)
    densify_and_split(model, grads, max_grad, extent, unit_samples=unit_samples)
    prune_mask = (model.get_opacity < min_opacity).squeeze()
    if max_screen_size:
        big_points_vs = model.max_radii2D > max_screen_size
        big_points_ws = model.get_scaling.max(dim=1).values > 0.1 * extent
        prune_mask = torch.logical_or(torch.logical_or(prune_mask, big_points_vs), big_points_ws)
    if t_vertices is not None:  # use the SMPL prior to prune points (:715-720)
        distance, _ = knn_nearest(t_vertices.reshape(-1, 3), model._xyz.detach())
        prune_mask = prune_mask | (distance > 0.05)
    prune_points(model, prune_mask)
    return prune_mask
