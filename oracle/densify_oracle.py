"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's densify / prune / optimizer-state surgery
(scene/gaussian_model.py: _prune_optimizer :421-441, prune_points :443-461, cat_tensors_to_optimizer :463-486,
densification_postfix :488-512, densify_and_split :514-542, densify_and_clone :544-565, densify_and_prune :697-726,
add_densification_stats :764-766, reset_opacity :348-351, kl_div :740-762, kl_densify_and_clone :566-606, kl_densify_and_split
:608-666, kl_merge :668-708), statement by statement, on a dict-of-arrays state:

    state = {"params": {name: [P, ...]}, "exp_avg": {name: ...}, "exp_avg_sq": {name: ...},
             "xyz_gradient_accum": [P,1], "denom": [P,1], "max_radii2D": [P]}

Parity unpinned against the reference itself (scene.gaussian_model is not importable here: knn_cuda, simple_knn, cv2 ...);
restated from the text.  The random draw of densify_and_split (torch.normal) is an INPUT (`unit_samples`, N(0,1); the
reference's sample is std * unit) so that both sides see the same numbers."""
import numpy as np

GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "normal", "albedo", "roughness")


def build_rotation(r):  # utils/general_utils.py:78-100
    q = r / np.sqrt((r * r).sum(1, dtype=np.float32))[:, None]
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.zeros((r.shape[0], 3, 3), np.float32)
    R[:, 0, 0] = 1 - 2 * (y * y + z * z)
    R[:, 0, 1] = 2 * (x * y - w * z)
    R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z)
    R[:, 1, 1] = 1 - 2 * (x * x + z * z)
    R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y)
    R[:, 2, 1] = 2 * (y * z + w * x)
    R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def _prune(state, valid):
    for g in GROUPS:
        state["params"][g] = state["params"][g][valid]
        if g in state["exp_avg"]:
            state["exp_avg"][g] = state["exp_avg"][g][valid]
            state["exp_avg_sq"][g] = state["exp_avg_sq"][g][valid]
    state["xyz_gradient_accum"] = state["xyz_gradient_accum"][valid]
    state["denom"] = state["denom"][valid]
    state["max_radii2D"] = state["max_radii2D"][valid]


def prune_points(state, mask):
    _prune(state, ~mask)


def _postfix(state, new):
    for g in GROUPS:
        ext = new[g]
        state["params"][g] = np.concatenate([state["params"][g], ext], 0)
        if g in state["exp_avg"]:
            state["exp_avg"][g] = np.concatenate([state["exp_avg"][g], np.zeros_like(ext)], 0)
            state["exp_avg_sq"][g] = np.concatenate([state["exp_avg_sq"][g], np.zeros_like(ext)], 0)
    P = state["params"]["xyz"].shape[0]
    state["xyz_gradient_accum"] = np.zeros((P, 1), np.float32)
    state["denom"] = np.zeros((P, 1), np.float32)
    state["max_radii2D"] = np.zeros((P,), np.float32)


def densify_and_clone(state, grads, grad_threshold, scene_extent, percent_dense):
    p = state["params"]
    sel = np.linalg.norm(grads, axis=-1) >= grad_threshold
    sel &= np.exp(p["scaling"]).max(1) <= percent_dense * scene_extent
    _postfix(state, {g: p[g][sel] for g in GROUPS})
    return sel


def densify_and_split(state, grads, grad_threshold, scene_extent, percent_dense, unit_samples, N=2):
    p = state["params"]
    n_init = p["xyz"].shape[0]
    padded = np.zeros(n_init, np.float32)
    padded[:grads.shape[0]] = grads.squeeze(-1)
    scal = np.exp(p["scaling"])
    sel = (padded >= grad_threshold) & (scal.max(1) > percent_dense * scene_extent)
    stds = np.tile(scal[sel], (N, 1))
    samples = (stds * unit_samples[:stds.shape[0]]).astype(np.float32)
    rots = np.tile(build_rotation(p["rotation"][sel]), (N, 1, 1))
    new = {g: np.tile(p[g][sel], (N,) + (1,) * (p[g].ndim - 1)) for g in GROUPS}
    new["xyz"] = (np.einsum("nij,nj->ni", rots, samples) + np.tile(p["xyz"][sel], (N, 1))).astype(np.float32)
    new["scaling"] = np.log(np.tile(scal[sel], (N, 1)) / np.float32(0.8 * N)).astype(np.float32)
    _postfix(state, new)
    prune_filter = np.concatenate([sel, np.zeros(N * int(sel.sum()), bool)])
    prune_points(state, prune_filter)
    return sel


def knn_self_2(xyz):
    """What KNN(k=2)(xyz, xyz) returns per point: ids of the two nearest points of the set itself (the point first), brute force,
    ties to the lower index."""
    x = xyz.astype(np.float64)
    d2 = ((x[:, None, :] - x[None, :, :]) ** 2).sum(-1)
    return np.argsort(d2, axis=1, kind="stable")[:, :2]


def kl_div(mu_0, q_0, s_0, mu_1, q_1, s_1):  # :740-762
    R0, R1 = build_rotation(q_0).astype(np.float64), build_rotation(q_1).astype(np.float64)
    s_0, s_1 = s_0.astype(np.float64), s_1.astype(np.float64)
    L0 = R0 * s_0[:, None, :]
    cov_0 = L0 @ L0.transpose(0, 2, 1)
    L1i = R1 * (1.0 / s_1)[:, None, :]
    cov_1_inv = L1i @ L1i.transpose(0, 2, 1)
    d = (mu_1 - mu_0).astype(np.float64)
    k0 = np.trace(cov_1_inv @ cov_0, axis1=1, axis2=2)
    k1 = np.einsum("ni,nij,nj->n", d, cov_1_inv, d)
    k2 = np.log(np.prod((s_1 / s_0) ** 2, axis=1))
    return 0.5 * (k0 + k1 + k2 - 3)


def kl_pairs(state):
    p = state["params"]
    ids = knn_self_2(p["xyz"])
    scal = np.exp(p["scaling"])
    i0, i1 = ids[:, 0], ids[:, 1]
    return kl_div(p["xyz"][i0], p["rotation"][i0], scal[i0], p["xyz"][i1], p["rotation"][i1], scal[i1]), ids


def kl_densify_and_clone(state, grads, grad_threshold, scene_extent, percent_dense, unit_samples, kl_threshold=0.4):
    p = state["params"]
    scal = np.exp(p["scaling"])
    sel = (np.linalg.norm(grads, axis=-1) >= grad_threshold) & (scal.max(1) <= percent_dense * scene_extent)
    kl, _ = kl_pairs(state)
    sel = sel & (kl > kl_threshold)
    stds = scal[sel]
    samples = (stds * unit_samples[:stds.shape[0]]).astype(np.float32)
    new = {g: p[g][sel] for g in GROUPS}
    new["xyz"] = (np.einsum("nij,nj->ni", build_rotation(p["rotation"][sel]), samples) + p["xyz"][sel]).astype(np.float32)
    new["scaling"] = np.log(stds).astype(np.float32)
    _postfix(state, new)
    return sel, kl


def kl_densify_and_split(state, grads, grad_threshold, scene_extent, percent_dense, unit_samples, kl_threshold=0.4, N=2):
    p = state["params"]
    n_init = p["xyz"].shape[0]
    padded = np.zeros(n_init, np.float32)
    padded[:grads.shape[0]] = grads.squeeze(-1)
    scal = np.exp(p["scaling"])
    sel = (padded >= grad_threshold) & (scal.max(1) > percent_dense * scene_extent)
    kl, _ = kl_pairs(state)
    sel = sel & (kl > kl_threshold)
    stds = np.tile(scal[sel], (N, 1))
    samples = (stds * unit_samples[:stds.shape[0]]).astype(np.float32)
    rots = np.tile(build_rotation(p["rotation"][sel]), (N, 1, 1))
    new = {g: np.tile(p[g][sel], (N,) + (1,) * (p[g].ndim - 1)) for g in GROUPS}
    new["xyz"] = (np.einsum("nij,nj->ni", rots, samples) + np.tile(p["xyz"][sel], (N, 1))).astype(np.float32)
    new["scaling"] = np.log(np.tile(scal[sel], (N, 1)) / np.float32(0.8 * N)).astype(np.float32)
    _postfix(state, new)
    prune_points(state, np.concatenate([sel, np.zeros(N * int(sel.sum()), bool)]))
    return sel, kl


def kl_merge(state, grads, grad_threshold, scene_extent, percent_dense, kl_threshold=0.1):
    """normal / albedo / roughness: mean over the pair (the reference's statements for them cannot execute, :694-698)."""
    p = state["params"]
    n_init = p["xyz"].shape[0]
    padded = np.zeros(n_init, np.float32)
    padded[:grads.shape[0]] = grads.squeeze(-1)
    scal = np.exp(p["scaling"])
    sel = (padded >= grad_threshold) & (scal.max(1) <= percent_dense * scene_extent)
    kl, ids = kl_pairs(state)
    sel = sel & (kl < kl_threshold)
    if sel.sum() >= 1:
        pair = ids[sel]
        new = {g: p[g][pair].mean(1, dtype=np.float32) for g in GROUPS}
        new["scaling"] = np.log(scal[pair][:, 0] / np.float32(0.8)).astype(np.float32)
        new["rotation"] = p["rotation"][pair][:, 0]
        _postfix(state, new)
        gone = sel.copy()
        gone[pair[:, 1]] = True
        prune_points(state, np.concatenate([gone, np.zeros(pair.shape[0], bool)]))
    return sel, kl


def densify_and_prune(state, max_grad, min_opacity, extent, max_screen_size, percent_dense, unit_samples, dist_to_smpl_fn):
    """dist_to_smpl_fn(xyz) -> Euclidean distance of every point to its nearest SMPL vertex (the knn of :716)."""
    with np.errstate(invalid="ignore", divide="ignore"):
        grads = state["xyz_gradient_accum"] / state["denom"]
    grads[np.isnan(grads)] = 0.0
    densify_and_clone(state, grads, max_grad, extent, percent_dense)
    densify_and_split(state, grads, max_grad, extent, percent_dense, unit_samples)
    p = state["params"]
    prune = (1.0 / (1.0 + np.exp(-p["opacity"])) < min_opacity).squeeze(-1)
    if max_screen_size:
        big_vs = state["max_radii2D"] > max_screen_size
        big_ws = np.exp(p["scaling"]).max(1) > 0.1 * extent
        prune = prune | big_vs | big_ws
    prune = prune | (dist_to_smpl_fn(p["xyz"]) > 0.05)
    prune_points(state, prune)
    return state


def add_densification_stats(state, viewspace_grad, update_filter):
    state["xyz_gradient_accum"][update_filter] += np.linalg.norm(viewspace_grad[update_filter, :2], axis=-1, keepdims=True)
    state["denom"][update_filter] += 1


def reset_opacity(state):
    op = 1.0 / (1.0 + np.exp(-state["params"]["opacity"]))
    new = np.minimum(op, 0.01).astype(np.float32)
    state["params"]["opacity"] = np.log(new / (1 - new)).astype(np.float32)
    if "opacity" in state["exp_avg"]:
        state["exp_avg"]["opacity"] = np.zeros_like(new)
        state["exp_avg_sq"]["opacity"] = np.zeros_like(new)
