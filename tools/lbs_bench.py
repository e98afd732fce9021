import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from mygauhuman_amd import lbs
rng = np.random.default_rng(0)
P, V = 200000, 6890
d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
verts = d(rng.uniform(-1, 1, (V, 3)) * [0.45, 0.9, 0.15])
q = (verts[torch.from_numpy(rng.integers(0, V, P)).cuda()] + 0.01 * torch.randn(P, 3, device='cuda')).requires_grad_(True)
n = torch.randn(P, 3, device='cuda', requires_grad=True)
A_big = torch.eye(4, device='cuda').repeat(24, 1, 1) + 0.01 * torch.randn(24, 4, 4, device='cuda'); A_big[:, 3] = torch.tensor([0., 0, 0, 1], device='cuda')
w = torch.rand(V, 24, device='cuda') ** 4; w = w / w.sum(1, keepdim=True)
z = torch.zeros(V, 3, device='cuda')
for need in (False, True):
    A_pose = (A_big + 0.01).clone().requires_grad_(need)
    off_pose = z.clone().requires_grad_(need)
    def step():
        for t in (q, n, A_pose, off_pose):
            t.grad = None
        o = lbs.lbs_deform(q, n, None, A_big, A_pose, z, z, off_pose, torch.eye(3, device='cuda'), torch.zeros(3, device='cuda'), verts, w, lean=True)
        (o['world_pts'].sum() + o['transforms'].sum() + o['world_normals'].sum()).backward()
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): step()
    torch.cuda.synchronize()
    print(f"LBS fwd+bwd 200k points, trainable pose path={need}: {(time.perf_counter()-t0)/30*1e3:.3f} ms", flush=True)
