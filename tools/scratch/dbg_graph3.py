import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
mode = sys.argv[1]
x = torch.rand(4000, 3, device="cuda", requires_grad=True)
w = torch.rand(3, 160, 112, device="cuda")
buf = torch.zeros(4000 * 32, device="cuda")
def step():
    if mode == "torch":
        y = (x * 2).sum() * 0.5 + (x[:, :1] * w[0, :100, :40].reshape(4000, 1)).mean()
        y.backward()
    elif mode == "memset":
        from mygauhuman_amd import _lib  # noqa
        torch.cuda.current_stream()
        buf.zero_()
        buf.add_(1.0)
        y = buf.sum() * x.sum()
        y.backward()
    return None
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        x.grad = None; step()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
x.grad = None; step(); torch.cuda.synchronize(); ref = x.grad.clone()
x.grad = None
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
gr = x.grad
def rel(): return float((gr - ref).abs().max() / ref.abs().max())
g.replay(); g.replay(); torch.cuda.synchronize(); print(mode, "back to back", rel())
g.replay(); torch.cuda.synchronize(); print(mode, "after host reductions", rel())
z = torch.empty(1 << 20, device="cuda"); z.fill_(1.0); z.sum().item()
g.replay(); torch.cuda.synchronize(); print(mode, "after eager fill+sum", rel())
