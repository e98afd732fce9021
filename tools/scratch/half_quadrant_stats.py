"""How many (Gaussian, 8x8 quadrant) survivors of the C3 scene touch only one 8x4 half of the quadrant?  (numpy estimate,
pixel-centre test alpha >= 1/255, saturation ignored) -- sizing of a 'two survivors per pass, one per half-wave' blend loop."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mygauhuman_amd import synthetic
from oracle import oracle

P, W, H = int(os.environ.get("P", 200000)), 1024, 1024
cam, g = synthetic.uniform_scene(P, W, H, 0, 3)
pre = oracle.preprocess(g["means3D"], g["opacities"], cam["viewmatrix"], cam["projmatrix"], cam["campos"], W, H, cam["tanfovx"],
                        cam["tanfovy"], scales=g["scales"], rotations=g["rotations"], shs=g["shs"], degree=3)
vis = np.nonzero(pre["radii"] > 0)[0]
rad = pre["radii"][vis]
xy = pre["means2D"][vis]
co = pre["conic_opacity"][vis]
QX, QY = W // 8, H // 8
# per quadrant counters: top-only, bottom-only, both ; also left-only/right-only/both
cnt = np.zeros((QY, QX, 3), np.int64)
cnt_lr = np.zeros((QY, QX, 3), np.int64)
lanes = 0
order = np.argsort(rad)
CH = 2000
for s in range(0, len(vis), CH):
    idx = order[s:s + CH]
    r = int(rad[idx].max())
    n = len(idx)
    x0 = np.floor(xy[idx, 0]).astype(np.int64) - r
    y0 = np.floor(xy[idx, 1]).astype(np.int64) - r
    ox = np.arange(2 * r + 2)
    px = x0[:, None] + ox[None, :]          # [n, w]
    py = y0[:, None] + ox[None, :]
    dx = xy[idx, 0][:, None] - px           # [n, w]
    dy = xy[idx, 1][:, None] - py
    a, b, c, o = co[idx, 0], co[idx, 1], co[idx, 2], co[idx, 3]
    power = -0.5 * (a[:, None, None] * dx[:, None, :] ** 2 + c[:, None, None] * dy[:, :, None] ** 2) - b[:, None, None] * dx[:, None, :] * dy[:, :, None]
    alpha = np.minimum(0.99, o[:, None, None] * np.exp(power))
    hit = (power <= 0) & (alpha >= 1.0 / 255.0)
    hit &= ((px >= 0) & (px < W))[:, None, :] & ((py >= 0) & (py < H))[:, :, None]
    gi, yi, xi = np.nonzero(hit)
    lanes += len(gi)
    PX, PY = px[gi, xi], py[gi, yi]
    q = (PY // 8) * QX + (PX // 8)
    key = gi.astype(np.int64) * (QX * QY) + q
    top = (PY % 8) < 4
    left = (PX % 8) < 4
    uk, inv = np.unique(key, return_inverse=True)
    has_top = np.zeros(len(uk), bool); has_bot = np.zeros(len(uk), bool)
    has_l = np.zeros(len(uk), bool); has_r = np.zeros(len(uk), bool)
    has_top[inv[top]] = True; has_bot[inv[~top]] = True
    has_l[inv[left]] = True; has_r[inv[~left]] = True
    qq = uk % (QX * QY)
    for arr, A, B in ((cnt, has_top, has_bot), (cnt_lr, has_l, has_r)):
        flat = arr.reshape(-1, 3)
        np.add.at(flat[:, 0], qq[A & ~B], 1)
        np.add.at(flat[:, 1], qq[~A & B], 1)
        np.add.at(flat[:, 2], qq[A & B], 1)
for name, arr in (("top/bottom 8x4", cnt), ("left/right 4x8", cnt_lr)):
    tot = arr.sum()
    it = (arr[..., 2] + np.maximum(arr[..., 0], arr[..., 1])).sum()
    print(f"{name}: survivors {tot}, only-first {arr[...,0].sum()/tot:.3f}, only-second {arr[...,1].sum()/tot:.3f}, both {arr[...,2].sum()/tot:.3f};"
          f" paired iterations {it} = {it/tot:.3f} of today's")
print("hit lanes per survivor:", lanes / cnt.sum(), " lane utilisation", lanes / cnt.sum() / 64)
