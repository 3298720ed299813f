import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gsplatloc_amd import context as C
from gsplatloc_amd.context import RenderContext
from gsplatloc_amd.parallel import strip_rows, gaussians_for_strip
from gsplatloc_amd.synthetic import random_scene, perturbed_pose
sig = float(sys.argv[1]); order = sys.argv[2]; world = int(sys.argv[3]); rank = int(sys.argv[4])
N, W, H = 1_000_000, 1200, 680
dev = torch.device("cuda")
sc = random_scene(N, W, H, sigma_px=sig, device=dev, order=order)
viewmat = torch.linalg.inv(perturbed_pose()).to(dev).contiguous()
K = sc["K"].contiguous()
g = torch.Generator().manual_seed(1)
v = torch.zeros(H, W, 4); v[..., 3] = torch.randn(H, W, generator=g); v = v.to(dev)
va = torch.zeros(H, W, 1, device=dev)
cal = RenderContext(N, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=dev, full_grads=False)
cal.calibrate(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["sh"], viewmat, K)
rows = strip_rows(cal.offs, cal.tw, cal.th, world)[rank]
idx = gaussians_for_strip(cal.Q0[:, 0:2], cal.radii, rows)
loc = {k: sc[k][idx].contiguous() for k in ("means", "quats", "scales", "opacities", "sh")}
n = loc["means"].shape[0]
ctx = RenderContext(n, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=dev, tile_rows=rows, full_grads=True)
inp = (loc["means"], loc["quats"], loc["scales"], loc["opacities"], loc["sh"], viewmat, K)
ni = ctx.calibrate(*inp)
print("rows", rows, "n_local", n, "isects", ni, "tiny", ctx.tiny)
print(C.time_stages(ctx, inp, v, va, True, steps=20))
