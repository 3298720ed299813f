"""Dev tool: where does the wall time of one drop-in fwd+bwd at S go?  cProfile over 300 calls."""
import cProfile, pstats, sys, time
import torch
sys.path.insert(0, ".")
import gsplatloc_amd as A
from gsplatloc_amd.synthetic import depth_frame_scene

sc = depth_frame_scene(640, 480, stride=3); W, H = 640, 480; V = sc["viewmat"]
def step(full=True):
    Vg = V.clone().requires_grad_()
    m = sc["means"].clone().requires_grad_(full)
    rc, ra, meta = A.rasterization(means=m, quats=sc["quats"], scales=sc["scales"], opacities=sc["opacities"],
                                   colors=sc["sh"], sh_degree=1, viewmats=Vg[None], Ks=sc["K"][None], width=W, height=H,
                                   packed=False, render_mode="RGB+ED", near_plane=1e-2, far_plane=1e10)
    (rc[..., 3] * 0.5).sum().backward()
for _ in range(20): step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize()
print("wall per step", (time.perf_counter() - t) / 200 * 1e3, "ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
