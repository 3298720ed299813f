"""Dev tool: wall-clock breakdown of the drop-in call at S (host side): forward pieces, backward pieces, autograd glue."""
import sys, time
import torch
sys.path.insert(0, ".")
import gsplatloc_amd as A
import gsplatloc_amd.fused as F
from gsplatloc_amd.context import RenderContext
from gsplatloc_amd.synthetic import depth_frame_scene

sc = depth_frame_scene(640, 480, stride=3); W, H = 640, 480; V = sc["viewmat"]
acc = {}
def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t; return r
    return w
for n in ("_project", "_bin", "_raster_fwd", "overflow_status", "_raster_bwd", "_project_bwd", "forward_checked", "backward"):
    setattr(RenderContext, n, timed("rc." + n, getattr(RenderContext, n)))
F._CachedRasterization.forward = staticmethod(timed("Fn.forward", F._CachedRasterization.forward))
F._CachedRasterization.backward = staticmethod(timed("Fn.backward", F._CachedRasterization.backward))
F.cached_rasterization = timed("cached_rasterization", F.cached_rasterization)
import gsplatloc_amd.rendering as R
R.cached_rasterization = F.cached_rasterization
def step(full=False):
    t0 = time.perf_counter()
    Vg = V.clone().requires_grad_()
    rc, ra, meta = A.rasterization(means=sc["means"], quats=sc["quats"], scales=sc["scales"], opacities=sc["opacities"],
                                   colors=sc["sh"], sh_degree=1, viewmats=Vg[None], Ks=sc["K"][None], width=W, height=H,
                                   packed=False, render_mode="RGB+ED", near_plane=1e-2, far_plane=1e10)
    t1 = time.perf_counter()
    loss = (rc[..., 3:4] * 0.5).sum()
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    for k, v in (("call", t1 - t0), ("loss ops", t2 - t1), ("loss.backward()", t3 - t2)):
        acc[k] = acc.get(k, 0.0) + v
for _ in range(50): step()
torch.cuda.synchronize(); acc.clear()
n = 400
t = time.perf_counter()
for _ in range(n): step()
torch.cuda.synchronize()
print(f"wall per step {(time.perf_counter() - t) / n * 1e3:.3f} ms")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:28s} {v / n * 1e6:7.1f} us")
