import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsplatloc_amd as A
from gsplatloc_amd import _lib
from tests.scenes import random_scene, sh_from_rgb, small_pose
lib = _lib.load_library()
x = torch.randn(64, 32, device='cuda'); out = torch.empty(64, device='cuda')
_lib.check(lib.gsl_debug_reduce_scatter(x.data_ptr(), out.data_ptr(), None), 'rs')
ref = x.sum(0)
print('reduce_scatter err', float((out.cpu() - ref.cpu().repeat_interleave(2)).abs().max()))
print(out[:8].cpu(), ref[:4].cpu())
W, H, N = 160, 120, 6000
sc = random_scene(N, W, H, dtype=torch.float32, sigma_px=1.2, opacity=(0.4, 1.0))
for mode, shd in (("D", None), ("ED", None), ("RGB", None), ("RGB+ED", 1)):
    colors = sh_from_rgb(sc["rgbs"]) if shd is not None else sc["rgbs"]
    V = torch.linalg.inv(small_pose(0.5, 0.01, dtype=torch.float32))[None]
    res = {}
    for fused in ("0", "1"):
        os.environ["GSLOC_DISABLE_FUSED"] = fused
        ins = [sc[k].cuda().clone().requires_grad_() for k in ("means", "quats", "scales", "opacities")] + [colors.cuda().clone().requires_grad_(), V.cuda().clone().requires_grad_()]
        rc, ra, meta = A.rasterization(*ins[:5], viewmats=ins[5], Ks=sc["K"][None].cuda(), width=W, height=H, sh_degree=shd, packed=False, render_mode=mode)
        g = torch.Generator().manual_seed(2); v = torch.randn(rc.shape, generator=g).cuda(); va = torch.randn(ra.shape, generator=g).cuda()
        ((rc * v).sum() + (ra * va).sum()).backward()
        res[fused] = [rc.detach(), ra.detach()] + [t.grad for t in ins]
    names = ["render", "alpha", "v_means", "v_quats", "v_scales", "v_opac", "v_colors", "v_view"]
    print(mode)
    for n, a, b in zip(names, res["0"], res["1"]):
        if a is None or b is None: print("  ", n, "None", a is None, b is None); continue
        print("  ", n, float((a - b).abs().max() / b.abs().max().clamp(min=1e-30)))
