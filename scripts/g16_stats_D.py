import ctypes, os, sys
sys.path.insert(0, "/root/repo")
import torch
from gsplatloc_amd import context as C
from gsplatloc_amd._lib import load_library
from gsplatloc_amd.synthetic import depth_frame_scene
dev = torch.device("cuda")
W, H = 1200, 680
sc = depth_frame_scene(W, H, stride=1, holes=False, device=dev)
N = sc["means"].shape[0]
ctx = C.RenderContext(N, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=dev, full_grads=True)
inp = (sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["sh"], sc["viewmat"], sc["K"].contiguous())
n_is = ctx.calibrate(*inp)
ctx.use_general_backward()
v = torch.zeros(H, W, 4); v[..., 3] = torch.randn(H, W, generator=torch.Generator().manual_seed(1)); v = v.to(dev)
va = torch.zeros(H, W, 1, device=dev)
lib = load_library(); out = (ctypes.c_ulonglong * 8)()
ctx.forward(*inp); torch.cuda.synchronize(); lib.gsl_g16_stats(out, 1)
ctx.backward(v, va, full=True); torch.cuda.synchronize(); lib.gsl_g16_stats(out, 1)
trips, valid, act, pairs, clash, lanes, kmax_sum, walked = [int(x) for x in out[:8]]
print(f"D: N {N} intersections {n_is}; trips {trips}; rows busy {act / max(trips,1):.2f}; pairs {pairs}; pixel pairs {lanes} ({lanes/max(pairs,1):.2f} lanes/pair); quadrant hits {int(ctx.hit_counts[:-1].sum())}")
