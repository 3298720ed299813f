#!/usr/bin/env python3
"""Dev tool: trip statistics of the G16 backward walk (needs a library built with -DGSL_G16_STATS:
make -C gsplatloc_amd/csrc clean all EXTRA=-DGSL_G16_STATS).  One forward + backward of workload R."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from gsplatloc_amd import context as C  # noqa: E402
from gsplatloc_amd._lib import load_library  # noqa: E402
from gsplatloc_amd.synthetic import perturbed_pose, random_scene  # noqa: E402

dev = torch.device("cuda")
N, W, H = 1_000_000, 1200, 680
sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
sc = random_scene(N, W, H, sigma_px=sigma, device=dev, order="random")
V = torch.linalg.inv(perturbed_pose()).to(dev).contiguous()
ctx = C.RenderContext(N, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=dev, full_grads=True)
inp = (sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["sh"], V, sc["K"].contiguous())
n_is = ctx.calibrate(*inp)
v = torch.zeros(H, W, 4)
v[..., 3] = torch.randn(H, W, generator=torch.Generator().manual_seed(1))
v = v.to(dev)
va = torch.zeros(H, W, 1, device=dev)
lib = load_library()
out = (ctypes.c_ulonglong * 8)()
ctx.forward(*inp)
torch.cuda.synchronize()
lib.gsl_g16_stats(out, 1)
ctx.backward(v, va, full=True)
torch.cuda.synchronize()
lib.gsl_g16_stats(out, 1)
trips, valid, act, pairs, clash, lanes, kmax_sum, walked = [int(x) for x in out[:8]]
print(f"workload R sigma {sigma}: intersections {n_is}")
print(f"wave trips {trips} (sum over batches of the longest row list: {kmax_sum}); trips with a composited pixel {valid} "
      f"({valid / max(trips, 1):.1%}); groups busy per trip {walked / max(trips, 1):.2f} (lists per quadrant: 4, or 8 with -DGSL_NG=8)")
print(f"(block, entry) pairs walked {walked}; pairs with a composited pixel {pairs}; composited (pixel, entry) pairs {lanes} "
      f"({lanes / max(pairs, 1):.2f} of 16 lanes per pair)")
