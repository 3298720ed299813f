"""Dev tool: the latency of ONE tile through the compositing kernels -- a 64x64 image whose Gaussians all project into
tile (1, 1), n of them, sigma 1 px -- i.e. the per-tile critical path that floors a strip with fewer tiles than the chip
has SIMDs.  Prints per n the stage times (HIP events, eager launches, median of 30)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gsplatloc_amd.context import RenderContext, time_stages  # noqa: E402
from gsplatloc_amd.synthetic import SH_C0, replica_intrinsics  # noqa: E402

dev = torch.device("cuda")
W = H = 64
K = replica_intrinsics(W, H)
fx, fy, cx, cy = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])
for n in (64, 256, 512, 768, 1024, 1536, 2048):
    g = torch.Generator().manual_seed(n)
    u = 19.0 + 10.0 * torch.rand(n, generator=g)   # 3 px inside tile (1, 1): the 3-sigma box stays in the tile
    v = 19.0 + 10.0 * torch.rand(n, generator=g)
    z = 1.0 + 4.0 * torch.rand(n, generator=g)
    means = torch.stack([(u - cx) / fx * z, (v - cy) / fy * z, z], -1).to(dev)
    quats = torch.tensor([1.0, 0, 0, 0]).repeat(n, 1).to(dev)
    scales = (1.0 * z / fx)[:, None].repeat(1, 3).to(dev)
    opac = torch.full((n,), 0.05, device=dev)      # low opacity: nobody stops early, the whole list is walked
    sh = torch.zeros(n, 4, 3, device=dev)
    sh[:, 0] = (torch.rand(n, 3, generator=g).to(dev) - 0.5) / SH_C0
    inp = (means, quats, scales, opac, sh, torch.eye(4, device=dev), K.to(dev).contiguous())
    ctx = RenderContext(n, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=dev, full_grads=True)
    ctx.calibrate(*inp)
    vr = torch.zeros(H, W, 4, device=dev)
    vr[..., 3] = 1.0
    st = time_stages(ctx, inp, vr, torch.zeros(H, W, 1, device=dev), True, steps=30)
    longest = int((ctx.offs[1:] - ctx.offs[:-1]).max())
    print(f"n={n} longest list {longest}: " + " ".join(f"{k}={x * 1e3:.1f}us" for k, x in st.items()), flush=True)
