"""Dev tool: stage times on a depth frame whose invalid pixels pile up in one tile: one Gaussian per pixel of a
640x480 frame with rectangular holes (depth 0: those points sit at the previous camera's origin), rendered from a pose
that has moved BACKWARDS, so that the pile passes the near plane and lands in one tile (~23 k entries in its list)."""
import sys

import torch

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gsplatloc_amd.context import RenderContext, time_stages  # noqa: E402
from gsplatloc_amd.my_gsplat.geometry import depth_to_points, init_gs_scales  # noqa: E402
from gsplatloc_amd.synthetic import SH_C0, frame_pair  # noqa: E402

W, H = 640, 480
dev = torch.device("cuda")
fp = frame_pair(W, H, rot_deg=0.4, trans=0.015, seed=3)
c2w1 = fp["c2w1"].clone()
if float((-c2w1[:3, :3].T @ c2w1[:3, 3])[2]) < 0:   # make the old origin lie in FRONT of the new camera
    c2w1[:3, 3] = -c2w1[:3, 3]
depth = fp["depth0"].clone()
g = torch.Generator().manual_seed(3)
for _ in range(int(0.08 * W * H / (24 * 18))):
    x0, y0 = int(torch.randint(0, W - 24, (1,), generator=g)), int(torch.randint(0, H - 18, (1,), generator=g))
    depth[y0:y0 + 18, x0:x0 + 24] = 0.0
K = fp["K"].to(dev)
pts = depth_to_points(depth.to(dev), K).contiguous()
valid = pts[:, 2] > 0
scales = torch.full_like(pts, 1e-6)
scales[valid] = init_gs_scales(pts[valid].contiguous())
N = pts.shape[0]
sh = torch.zeros(N, 4, 3, device=dev)
sh[:, 0] = (fp["rgb"].to(dev) - 0.5) / SH_C0
inp = (pts, torch.tensor([1.0, 0, 0, 0], device=dev).repeat(N, 1).contiguous(), scales.contiguous(),
       torch.ones(N, device=dev), sh, torch.linalg.inv(c2w1).to(dev).contiguous(), K.contiguous())
ctx = RenderContext(N, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=dev, full_grads=True)
n_is = ctx.calibrate(*inp)
ctx.forward(*inp)
torch.cuda.synchronize()
longest = int((ctx.offs[1:] - ctx.offs[:-1]).max())
v = torch.zeros(H, W, 4, device=dev)
v[..., 3] = 1.0 / (W * H)
st = time_stages(ctx, inp, v, torch.zeros(H, W, 1, device=dev), True, steps=10)
print(f"intersections {n_is}, longest tile list {longest} (bins: {'direct' if ctx.bins is not None else 'two-pass'}), "
      "stages (ms): " + ", ".join(f"{k}={x:.3f}" for k, x in st.items()))
