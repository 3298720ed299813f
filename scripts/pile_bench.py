"""Dev tool: stage times on the depth frame whose invalid pixels pile up in one tile (the frame of
tests/test_gpu_configs.py::test_config_T_depth_frame_with_invalid_pixels; 24 k entries in one tile list)."""
import sys

import torch

sys.path.insert(0, ".")
from gsplatloc_amd.context import RenderContext, time_stages  # noqa: E402
from tests.test_gpu_configs import _tum_like_frame  # noqa: E402

sc, fp, n_valid = _tum_like_frame()
W, H = 640, 480
dev = torch.device("cuda")
V = torch.linalg.inv(fp["c2w1"])
inp = tuple(sc[k].to(dev).contiguous() for k in ("means", "quats", "scales", "opacities", "sh")) + (
    V.to(dev).contiguous(), sc["K"].to(dev).contiguous())
ctx = RenderContext(sc["means"].shape[0], W, H, "RGB+ED", sh_degree=1, K_sh=4, device=dev, full_grads=True)
n_is = ctx.calibrate(*inp)
ctx.forward(*inp)
torch.cuda.synchronize()
off = ctx.offs.cpu()
longest = int((off[1:] - off[:-1]).max())
v = torch.zeros(H, W, 4, device=dev)
v[..., 3] = 1.0 / (W * H)
st = time_stages(ctx, inp, v, torch.zeros(H, W, 1, device=dev), True, steps=10)
print(f"intersections {n_is}, longest tile list {longest}, stages (ms): " + ", ".join(f"{k}={x:.3f}" for k, x in st.items()))
