"""Dev tool: device k-NN scale initialisation (init_gs_scales) on back-projected depth frames."""
import sys, time
import torch
sys.path.insert(0, ".")
from gsplatloc_amd.my_gsplat import init_gs_scales
from gsplatloc_amd.my_gsplat.geometry import depth_to_points
from gsplatloc_amd.synthetic import frame_pair
for W, H in ((640, 480), (1200, 680)):
    fp = frame_pair(W, H)
    pts = depth_to_points(fp["depth0"].cuda(), fp["K"].cuda()).contiguous()
    for _ in range(3): init_gs_scales(pts)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): init_gs_scales(pts)
    torch.cuda.synchronize()
    print(f"{W}x{H}: {pts.shape[0]} points, init_gs_scales {(time.perf_counter() - t) / 10 * 1e3:.2f} ms")
