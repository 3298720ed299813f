import os, sys, time, json
sys.path.insert(0, os.getcwd())
import torch
import gsplatloc_amd.my_gsplat as M
from gsplatloc_amd.graph_tracker import GraphTracker
from gsplatloc_amd.synthetic import frame_pair
from gsplatloc_amd.my_gsplat.geometry import depth_to_points
dev = "cuda"
for name, W, H, stride in (("S", 640, 480, 3), ("T", 640, 480, 1), ("R", 1200, 680, 1)):
    fp = frame_pair(W, H, rot_deg=0.3, trans=0.01)
    K = fp["K"].to(dev)
    pts0 = depth_to_points(fp["depth0"].to(dev), K)[::stride].contiguous()
    rgb = fp["rgb"].to(dev)[::stride].contiguous()
    pts1 = depth_to_points(fp["depth1"].to(dev), K)
    scales = M.init_gs_scales(pts0)
    src = M.compute_depth_gt(pts1, fp["rgb"].to(dev), K[None], torch.eye(4, device=dev)[None], H, W)[None, ..., None]
    cfg = M.TrackerConfig(max_steps=200, min_step=100, patience=10**9)
    out = {}
    for mode in ("RGB+ED", "ED"):
        gt = GraphTracker(pts0.shape[0], W, H, cfg, device=dev, poll=50, render_mode=mode)
        args = (pts0, rgb, scales, src, fp["c2w0"].to(dev), fp["c2w1"].to(dev), K)
        gt.load_frame(*args); gt.run(); gt.load_frame(*args)
        torch.cuda.synchronize(); t = time.perf_counter(); res = gt.run(); torch.cuda.synchronize(); dt = time.perf_counter() - t
        out[mode] = (round(res.steps / dt), res.best_eT, res.losses[-1])
    print(name, pts0.shape[0], out)
