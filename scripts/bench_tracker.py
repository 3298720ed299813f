"""Pose-optimisation iterations/s (BASELINE metric 2) on synthetic depth-map-like frame pairs.
Configs of BASELINE.json: S ~100k Gaussians 640x480 200 iters; T ~300k 640x480; R 816k 1200x680."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gsplatloc_amd.my_gsplat as M
from gsplatloc_amd.graph_tracker import GraphTracker
from gsplatloc_amd.synthetic import frame_pair
from gsplatloc_amd.my_gsplat.geometry import depth_to_points

def run(name, W, H, stride, iters, engines, normal_lambda=0.0):
    dev = "cuda"
    fp = frame_pair(W, H, rot_deg=0.3, trans=0.01)
    K = fp["K"].to(dev)
    pts0 = depth_to_points(fp["depth0"].to(dev), K)[::stride].contiguous()
    rgb = fp["rgb"].to(dev)[::stride].contiguous()
    pts1 = depth_to_points(fp["depth1"].to(dev), K)
    N = pts0.shape[0]
    t0 = time.perf_counter(); scales = M.init_gs_scales(pts0); t_knn = time.perf_counter() - t0
    src_depth = M.compute_depth_gt(pts1, fp["rgb"].to(dev), K[None], torch.eye(4, device=dev)[None], H, W)[None, ..., None]
    cfg = M.TrackerConfig(max_steps=iters, min_step=100, patience=10**9, depth_lambda=0.8 - normal_lambda,
                          normal_lambda=normal_lambda)
    out = {"config": name, "normal_lambda": normal_lambda, "N": N, "W": W, "H": H, "iters": iters, "knn_scale_init_s": t_knn}
    for eng in engines:
        if eng == "graph":
            gt = GraphTracker(N, W, H, cfg, device=dev, poll=50)
            gt.load_frame(pts0, rgb, scales, src_depth, fp["c2w0"].to(dev), fp["c2w1"].to(dev), K)
            gt.run(); gt.load_frame(pts0, rgb, scales, src_depth, fp["c2w0"].to(dev), fp["c2w1"].to(dev), K)
            torch.cuda.synchronize(); t = time.perf_counter(); res = gt.run(); torch.cuda.synchronize(); dt = time.perf_counter() - t
        else:
            trk = M.PoseTracker(cfg, engine=eng)
            args = (pts0, rgb, src_depth, fp["c2w0"].to(dev), fp["c2w1"].to(dev), K, W, H)
            c2 = M.TrackerConfig(max_steps=5, min_step=100, patience=10**9, depth_lambda=0.8 - normal_lambda,
                                 normal_lambda=normal_lambda)
            M.PoseTracker(c2, engine=eng).track_frame(*args, scales=scales)
            torch.cuda.synchronize(); t = time.perf_counter(); res = trk.track_frame(*args, scales=scales); torch.cuda.synchronize(); dt = time.perf_counter() - t
        e0 = M.calculate_translation_error(fp["c2w0"], fp["c2w1"])
        out[eng] = {"iters_per_s": res.steps / dt, "ms_per_iter": dt / res.steps * 1e3, "loss0": res.losses[0], "lossN": res.losses[-1],
                    "eT_init": e0, "best_eT": res.best_eT, "best_eR": res.best_eR}
    print(json.dumps(out))

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "S"
    engines = sys.argv[2].split(",") if len(sys.argv) > 2 else ["graph", "context", "autograd"]
    if which == "S": run("S", 640, 480, 3, 200, engines)
    if which == "T": run("T", 640, 480, 1, 200, engines)
    if which == "Tn": run("T + normal term (BASELINE configs[2] as worded)", 640, 480, 1, 200, engines, normal_lambda=0.1)
    if which == "R": run("R", 1200, 680, 1, 200, engines)
