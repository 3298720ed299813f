"""Dev tool: where one tracked frame's wall time goes (synthetic Replica-format sequence, 640x480): reading the
frame pair, the per-frame set-up on the device (k-NN scales, target depth render), load_frame (calibration), the
optimisation loop."""
import pathlib, sys, tempfile, time
import torch
sys.path.insert(0, ".")
from gsplatloc_amd.synthetic import write_replica_sequence as _write_sequence
from gsplatloc_amd.data.dataset import Parser
from gsplatloc_amd.graph_tracker import GraphTracker
from gsplatloc_amd.my_gsplat import TrackerConfig, init_gs_scales

def sync():
    torch.cuda.synchronize()
    return time.perf_counter()

W, H, n = 640, 480, 6
root = pathlib.Path(tempfile.mkdtemp())
_write_sequence(root, W, H, n)
parser = Parser("Replica", "room0", normalize=True, input_folder=str(root))
cfg = TrackerConfig(max_steps=2000)
tracker = None
acc = {}
for i in range(min(len(parser), n - 1)):
    t0 = sync(); d = parser[i]; t1 = sync()
    if tracker is None:
        tracker = GraphTracker(d.tar_points.shape[0], W, H, cfg, device=d.tar_points.device)
    t1 = sync(); sc = init_gs_scales(d.tar_points); t2 = sync()
    tracker.load_frame(d.tar_points, d.colors, sc, d.src_depth, d.tar_c2w, d.src_c2w, parser.K); t3 = sync()
    res = tracker.run(); t4 = sync()
    row = dict(read_and_depth_gt=t1 - t0, knn_scales=t2 - t1, load_frame=t3 - t2, optimise=t4 - t3, steps=res.steps)
    print(i, {k: (round(v * 1e3, 2) if k != "steps" else v) for k, v in row.items()})
