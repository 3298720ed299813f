"""GPU: the N > 1 path with real processes -- two ranks sharing the one GPU of the box, the 16-float all-reduce over
gloo through a pinned host buffer (the rehearsal of what the driver runs over RCCL on 2/4/8 GPUs).  Both halves use
HIP-graph replay; round 1's fault on exactly this path came from a memset node inside the captured iteration
(profiles/r02_graph_fault_diagnosis.txt)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

TRACKER_RANK = r"""
import os, sys, json
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
import gsplatloc_amd.my_gsplat as M
from gsplatloc_amd.graph_tracker import GraphTracker
from gsplatloc_amd.my_gsplat.geometry import depth_to_points
from gsplatloc_amd.synthetic import frame_pair
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
backend = sys.argv[3] if len(sys.argv) > 3 else "gloo"
if backend == "nccl":
    dist.init_process_group("nccl", device_id=dev)
else:
    dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
W, H = 320, 240
fp = frame_pair(W, H, rot_deg=0.3, trans=0.01)
K = fp["K"].to(dev)
pts0 = depth_to_points(fp["depth0"].to(dev), K)
pts1 = depth_to_points(fp["depth1"].to(dev), K)
scales = M.init_gs_scales(pts0)
src = M.compute_depth_gt(pts1, fp["rgb"].to(dev), K[None], torch.eye(4, device=dev)[None], H, W)
cfg = M.TrackerConfig(max_steps=40, min_step=5, patience=1000)
th = (H + 15) // 16
rows = None if world == 1 else [(0, th // 2), (th // 2, th)][rank]
group = dist.group.WORLD if (world > 1 or backend == "nccl") else None
guard = int(sys.argv[4]) if len(sys.argv) > 4 else 1
gt = GraphTracker(pts0.shape[0], W, H, cfg, device=dev, rows=rows, group=group, poll=10, guard_tiles=guard)
gt.load_frame(pts0, fp["rgb"].to(dev), scales, src, fp["c2w0"].to(dev), fp["c2w1"].to(dev), K)
kept0 = None if gt.kept is None else int(gt.kept.numel())
res = gt.run()
assert gt.graph is not None and (group is None or gt.collective_captured or gt.graph_tail is not None)
if rows is not None:
    assert gt.prune and gt.rc.N == int(gt.kept.numel()) < pts0.shape[0]  # the strip renders its kept Gaussians only
if rank == 0:
    json.dump({"losses": res.losses, "eT": res.best_eT, "steps": res.steps, "captured": gt.collective_captured,
               "rebuckets": gt.rebuckets, "guard": gt.guard, "kept_first": kept0,
               "kept": None if gt.kept is None else int(gt.kept.numel()), "N": int(pts0.shape[0])},
              open(sys.argv[2], "w"))
dist.destroy_process_group()
"""


def _torchrun(nproc, port, script, *args, timeout=600):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
           "127.0.0.1", "--master-port", str(port), script, *args]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=root)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    return res


def test_two_rank_bench_with_graph_replay(repo_root):
    res = _torchrun(2, 29621, "bench.py", "--gpus", "2", "--rehearse-on-one-gpu", "--gaussians", "200000", "--width", "640",
                    "--height", "480", "--steps", "10", "--warmup", "4", "--no-cpu-baseline", "--no-tracker", "--no-variants")
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["launch"].startswith("hipGraph replay") and d["scaling"] == "strong"
    assert "2 screen-tile strips" in d["config"]["parallelism"] and d["value"] > 0
    # the strip was cut out of the caller's arrays with caller-order indices and holds the full frame's lists of its rows
    # (round 4: a placed calibration context had handed out storage-order indices -- half the intersections went missing)
    assert d["config"]["strip_lists_match_full_frame_rank0"] is True


def test_graph_tracker_two_ranks_match_one(tmp_path, repo_root):
    """GraphTracker split over two tile-row strips (one-pixel halo, pack kernel, ONE all-reduce per iteration, both
    graph halves replayed) follows the single-rank tracker: same loss trajectory, same pose."""
    script = tmp_path / "tracker_rank.py"
    script.write_text(TRACKER_RANK)
    out = {}
    for n, port in ((1, 29622), (2, 29623)):
        f = tmp_path / f"res{n}.json"
        _torchrun(n, port, str(script), repo_root, str(f))
        out[n] = json.loads(f.read_text())
    a, b = torch.tensor(out[1]["losses"]), torch.tensor(out[2]["losses"])
    assert out[1]["steps"] == out[2]["steps"] == 40
    # the sum over strips regroups float32 additions: the first iterations agree to rounding, later ones carry the
    # difference through Adam's normalisation (as every trajectory comparison in this suite)
    assert torch.allclose(a[:3], b[:3], rtol=1e-5), (a[:3], b[:3])
    assert torch.allclose(a, b, rtol=5e-3), (a, b)
    assert abs(out[1]["eT"] - out[2]["eT"]) < 1e-4


def test_allreduce_is_captured_inside_the_iteration_graph_over_rccl(tmp_path, repo_root):
    """over RCCL the 16-float all-reduce is a node of the iteration's HIP graph (one replay per
    iteration).  A one-GPU box can only form an RCCL group of ONE rank; that still sends the collective through
    torch's ProcessGroupNCCL and RCCL's captured launch path.  The tracker with the captured collective must follow the
    tracker without a group exactly (a sum over one rank), and bench.py must report the captured form."""
    script = tmp_path / "tracker_rank.py"
    script.write_text(TRACKER_RANK)
    f0, f1 = tmp_path / "plain.json", tmp_path / "rccl.json"
    _torchrun(1, 29624, str(script), repo_root, str(f0))
    _torchrun(1, 29625, str(script), repo_root, str(f1), "nccl")
    a, b = json.loads(f0.read_text()), json.loads(f1.read_text())
    assert b["captured"] is True and a["captured"] is False
    assert a["steps"] == b["steps"] == 40
    assert torch.allclose(torch.tensor(a["losses"]), torch.tensor(b["losses"]), rtol=1e-6), (a["losses"][:3], b["losses"][:3])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "bench.py", "--collective-with-one-rank", "--gaussians", "200000", "--width", "640",
                          "--height", "480", "--steps", "10", "--warmup", "4", "--no-cpu-baseline", "--no-tracker",
                          "--no-variants"], capture_output=True, text=True, timeout=600, cwd=root)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert "all-reduce captured in the graph" in d["config"]["launch"], d["config"]["launch"]


def test_gaussians_that_leave_the_guard_band_rebucket_every_rank(tmp_path, repo_root):
    """SURVEY.md 8(e) / VERDICT r3 item 6 on hardware: two ranks on the one GPU, strips that keep ONLY the Gaussians
    touching their rendered rows at the initial pose (guard band of 0 tiles): the first pose updates bring other splats
    into the strips, the poll's full-N projection sees them, both ranks widen the band to one tile, bucket again and re-run
    the frame -- and the trajectory is the one-rank tracker's."""
    script = tmp_path / "tracker_rank.py"
    script.write_text(TRACKER_RANK)
    out = {}
    for n, port, extra in ((1, 29626, ()), (2, 29627, ("gloo", "0"))):
        f = tmp_path / f"res{n}.json"
        _torchrun(n, port, str(script), repo_root, str(f), *extra)
        out[n] = json.loads(f.read_text())
    two = out[2]
    assert two["rebuckets"] >= 1 and two["guard"] >= 1, two  # the band was left, noticed and widened
    assert two["kept_first"] < two["kept"] < two["N"], two    # the wider band keeps more, still not everything
    a, b = torch.tensor(out[1]["losses"]), torch.tensor(two["losses"])
    assert out[1]["steps"] == two["steps"] == 40
    assert torch.allclose(a[:3], b[:3], rtol=1e-5), (a[:3], b[:3])
    assert torch.allclose(a, b, rtol=5e-3), (a, b)
    assert abs(out[1]["eT"] - two["eT"]) < 1e-4
