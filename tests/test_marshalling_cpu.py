"""Python -> ctypes -> C marshalling of every call the RenderContext and the GraphTracker make, checked on a
machine WITHOUT a GPU: the HIP runtime refuses the launch and each entry point returns GSL_ERR_HIP (-3), which
can only happen after ctypes accepted the argument list (count and types) and the entry point's own argument
validation passed.  A wrong argument list raises ctypes.ArgumentError / TypeError instead, a rejected argument
returns -1.  Never runs where a GPU is present (the pointers are host pointers)."""
import pytest
import torch

pytestmark = pytest.mark.skipif(torch.cuda.is_available(), reason="host-pointer calls: only meaningful without a GPU")


def _expect_hip_refusal(fn, what):
    with pytest.raises(RuntimeError, match=r"HIP launch error \(status -3\)"):
        fn()
    return what


@pytest.mark.parametrize("mode,full,tiny,pixel_rows", [("RGB+ED", True, False, None), ("ED", False, True, None),
                                                       ("RGB+ED", True, True, (15, 48)), ("RGB", True, False, (16, 33))])
def test_render_context_stage_calls_marshal(mode, full, tiny, pixel_rows, monkeypatch):
    import gsplatloc_amd.context as CX
    from gsplatloc_amd.synthetic import perturbed_pose, random_scene

    monkeypatch.setattr(CX, "current_stream", lambda: None)
    N, W, H = 500, 64, 48
    rows = (0, 3) if pixel_rows == (15, 48) else (1, 3)
    ctx = CX.RenderContext(N, W, H, mode, sh_degree=1, K_sh=4, device="cpu", full_grads=full, tile_rows=rows,
                           pixel_rows=pixel_rows)
    ctx._alloc_isects(4096)
    if tiny:
        ctx.tiny = True
        ctx.trec = torch.zeros(N, 32)
        ctx.vcT = torch.zeros(H, W, ctx.D)
    sc = random_scene(N, W, H, sigma_px=1.0)
    inp = (sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["sh"],
           torch.linalg.inv(perturbed_pose()).contiguous(), sc["K"].contiguous())
    ctx._inputs = inp
    v, va = torch.zeros(H, W, ctx.D), torch.zeros(H, W, 1)
    _expect_hip_refusal(lambda: ctx._project(*inp), "project")
    _expect_hip_refusal(ctx._bin, "bin")
    _expect_hip_refusal(ctx._raster_fwd, "raster fwd")
    _expect_hip_refusal(lambda: ctx._raster_bwd(v, va), "raster bwd")
    with pytest.raises(RuntimeError, match=r"gsl_fused_project_bwd failed: HIP launch error \(status -3\)"):
        ctx._project_bwd(full)


def test_tracker_kernel_calls_marshal():
    from gsplatloc_amd._lib import check, load_library, ptr

    lib = load_library()
    W, H, D = 64, 48, 4
    render, gt, v_render = torch.zeros(H, W, D), torch.ones(H, W), torch.zeros(H, W, D)
    ws_bytes = lib.gsl_loss_ws_bytes(W, H)
    ws = torch.zeros(ws_bytes, dtype=torch.uint8)
    n_part = lib.gsl_loss_n_partials(W, H, 0, H)
    assert n_part == 4 * 3 and lib.gsl_loss_n_partials(W, H, 16, 32) == 4 * 2 and lib.gsl_loss_n_partials(W, H, 5, 5) == 0
    partials = torch.zeros(n_part * 2)
    pose_f, pose_i = torch.zeros(32), torch.zeros(4, dtype=torch.int32)
    c2w, viewmat, eye = torch.eye(4), torch.eye(4), torch.eye(4)
    hist, v_viewmat = torch.zeros(10), torch.zeros(16)
    _expect_hip_refusal(lambda: check(lib.gsl_tracking_loss(ptr(render), D, ptr(gt), W, H, 0, H, 0.8, 0.2, ptr(v_render),
                                                            ptr(partials), None, ptr(ws), ws_bytes, None), "loss"), "loss")
    _expect_hip_refusal(lambda: check(lib.gsl_pose_init(ptr(pose_f), ptr(pose_i), ptr(eye), 5e-4, 1e-3, ptr(c2w),
                                                        ptr(viewmat), None), "pose_init"), "pose init")
    nws_bytes = lib.gsl_normal_ws_bytes(W, H)
    assert nws_bytes == (W * H * 6 + H * 10) * 4
    nws, nsum = torch.zeros(nws_bytes, dtype=torch.uint8), torch.zeros(1)
    _expect_hip_refusal(lambda: check(lib.gsl_normal_loss(ptr(render), D, ptr(gt), W, H, 0, H, 300.0, 300.0, 31.5, 23.5, 0.1,
                                                          ptr(v_render), ptr(nsum), ptr(nws), nws_bytes, None), "normal"),
                        "normal loss")
    assert lib.gsl_normal_loss(ptr(render), D, ptr(gt), W, H, 0, H, 0.0, 300.0, 31.5, 23.5, 0.1, ptr(v_render), ptr(nsum),
                               ptr(nws), nws_bytes, None) == -1  # fx = 0: bad argument, checked before any launch
    assert lib.gsl_normal_loss(ptr(render), D, ptr(gt), W, H, 0, H, 300.0, 300.0, 31.5, 23.5, 0.1, ptr(v_render), ptr(nsum),
                               ptr(nws), nws_bytes - 4, None) < 0  # workspace too small
    _expect_hip_refusal(lambda: check(lib.gsl_pose_step(ptr(pose_f), ptr(pose_i), ptr(v_viewmat), None, 0, None, ptr(partials), n_part,
                                                        None, ptr(nsum), ptr(eye), W, H, 0.7, 0.2, 0.1, 0.9, 0.999, 1e-8,
                                                        1e-3, 1e-3, 0.99, 100, 200, 1, 10, ptr(c2w), ptr(viewmat),
                                                        ptr(hist), None),
                                      "pose_step"), "pose step")
    # k-NN set-up
    pts, bbox = torch.rand(100, 3), torch.tensor([0.0, 0, 0, 1, 1, 1])
    kws = lib.gsl_knn_ws_bytes(100)
    kw = torch.zeros(kws, dtype=torch.uint8)
    _expect_hip_refusal(lambda: check(lib.gsl_knn_count(ptr(pts), 100, ptr(bbox), ptr(kw), kws, None), "knn"), "knn count")


def test_graph_tracker_iteration_marshals_every_call(monkeypatch):
    """GraphTracker.load_frame() and one _iteration() on host tensors: every C call is reached (the status check
    is relaxed to 'refused by the HIP runtime') and none is rejected by ctypes or by argument validation."""
    import gsplatloc_amd.context as CX
    import gsplatloc_amd.graph_tracker as GT
    from gsplatloc_amd.my_gsplat import TrackerConfig
    from gsplatloc_amd.synthetic import frame_pair

    calls = []

    def refused(status, what):
        calls.append(what)
        assert status == -3, (what, status)

    class _Stream:
        def __init__(self, *a, **k):
            pass

    monkeypatch.setattr(CX, "current_stream", lambda: None)
    monkeypatch.setattr(GT, "current_stream", lambda: None)
    monkeypatch.setattr(CX, "check", refused)
    monkeypatch.setattr(GT, "check", refused)
    monkeypatch.setattr(torch.cuda, "Stream", _Stream)
    W, H = 64, 48
    fp = frame_pair(W, H, rot_deg=0.3, trans=0.01)
    from gsplatloc_amd.my_gsplat.geometry import depth_to_points
    pts = depth_to_points(fp["depth0"], fp["K"])
    gt = GT.GraphTracker(pts.shape[0], W, H, TrackerConfig(max_steps=5), device="cpu", use_graph=False)
    gt.load_frame(pts, fp["rgb"], torch.full((pts.shape[0], 3), 0.01), fp["depth1"], fp["c2w0"], fp["c2w1"], fp["K"])
    n_setup = len(calls)
    gt._iteration()
    assert calls[:n_setup] == ["gsl_pose_init", "gsl_fused_project"]
    # nothing was projected (the launch was refused), so every r_cull is 0 and calibration picked the tiny backward
    assert gt.rc.tiny
    # one rank, no normal term, tiny backward: the compositing forward sorts its own tile's bin (no gsl_fused_bin) and
    # the compositing backward computes the loss itself (no gsl_tracking_loss): five launches
    assert gt.rc.sorts_in_forward()
    assert calls[n_setup:] == ["gsl_fused_project", "gsl_fused_raster_fwd", "gsl_tiny_raster_bwd",
                               "gsl_fused_project_bwd", "gsl_pose_step"]
    monkeypatch.setenv("GSLOC_FUSE_LOSS", "0")
    monkeypatch.setenv("GSLOC_SORT_IN_FORWARD", "0")
    del calls[:]
    gt._iteration()
    assert calls == ["gsl_fused_project", "gsl_fused_bin", "gsl_fused_raster_fwd", "gsl_tracking_loss",
                     "gsl_tiny_raster_bwd", "gsl_fused_project_bwd", "gsl_pose_step"]


@pytest.mark.parametrize("mode,sh_degree", [("RGB+ED", 1), ("ED", None), ("RGB", None)])
def test_fused_autograd_function_marshals(mode, sh_degree, monkeypatch):
    """The gsplat-compatible entry's autograd function (fused.py), forward and backward, on host tensors."""
    import gsplatloc_amd.fused as F
    from gsplatloc_amd.synthetic import perturbed_pose, random_scene

    calls = []

    def refused(status, what):  # 0: nothing to launch (no intersections); -3: launch refused; never a rejected argument
        calls.append(what)
        assert status in (0, -3), (what, status)

    monkeypatch.setattr(F, "check", refused)
    monkeypatch.setattr(F, "current_stream", lambda: None)
    monkeypatch.setattr(torch, "empty", torch.zeros)  # the intersection count is read back from an output buffer
    N, W, H = 300, 64, 48
    sc = random_scene(N, W, H, sigma_px=1.0)
    colors = sc["sh"] if sh_degree is not None else torch.rand(N, 3)
    ins = [sc[k].clone().requires_grad_() for k in ("means", "quats", "scales", "opacities")]
    col = colors.clone().requires_grad_()
    V = torch.linalg.inv(perturbed_pose()).contiguous().requires_grad_()
    cfg = (W, H, -1 if sh_degree is None else sh_degree, mode, 0.3, 0.01, 1e10, 0.0, False, 0, 3, True)
    raw = {}
    render, alphas, last = F._FusedRasterization.apply(*ins, col, V, sc["K"].contiguous(), cfg, raw)
    assert render.shape == (H, W, F._MODES[mode][0]) and alphas.shape == (H, W, 1) and last.shape == (H, W)
    assert {"radii", "Q0", "Q1", "tile_offsets", "flatten_ids", "n_isects"} <= set(raw)
    assert raw["Q0"].shape == (N, 4) and raw["Q0"].is_contiguous()
    (render.sum() + alphas.sum()).backward()
    assert calls[:3] == ["gsl_fused_project", "gsl_fused_bin", "gsl_fused_raster_fwd"]
    assert calls[3:] == ["gsl_fused_raster_bwd", "gsl_fused_project_bwd"]
    assert V.grad is not None and V.grad.shape == (4, 4) and ins[0].grad.shape == (N, 3)
    assert (col.grad is not None) == mode.startswith("RGB")


def test_sequence_evaluation_cli_on_host_tensors(tmp_path, monkeypatch, capsys):
    """python -m gsplatloc_amd.eval end to end on Replica-format files with every kernel launch refused: the
    reader, the Parser, the GraphTracker glue, the report writer and the argument parser all run."""
    import functools
    import json

    import gsplatloc_amd.context as CX
    import gsplatloc_amd.eval as EV
    import gsplatloc_amd.graph_tracker as GT
    from gsplatloc_amd.data import Parser
    from tests.test_data import write_replica

    def refused(status, what):
        assert status in (0, -3), (what, status)

    class _Stream:
        def __init__(self, *a, **k):
            pass

    monkeypatch.setattr(CX, "current_stream", lambda: None)
    monkeypatch.setattr(GT, "current_stream", lambda: None)
    monkeypatch.setattr(CX, "check", refused)
    monkeypatch.setattr(GT, "check", refused)
    monkeypatch.setattr(torch.cuda, "Stream", _Stream)
    monkeypatch.setattr(EV, "Parser", functools.partial(Parser, device="cpu"))
    monkeypatch.setattr(EV, "GraphTracker", functools.partial(GT.GraphTracker, use_graph=False))  # no graph capture here
    write_replica(tmp_path, n=3)
    out = tmp_path / "res.json"
    EV.main(["--dataset", "Replica", "--rooms", "room0", "--root", str(tmp_path), "--num-iters", "3", "--no-normalize",
             "--max-frames", "2", "--out", str(out)])
    rep = json.loads(out.read_text())["room0"]["gsplatloc_amd"]
    assert rep["frames"] == 2 and set(rep) >= {"ATE", "AAE", "frames_with_result", "mean_steps", "seconds"}
    assert "room0" in capsys.readouterr().out


GROUP_RANK = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
import gsplatloc_amd.context as CX
import gsplatloc_amd.graph_tracker as GT
from gsplatloc_amd.my_gsplat import TrackerConfig
from gsplatloc_amd.my_gsplat.geometry import depth_to_points
from gsplatloc_amd.synthetic import frame_pair

calls = []
def refused(status, what):
    calls.append(what)
    assert status in (0, -3), (what, status)
class _Stream:
    def __init__(self, *a, **k): pass
CX.current_stream = GT.current_stream = lambda: None
CX.check = GT.check = refused
torch.cuda.Stream = _Stream
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
W, H = 64, 48
fp = frame_pair(W, H, rot_deg=0.3, trans=0.01)
pts = depth_to_points(fp["depth0"], fp["K"])
rows = [(0, 2), (2, 3)][rank]                       # 3 tile rows split over the two ranks
gt = GT.GraphTracker(pts.shape[0], W, H, TrackerConfig(max_steps=5), device="cpu", rows=rows, group=dist.group.WORLD)
assert gt.render_rows == [(0, 3), (1, 3)][rank]      # one halo tile row towards the neighbour
gt.load_frame(pts, fp["rgb"], torch.full((pts.shape[0], 3), 0.01), fp["depth1"], fp["c2w0"], fp["c2w1"], fp["K"])
assert gt.pixel_rows == [(0, 33), (31, 48)][rank]    # of which ONE pixel row is composited
# what gsl_pack_pose_reduce would have left (the launch is refused here): this rank's gradient and loss sums
gt.reduce_buf[:12] = float(rank + 1)
gt.reduce_buf[12:14] = torch.tensor([0.5 * gt.n_partials, 0.25 * gt.n_partials])
mine = gt.reduce_buf[12:14].clone()
gt._iteration()
assert calls[-1] == "gsl_pose_step" and "gsl_tracking_loss" in calls and "gsl_pack_pose_reduce" in calls
assert torch.all(gt.reduce_buf[:12] == 3.0), gt.reduce_buf        # 1 + 2: the pose gradient of both strips
sums = [torch.zeros(2) for _ in range(world)]
dist.all_gather(sums, mine)
assert torch.allclose(gt.reduce_buf[12:14], sums[0] + sums[1])    # loss sums of both strips
print(f"rank {rank} ok", flush=True)
dist.destroy_process_group()
"""


def test_graph_tracker_group_mode_over_gloo(tmp_path):
    """GraphTracker with rows= / group= on two gloo ranks (launches refused): strip bookkeeping, the single
    16-float all-reduce and the shared pose step are reached on both ranks with identical reduced values."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "group_rank.py"
    script.write_text(GROUP_RANK)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29547", str(script), root]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    assert "rank 0 ok" in res.stdout and "rank 1 ok" in res.stdout


GROUP_OVERFLOW = GROUP_RANK[:GROUP_RANK.index("# what gsl_pack_pose_reduce would have left")] + """
# an overflow on ONE strip only (rank 1: a splat outgrew the tiny backward): the decision is MAX-reduced at the poll,
# so both ranks re-run the frame together, both switch to the general backward, and their collectives stay paired
gt.use_graph = False
gt.rc.tiny = True
if rank == 1:
    gt.rc.flags[0] = 1
n_coll = [0]
real = gt._collective
def counted():
    n_coll[0] += 1
    real()
gt._collective = counted
res = gt.run()
assert n_coll[0] == 10, n_coll          # 5 iterations, recovery, 5 iterations again -- on BOTH ranks
assert gt.rc.tiny is False and int(gt.rc.flags[0]) == 0
print(f"rank {rank} ok", flush=True)
dist.destroy_process_group()
"""


def test_graph_tracker_overflow_on_one_rank_is_recovered_by_all(tmp_path):
    """the overflow recovery must not be decided from rank-local state.  Two gloo ranks, the sticky
    tiny-backward flag raised on rank 1 only: both ranks redo the frame (10 collectives each) and finish together."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "group_overflow.py"
    script.write_text(GROUP_OVERFLOW)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29548", str(script), root]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    assert "rank 0 ok" in res.stdout and "rank 1 ok" in res.stdout


GROUP_BAND = GROUP_RANK[:GROUP_RANK.index("# what gsl_pack_pose_reduce would have left")] + """
# A Gaussian leaves the guard band of ONE strip (rank 0 reports 3 violators at its first poll): the count is MAX-reduced
# with the overflow flags, so BOTH ranks widen their band, bucket again at the initial pose and re-run the frame
# together -- their collectives stay paired (SURVEY.md 8e: "fall back ... if a splat leaves its band").
assert gt.prune and gt.guard == 1 and gt.kept_mask is not None
gt.use_graph = False
fired = [False]
def violations():
    if rank == 0 and not fired[0]:
        fired[0] = True
        return 3
    return 0
gt._band_violations = violations
buckets = [0]
real_bucket = gt._bucket
def counted_bucket():
    buckets[0] += 1
    real_bucket()
gt._bucket = counted_bucket
n_coll = [0]
real = gt._collective
def counted():
    n_coll[0] += 1
    real()
gt._collective = counted
res = gt.run()
assert n_coll[0] == 10, n_coll          # 5 iterations, recovery, 5 iterations again -- on BOTH ranks
assert gt.guard == 2 and gt.rebuckets == 1 and buckets[0] == 1, (gt.guard, gt.rebuckets, buckets)
print(f"rank {rank} ok", flush=True)
dist.destroy_process_group()
"""


def test_graph_tracker_guard_band_violation_on_one_rank_rebuckets_all(tmp_path):
    """SURVEY.md 8(e) / VERDICT r3 item 6: strips keep only the Gaussians within their guard band, and the band is
    checked at every poll.  Two gloo ranks, a violation reported on rank 0 only: both ranks double the band, bucket
    again and redo the frame (10 collectives each)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "group_band.py"
    script.write_text(GROUP_BAND)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29549", str(script), root]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    assert "rank 0 ok" in res.stdout and "rank 1 ok" in res.stdout
