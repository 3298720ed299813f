"""GPU: the pose-tracking loop (mirror of gs_trainer_total.Runner.train's per-frame body) against
the CPU oracle's tracker on a synthetic frame pair, and the two engines against each other."""
import pytest
import torch

from gsplatloc_amd.synthetic import frame_pair
from oracle import tracker_oracle as T

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _setup(W=160, H=120):
    import gsplatloc_amd.my_gsplat as M
    fp = frame_pair(W, H, rot_deg=0.3, trans=0.01)
    K = fp["K"]
    pts0 = T.depth_to_points(fp["depth0"], K)          # tar cloud, world == camera-0 frame
    pts1 = T.depth_to_points(fp["depth1"], K)          # src cloud in its own camera frame
    scales0 = T.init_gs_scales(pts0, as_coded=True)     # reference's as-coded kNN^2 scales (CPU KD-tree)
    scales1 = T.init_gs_scales(pts1, as_coded=True)
    return M, fp, K, pts0, pts1, scales0, scales1


def test_depth_gt_and_tracking_follow_the_oracle():
    M, fp, K, pts0, pts1, scales0, scales1 = _setup()
    W, H = fp["W"], fp["H"]
    N = pts1.shape[0]
    # query depth: "ED" render of the src cloud from the identity pose (geometry.py:69-135)
    quats = torch.tensor([1.0, 0, 0, 0]).repeat(N, 1)
    sh = torch.zeros(N, 4, 3)
    sh[:, 0] = T.rgb_to_sh(fp["rgb"])
    with torch.no_grad():
        gt_o, _, _ = T.gs_forward(pts1.double(), quats.double(), scales1.double(), torch.ones(N).double(), sh.double(),
                                  torch.eye(4).double(), K.double(), W, H, render_mode="ED")
    import gsplatloc_amd as A
    gt_g, _, _ = A.rasterization(means=pts1.to(DEV), quats=quats.to(DEV), scales=scales1.to(DEV),
                                 opacities=torch.ones(N, device=DEV), colors=sh.to(DEV), sh_degree=1,
                                 viewmats=torch.eye(4, device=DEV)[None], Ks=K[None].to(DEV), width=W, height=H,
                                 far_plane=1e10, near_plane=1e-2, render_mode="ED", rasterize_mode="classic", packed=False)
    bad = ((gt_g.cpu().double() - gt_o).abs() > 1e-5 + 1e-4 * gt_o.abs()).double().mean().item()
    assert bad < 3e-3, bad
    src_depth = gt_g.detach()  # [1,H,W,1]

    steps = 12
    res_o = T.track_frame(pts0.double(), scales0.double(), fp["rgb"].double(), gt_o, K.double(), W, H,
                          init_c2w=fp["c2w0"].double(), gt_c2w=fp["c2w1"].double(), max_steps=steps, min_step=2)
    cfg = M.TrackerConfig(max_steps=steps, min_step=2)
    out = {}
    for engine in ("autograd", "context"):
        trk = M.PoseTracker(cfg, engine=engine)
        out[engine] = trk.track_frame(pts0.to(DEV), fp["rgb"].to(DEV), src_depth, fp["c2w0"].to(DEV), fp["c2w1"].to(DEV),
                                      K.to(DEV), W, H, scales=scales0.to(DEV))
    lo = torch.tensor(res_o.losses, dtype=torch.float64)
    for engine, r in out.items():
        lg = torch.tensor(r.losses, dtype=torch.float64)
        assert r.steps == steps
        # Adam's first steps amplify tiny gradient differences, so the per-step loss tolerance is 1e-3 relative
        assert torch.allclose(lg, lo, rtol=2e-3, atol=1e-7), (engine, lg, lo)
        assert abs(r.best_eT - res_o.best_eT) < 2e-4 and abs(r.best_eR - res_o.best_eR) < 5e-3
    la, lc = torch.tensor(out["autograd"].losses, dtype=torch.float64), torch.tensor(out["context"].losses, dtype=torch.float64)
    # two engines, two backward kernels (the operator API keeps the general backward, the context picks the tiny-splat
    # one): gradients agree to ~1e-5, and twelve Adam steps carry that into the fourth digit of the loss
    assert torch.allclose(la, lc, rtol=5e-4), "context engine must reproduce the autograd engine"


def test_tracker_converges_to_ground_truth_pose():
    """Perturbed start (0.3 deg, 1 cm) returns towards GT: the synthetic stand-in for the ATE check."""
    M, fp, K, pts0, pts1, scales0, scales1 = _setup()
    W, H = fp["W"], fp["H"]
    src_depth = M.compute_depth_gt(pts1.to(DEV), fp["rgb"].to(DEV), K[None].to(DEV), torch.eye(4, device=DEV)[None], H, W)
    # (utils.knn inside compute_depth_gt recomputes the as-coded scales on the host)
    src_depth = src_depth[None, ..., None]
    cfg = M.TrackerConfig(max_steps=150, min_step=20, patience=1000)
    trk = M.PoseTracker(cfg, engine="context")
    r = trk.track_frame(pts0.to(DEV), fp["rgb"].to(DEV), src_depth, fp["c2w0"].to(DEV), fp["c2w1"].to(DEV), K.to(DEV),
                        W, H, scales=scales0.to(DEV))
    e0 = M.calculate_translation_error(fp["c2w0"], fp["c2w1"])
    assert r.losses[-1] < 0.5 * r.losses[0]
    assert r.best_eT < 0.5 * e0, (r.best_eT, e0)


def test_fused_loss_kernel_matches_autograd_loss():
    """gsl_tracking_loss (depth L1 + Sobel-edge L1, value and gradient) vs the PyTorch loss of my_gsplat.loss,
    whole image and a strip with halo."""
    import gsplatloc_amd.my_gsplat as M
    from gsplatloc_amd._lib import check, load_library, ptr
    from gsplatloc_amd.parallel import strip_tracking_loss
    lib = load_library()
    W, H, D = 75, 52, 4
    g = torch.Generator().manual_seed(5)
    render = torch.rand(H, W, D, generator=g) * 3 + 0.5
    render[5:9, 10:30, 3] = 0.0           # holes: mask = (depth != 0)
    render[:, 0, 3] = 0.0
    gt = torch.rand(H, W, generator=g) * 3 + 0.5
    render, gt = render.to(DEV), gt.to(DEV)
    trk = M.PoseTracker()
    ws_bytes = lib.gsl_loss_ws_bytes(W, H)
    ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=DEV)
    for rows in (None, (1, 3)):
        r = render.clone().requires_grad_()
        depths = r[None, ..., 3:4]
        if rows is None:
            total, dl, sl = trk.tracking_loss(depths, gt[None, ..., None])
            row0, row1 = 0, H
        else:
            total, dl, sl = strip_tracking_loss(depths, gt[None, ..., None], rows, H)
            row0, row1 = rows[0] * 16, min(rows[1] * 16, H)
        total.backward()
        v = torch.zeros(H, W, D, device=DEV)
        nb = lib.gsl_loss_n_partials(W, H, row0, row1)
        partials = torch.zeros(nb * 2, device=DEV)
        check(lib.gsl_tracking_loss(ptr(render), D, ptr(gt), W, H, row0, row1, 0.8, 0.2, ptr(v), ptr(partials), None,
                                    ptr(ws), ws_bytes, None), "loss")
        torch.cuda.synchronize()
        sums = partials.view(-1, 2).sum(0) / (W * H)
        assert abs(float(sums[0]) - float(dl)) < 1e-5 * float(dl) + 1e-9
        assert abs(float(sums[1]) - float(sl)) < 1e-4 * float(sl) + 1e-9
        ref = r.grad[..., 3]
        assert float((v[..., 3] - ref).abs().max()) < 1e-5 * float(ref.abs().max()) + 1e-12
        assert float(v[..., :3].abs().max()) == 0.0


def test_fused_normal_loss_kernel_matches_autograd_loss():
    """gsl_normal_loss (the normal-consistency term the reference keeps switched off: row-wise cosine of the normal
    maps of the masked depth images) on top of gsl_tracking_loss vs autograd of the float64 oracle loss, whole image
    and a strip with its one-row halo.  Tolerance: 2e-4 of the largest gradient entry (float32 cross products of
    central differences against float64)."""
    import oracle.tracker_oracle as TO
    from gsplatloc_amd._lib import check, load_library, ptr
    from gsplatloc_amd.parallel import strip_tracking_loss
    from gsplatloc_amd.synthetic import replica_intrinsics
    lib = load_library()
    W, H, D = 75, 52, 4
    K = replica_intrinsics(W, H, dtype=torch.float64)
    g = torch.Generator().manual_seed(5)
    gt = torch.rand(H, W, generator=g, dtype=torch.float64) * 3 + 0.5
    render = torch.rand(H, W, D, generator=g, dtype=torch.float64)
    render[..., 3] = gt + 0.05 * torch.randn(H, W, generator=g, dtype=torch.float64)
    render[5:9, 10:30, 3] = 0.0           # holes: mask = (depth != 0)
    render[:, 0, 3] = 0.0
    render, gt = render.float().double(), gt.float().double()  # the values the kernels see
    r32, gt32 = render.float().to(DEV), gt.float().to(DEV)
    ws_bytes, nws_bytes = lib.gsl_loss_ws_bytes(W, H), lib.gsl_normal_ws_bytes(W, H)
    ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=DEV)
    nws = torch.zeros(nws_bytes, dtype=torch.uint8, device=DEV)
    lam_d, lam_n = 0.7, 0.1
    fx, fy, cx, cy = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])
    for rows in (None, (1, 3), (0, 1)):
        r = render.clone().requires_grad_()
        depths = r[None, ..., 3:4]
        if rows is None:
            total, dl, sl = TO.tracking_loss(depths, gt[None, ..., None], lam_d, lam_n, K)
            row0, row1 = 0, H
        else:
            total, dl, sl = strip_tracking_loss(depths, gt[None, ..., None], rows, H, lam_d, lam_n, K=K)
            row0, row1 = rows[0] * 16, min(rows[1] * 16, H)
        total.backward()
        v = torch.zeros(H, W, D, device=DEV)
        nb = lib.gsl_loss_n_partials(W, H, row0, row1)
        partials = torch.zeros(nb * 2, device=DEV)
        nsum = torch.full((1,), -7.0, device=DEV)
        check(lib.gsl_tracking_loss(ptr(r32), D, ptr(gt32), W, H, row0, row1, lam_d, 1 - lam_d - lam_n, ptr(v),
                                    ptr(partials), None, ptr(ws), ws_bytes, None), "loss")
        check(lib.gsl_normal_loss(ptr(r32), D, ptr(gt32), W, H, row0, row1, fx, fy, cx, cy, lam_n, ptr(v), ptr(nsum),
                                  ptr(nws), nws_bytes, None), "normal")
        torch.cuda.synchronize()
        sums = partials.view(-1, 2).sum(0).double().cpu() / (W * H)
        value = lam_d * sums[0] + (1 - lam_d - lam_n) * sums[1] + lam_n * ((row1 - row0) / H - float(nsum) / (3 * H))
        assert abs(float(value) - float(total)) < 2e-5 * abs(float(total)), (float(value), float(total))
        ref = r.grad[..., 3]
        err = float((v[..., 3].double().cpu() - ref).abs().max()) / float(ref.abs().max())
        print(f"[parity] normal-consistency term rows={rows}: value rel {abs(float(value) - float(total)) / abs(float(total)):.1e}, "
              f"v_depth rel-inf {err:.1e}")
        assert err < 2e-4, err
        assert float(v[..., :3].abs().max()) == 0.0


def test_graph_tracker_with_the_normal_term_follows_the_autograd_tracker():
    """normal_lambda != 0 end to end: the device loop (loss kernels + normal kernels + pose step in one graph) against
    the PyTorch loop with the same weights."""
    M, fp, K, pts0, pts1, scales0, scales1 = _setup()
    from gsplatloc_amd.graph_tracker import GraphTracker
    W, H = fp["W"], fp["H"]
    src_depth = M.compute_depth_gt(pts1.to(DEV), fp["rgb"].to(DEV), K[None].to(DEV), torch.eye(4, device=DEV)[None], H, W)
    src_depth = src_depth[None, ..., None]
    steps = 30
    cfg = M.TrackerConfig(max_steps=steps, min_step=5, patience=1000, depth_lambda=0.7, normal_lambda=0.05)
    ref = M.PoseTracker(cfg, engine="context").track_frame(
        pts0.to(DEV), fp["rgb"].to(DEV), src_depth, fp["c2w0"].to(DEV), fp["c2w1"].to(DEV), K.to(DEV), W, H,
        scales=scales0.to(DEV))
    gt = GraphTracker(pts0.shape[0], W, H, cfg)
    gt.load_frame(pts0.to(DEV), fp["rgb"].to(DEV), scales0.to(DEV), src_depth, fp["c2w0"].to(DEV), fp["c2w1"].to(DEV),
                  K.to(DEV))
    res = gt.run()
    assert res.steps == ref.steps == steps
    lg, lr = torch.tensor(res.losses), torch.tensor(ref.losses)
    rel = float(((lg - lr).abs() / lr).max())
    print(f"[parity] tracker with normal term: loss0 rel {abs(float(lg[0] - lr[0])) / float(lr[0]):.1e}, trajectory rel {rel:.1e}")
    assert abs(float(lg[0] - lr[0])) < 2e-5 * float(lr[0])
    assert rel < 5e-3, rel  # Adam amplifies float32 differences of the first updates (same bound as the two-engine test)


@pytest.mark.parametrize("use_graph", [False, True])
def test_graph_tracker_matches_pose_tracker(use_graph):
    """Device-side loss + pose chain + Adam + LR decay + early-stop bookkeeping reproduce the PyTorch loop."""
    M, fp, K, pts0, pts1, scales0, scales1 = _setup()
    from gsplatloc_amd.graph_tracker import GraphTracker
    W, H = fp["W"], fp["H"]
    src_depth = M.compute_depth_gt(pts1.to(DEV), fp["rgb"].to(DEV), K[None].to(DEV), torch.eye(4, device=DEV)[None], H, W)
    src_depth = src_depth[None, ..., None]
    steps = 40
    cfg = M.TrackerConfig(max_steps=steps, min_step=5, patience=1000)
    ref = M.PoseTracker(cfg, engine="context").track_frame(
        pts0.to(DEV), fp["rgb"].to(DEV), src_depth, fp["c2w0"].to(DEV), fp["c2w1"].to(DEV), K.to(DEV), W, H,
        scales=scales0.to(DEV))
    gt = GraphTracker(pts0.shape[0], W, H, cfg, device=DEV, use_graph=use_graph, poll=10)
    gt.load_frame(pts0.to(DEV), fp["rgb"].to(DEV), scales0.to(DEV), src_depth, fp["c2w0"].to(DEV), fp["c2w1"].to(DEV),
                  K.to(DEV))
    res = gt.run()
    assert res.steps == steps == ref.steps
    la, lb = torch.tensor(res.losses, dtype=torch.float64), torch.tensor(ref.losses, dtype=torch.float64)
    assert torch.allclose(la, lb, rtol=2e-3, atol=1e-8), (la, lb)
    assert abs(res.best_loss - ref.best_loss) < 2e-3 * ref.best_loss
    assert abs(res.best_eT - ref.best_eT) < 1e-4 and abs(res.best_eR - ref.best_eR) < 2e-3
    # a second frame on the same tracker (buffers and graph reused)
    gt.load_frame(pts0.to(DEV), fp["rgb"].to(DEV), scales0.to(DEV), src_depth, fp["c2w0"].to(DEV), fp["c2w1"].to(DEV),
                  K.to(DEV))
    res2 = gt.run()
    assert torch.allclose(torch.tensor(res2.losses, dtype=torch.float64), la, rtol=1e-4)


def test_graph_tracker_early_stop():
    M, fp, K, pts0, pts1, scales0, scales1 = _setup()
    from gsplatloc_amd.graph_tracker import GraphTracker
    W, H = fp["W"], fp["H"]
    src_depth = M.compute_depth_gt(pts1.to(DEV), fp["rgb"].to(DEV), K[None].to(DEV), torch.eye(4, device=DEV)[None], H, W)
    cfg = M.TrackerConfig(max_steps=400, min_step=3, patience=4)
    gt = GraphTracker(pts0.shape[0], W, H, cfg, device=DEV, poll=8)
    gt.load_frame(pts0.to(DEV), fp["rgb"].to(DEV), scales0.to(DEV), src_depth, fp["c2w0"].to(DEV), fp["c2w1"].to(DEV), K.to(DEV))
    res = gt.run()
    ref = M.PoseTracker(cfg, engine="context").track_frame(
        pts0.to(DEV), fp["rgb"].to(DEV), src_depth[None, ..., None], fp["c2w0"].to(DEV), fp["c2w1"].to(DEV), K.to(DEV),
        W, H, scales=scales0.to(DEV))
    assert res.steps < 400
    assert abs(res.steps - ref.steps) <= 2, (res.steps, ref.steps)  # fp32 ties in "loss < best" may shift the stop by an iteration


@pytest.mark.parametrize("odd_size,mode", [(False, "RGB+ED"), (True, "RGB+ED"), (False, "ED")])
def test_loss_computed_inside_the_tiny_backward_matches_the_separate_loss_launch(odd_size, mode, monkeypatch):
    """One rank, no normal term, tiny-splat backward: gsl_tiny_raster_bwd(..., loss_depth_gt, ...) computes the tracking
    loss of its tile and back-propagates from it -- same upstream gradient, same loss partials, same pose gradient rows
    and the same trajectory as gsl_tracking_loss followed by the plain backward (GSLOC_FUSE_LOSS=0).  odd_size: an image
    that is not a multiple of the tile size (partial tiles at the right and bottom borders)."""
    M, fp, K, pts0, pts1, scales0, scales1 = _setup() if not odd_size else _setup(W=150, H=107)
    from gsplatloc_amd.graph_tracker import GraphTracker
    W, H = fp["W"], fp["H"]
    src_depth = M.compute_depth_gt(pts1.to(DEV), fp["rgb"].to(DEV), K[None].to(DEV), torch.eye(4, device=DEV)[None], H, W)
    cfg = M.TrackerConfig(max_steps=40, min_step=5, patience=1000)
    frame = (pts0.to(DEV), fp["rgb"].to(DEV), scales0.to(DEV), src_depth, fp["c2w0"].to(DEV), fp["c2w1"].to(DEV), K.to(DEV))
    out = {}
    for fuse in ("0", "1"):
        monkeypatch.setenv("GSLOC_FUSE_LOSS", fuse)
        gt = GraphTracker(pts0.shape[0], W, H, cfg, device=DEV, poll=10, use_graph=False, render_mode=mode)
        gt.load_frame(*frame)
        assert gt.rc.tiny and gt.rc.can_fuse_tracking_loss() and gt.rc.D == (4 if mode == "RGB+ED" else 1)
        gt._render_and_loss()  # one iteration's forward, loss, backward at the initial pose
        torch.cuda.synchronize()
        rows_ptr, n_rows = gt.rc.viewmat_rows()
        one = dict(v=gt.v_render.clone(), partials=gt.partials.clone(), vcT=gt.rc.vcT.clone())
        gt = GraphTracker(pts0.shape[0], W, H, cfg, device=DEV, poll=10, render_mode=mode)
        gt.load_frame(*frame)
        res = gt.run()
        out[fuse] = dict(one=one, losses=torch.tensor(res.losses, dtype=torch.float64), c2w=res.final_c2w.clone())
    a, b = out["0"], out["1"]
    assert torch.equal(a["one"]["v"], b["one"]["v"])
    assert torch.equal(a["one"]["partials"], b["one"]["partials"])
    assert torch.equal(a["one"]["vcT"], b["one"]["vcT"])
    assert torch.allclose(a["losses"], b["losses"], rtol=1e-6)
    assert torch.allclose(a["c2w"], b["c2w"], rtol=0, atol=1e-6)


def test_graph_tracker_recovers_from_overflows():
    """Neither a splat that outgrows the tiny backward nor an intersection list that outgrows its buffer nor a tile
    list that outgrows its bin costs the frame: all are flagged on the device, read at the poll, and the frame is re-run from its initial pose with the
    general backward / a larger buffer -- with the result of a tracker that had the right set-up from the start."""
    M, fp, K, pts0, pts1, scales0, scales1 = _setup()
    from gsplatloc_amd.graph_tracker import GraphTracker
    W, H = fp["W"], fp["H"]
    src_depth = M.compute_depth_gt(pts1.to(DEV), fp["rgb"].to(DEV), K[None].to(DEV), torch.eye(4, device=DEV)[None], H, W)
    cfg = M.TrackerConfig(max_steps=30, min_step=5, patience=1000)
    big = torch.full_like(scales0, 6e-3).to(DEV)  # ~1.5 px at 2.5 m: r_cull > 2 px
    frame = lambda sc: (pts0.to(DEV), fp["rgb"].to(DEV), sc, src_depth, fp["c2w0"].to(DEV), fp["c2w1"].to(DEV), K.to(DEV))  # noqa: E731
    ref = GraphTracker(pts0.shape[0], W, H, cfg, device=DEV, poll=10)
    ref.load_frame(*frame(big))
    assert not ref.rc.tiny
    want = ref.run()
    gt = GraphTracker(pts0.shape[0], W, H, cfg, device=DEV, poll=10)
    gt.load_frame(*frame(scales0.to(DEV)))      # calibrated on the as-coded (pixel-sized) splats ...
    assert gt.rc.tiny
    gt.scales.copy_(big)                        # ... which then grow: tiny overflow AND more intersections
    gt.rc._alloc_isects(int(gt.rc.n_is.item()) + 16)
    gt.rc._alloc_bins(8)                        # ... and tile lists that outgrow their bins
    got = gt.run()
    assert not gt.rc.tiny and gt.rc.capacity > int(gt.rc.n_is.item())
    assert gt.rc.bin_cap > 8 and gt.rc.bins_overflowed() == 0
    assert got.steps == want.steps == 30
    assert torch.allclose(torch.tensor(got.losses), torch.tensor(want.losses), rtol=1e-4)


@pytest.mark.parametrize("case", ["depthmap", "random", "duplicates"])
def test_device_knn_matches_kdtree(case):
    """csrc/knn.hip vs scipy cKDTree (the stand-in for small_gicp's KdTree, utils.py:16-22): exact k-NN."""
    import numpy as np
    from scipy.spatial import cKDTree
    from gsplatloc_amd.my_gsplat.utils import knn_device
    g = torch.Generator().manual_seed(3)
    if case == "depthmap":
        fp = frame_pair(160, 120)
        pts = T.depth_to_points(fp["depth0"], fp["K"])
    elif case == "random":
        pts = torch.rand(20000, 3, generator=g) * torch.tensor([4.0, 0.3, 2.0]) + torch.tensor([-1.0, 5.0, 0.0])
    else:
        pts = torch.rand(3000, 3, generator=g)
        pts[:700] = 0.0  # invalid-depth pixels all back-project to the origin
    k = 5
    d_ref, _ = cKDTree(pts.double().numpy()).query(pts.double().numpy(), k=k)
    d2 = knn_device(pts.to(DEV), k).cpu().double().numpy()
    np.testing.assert_allclose(d2, d_ref ** 2, rtol=2e-5, atol=1e-10)
    # and through the reference-shaped scale initialisation (as-coded: squared distances squared again)
    import gsplatloc_amd.my_gsplat as M
    s_dev = M.init_gs_scales(pts.to(DEV)).cpu()
    s_ref = T.init_gs_scales(pts, as_coded=True)
    assert torch.allclose(s_dev, s_ref, rtol=1e-4, atol=1e-12)


def test_graph_tracker_on_a_frame_with_a_pile():
    """A TUM-like frame pair whose camera steps BACKWARDS: the invalid (zero-depth) points of the previous frame pass
    the near plane and pile ~23 k entries into one tile list (/root/reference/src/data/Image.py:29-35 filters nothing).
    The tracker -- tiny-splat backward for the ordinary tiles, long-list split for the pile, both feeding the
    projection backward, everything inside one HIP graph -- follows the tracker that walks the pile with one workgroup
    (GSLOC_LONG_LISTS=0), and is an order of magnitude faster per iteration."""
    import os
    import time

    import gsplatloc_amd.my_gsplat as M
    from gsplatloc_amd.graph_tracker import GraphTracker
    from gsplatloc_amd.synthetic import depth_frame_scene

    W, H = 640, 480
    sc = depth_frame_scene(W, H, stride=1, holes=True, device=DEV, pile=True)
    N = sc["means"].shape[0]
    c2w1 = torch.linalg.inv(sc["viewmat"])
    # target depth: the cloud seen from a slightly different pose; initial pose = the scene's camera
    gt_c2w = c2w1.clone()
    gt_c2w[0, 3] += 0.004
    rgb = (sc["sh"][:, 0, :] * 0.28209479177387814 + 0.5).contiguous()
    target = M.compute_depth_gt(sc["means"], rgb, sc["K"][None], gt_c2w[None], H, W)
    cfg = M.TrackerConfig(max_steps=40, min_step=5, patience=1000)
    out = {}
    for mode in ("1", "0"):
        os.environ["GSLOC_LONG_LISTS"] = mode
        try:
            gt = GraphTracker(N, W, H, cfg, device=DEV, poll=20)
            gt.load_frame(sc["means"], rgb, sc["scales"], target, c2w1, gt_c2w, sc["K"])
        finally:
            os.environ.pop("GSLOC_LONG_LISTS", None)
        assert (gt.rc.long_min > 0) == (mode == "1")
        longest = int((gt.rc.offs[1:] - gt.rc.offs[:-1]).max())
        assert longest > 20_000, longest
        torch.cuda.synchronize()
        t = time.perf_counter()
        res = gt.run()
        torch.cuda.synchronize()
        out[mode] = (res, (time.perf_counter() - t) / max(res.steps, 1))
    (ra, ta), (rb, tb) = out["1"], out["0"]
    assert ra.steps == rb.steps == 40
    la, lb = torch.tensor(ra.losses), torch.tensor(rb.losses)
    # the first iterations agree to rounding; later the pile jumps tens of pixels per step (it sits 1.5 cm in front of the
    # camera) and Adam amplifies last-bit differences of the transmittance product into slightly different steps
    assert torch.allclose(la[:3], lb[:3], rtol=2e-5), (la[:3], lb[:3])
    assert torch.allclose(la[:8], lb[:8], rtol=2e-3), (la[:8], lb[:8])
    assert torch.allclose(la, lb, rtol=0.15), (la, lb)
    assert la[-1] < 0.5 * la[0]
    print(f"[perf] pile frame, tracker iteration: {ta * 1e3:.3f} ms split over workgroups, {tb * 1e3:.3f} ms one workgroup")
    assert ta < 0.5 * tb


def test_pile_that_comes_into_view_during_the_optimisation_switches_the_split_on():
    """The usual way a pile arises in GsplatLoc: the invalid points sit at the TARGET frame's camera origin, which is
    where the optimisation starts (/root/reference/src/data/dataset.py:349-350: both clouds are placed with the target
    pose), so at calibration they are behind the near plane; they only enter the view once the camera has moved back
    far enough.  The bins sized at calibration then overflow, the poll sees it, and the recovery sizes the bins for the
    pile AND switches the long-list split on (calibrate() could not have)."""
    import gsplatloc_amd.my_gsplat as M
    from gsplatloc_amd.graph_tracker import GraphTracker
    from gsplatloc_amd.synthetic import depth_frame_scene

    W, H = 640, 480
    sc = depth_frame_scene(W, H, stride=1, holes=True, device=DEV, pile=True)
    N = sc["means"].shape[0]
    gt_c2w = torch.linalg.inv(sc["viewmat"])          # 1.5 cm behind the previous camera: the pile is in view there
    start = torch.eye(4, device=DEV)                  # the previous camera itself: the pile sits at z = 0, culled
    rgb = (sc["sh"][:, 0, :] * 0.28209479177387814 + 0.5).contiguous()
    target = M.compute_depth_gt(sc["means"], rgb, sc["K"][None], gt_c2w[None], H, W)
    cfg = M.TrackerConfig(max_steps=150, min_step=5, patience=1000)
    gt = GraphTracker(N, W, H, cfg, device=DEV, poll=25)
    gt.load_frame(sc["means"], rgb, sc["scales"], target, start, gt_c2w, sc["K"])
    assert gt.rc.long_min == 0 and int((gt.rc.offs[1:] - gt.rc.offs[:-1]).max()) < 2000
    res = gt.run()
    assert res.steps == 150 and torch.isfinite(torch.tensor(res.losses)).all()
    assert gt.rc.long_min > 0 and gt.rc.bin_cap > 20_000, (gt.rc.long_min, gt.rc.bin_cap)
    assert res.losses[-1] < 0.5 * res.losses[0]
