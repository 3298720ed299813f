"""Parity at BASELINE.json's configurations, full size, on the GPU (configs[1..4]; configs[0] is the CPU ICP case,
tests/test_icp_baseline.py):

  S  Replica room0-like frame pair: ~100k Gaussians, 640x480, 200 pose-optimisation iterations (GraphTracker)
  T  TUM fr1/desk-like frame: one Gaussian per pixel of a 640x480 depth frame (~300k) with invalid (zero) depths
  R  1 M Gaussians, 1200x680 (the bench headline, random order sigma_px = 1, and the depth-frame regime)
  X  5 M random Gaussians, 1920x1080

Checker: the float64 C restatement (oracle/csrc/gsplat_oracle.c), which finishes these sizes in seconds.
Tolerances: images 1e-4 relative (+2e-5 absolute), pose gradient 1e-4 of its largest entry, both as north_star
states them; the gradient comparison is flip-aware (tests/parity.py) and the flipped-pixel fraction is bounded.
"""
import os

import numpy as np
import pytest
import torch

from tests.parity import FLOOR32_MAX, POSE_GRAD_TOL, agreeing_pixels, pose_grad_bound, rel_inf, report

pytestmark = pytest.mark.gpu
THREADS = min(os.cpu_count() or 1, 16)


def _context_vs_c_oracle(tag, sc, V, W, H, v, max_flipped, grad_names=("means", "scales", "opacities"),
                         with_deterministic=True):
    """RenderContext forward + backward against the C oracle on the same inputs; returns the error report."""
    from gsplatloc_amd.context import RenderContext
    from oracle import c_oracle as C

    dev = torch.device("cuda")
    N = sc["means"].shape[0]
    cpu = [sc[k] for k in ("means", "quats", "scales", "opacities", "sh")]
    want_f = C.rasterization(*cpu, V, sc["K"], W, H, sh_degree=1, render_mode="RGB+ED", precision="f64", threads=THREADS)
    ctx = RenderContext(N, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=dev, full_grads=True)
    inp = tuple(t.to(dev).contiguous() for t in cpu) + (V.to(dev).contiguous(), sc["K"].to(dev).contiguous())
    n_is = ctx.calibrate(*inp)
    render, alphas = ctx.forward(*inp)
    torch.cuda.synchronize()
    assert abs(n_is - want_f["n_isects"]) <= max(8, int(2e-6 * n_is)), (n_is, want_f["n_isects"])  # ceil() borderlines
    ref_r, ref_a = torch.from_numpy(want_f["render"]), torch.from_numpy(want_f["alphas"])[..., None]
    ok = agreeing_pixels(render, alphas, ref_r, ref_a)  # the gradient comparison runs on these (tests/parity.py)
    flipped = 1.0 - ok.double().mean().item()
    assert flipped < max_flipped, f"{tag}: {flipped:.2e} of the pixels disagree with the oracle beyond 1e-4"
    rel = ((render.cpu().double() - ref_r).abs() / (ref_r.abs() + 2e-2)).max(-1).values.reshape(-1)
    qs = torch.quantile(rel[torch.randperm(rel.numel())[:200_000]], torch.tensor([0.5, 0.99], dtype=torch.float64))
    depth_g, depth_o = render[..., 3].cpu().double()[ok], torch.from_numpy(want_f["render"][..., 3])[ok]
    valid = depth_o > 0
    depth_rel = float(((depth_g - depth_o).abs()[valid] / depth_o[valid]).max())
    alpha_abs = float((alphas[..., 0].cpu().double()[ok] - torch.from_numpy(want_f["alphas"])[ok]).abs().max())
    # pose gradient for SEVERAL upstream gradients (white noise on the depth channel, seeds 1..3 by default): the number
    # moves with the noise realisation, so one draw is not a parity statement.  Per seed: HIP against the
    # float64 oracle, HIP against the oracle's float32 build (like for like), and the float32 floor of the
    # configuration (the oracle's float32 build against its float64 build).
    upstreams = v if isinstance(v, (list, tuple)) else [v]
    pose_err = hip_vs_f32 = floor32 = det_err = 0.0
    grads = want = None
    # the same backward without float atomics (fixed summation order): separates "the atomics' order moved the sum" from
    # "the records are rounded" in the pose-gradient figure.  Not for the 5 M scene (memory, time).
    det = None
    if with_deterministic and N <= 1_100_000:
        det = RenderContext(N, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=dev, full_grads=False, deterministic=True,
                            reorder=False)
        det.calibrate(*inp)
    for vk in upstreams:
        vm = vk * ok[..., None]
        want = C.rasterization(*cpu, V, sc["K"], W, H, sh_degree=1, render_mode="RGB+ED", v_render=vm, precision="f64",
                               threads=THREADS)
        ctx.forward(*inp)
        grads = ctx.backward(vm.float().to(dev).contiguous(), torch.zeros(H, W, 1, device=dev), full=True)
        grads = ctx.grads_in_input_order(grads)  # (random-order input: the context placed the Gaussians in tile order)
        torch.cuda.synchronize()
        ctx.check_capacity()
        want32 = C.rasterization(*cpu, V, sc["K"], W, H, sh_degree=1, render_mode="RGB+ED", v_render=vm, precision="f32",
                                 threads=THREADS)
        if det is not None:
            det.forward(*inp)
            gd = det.backward(vm.float().to(dev).contiguous(), torch.zeros(H, W, 1, device=dev), full=False)
            torch.cuda.synchronize()
            det_err = max(det_err, rel_inf(gd["viewmat"][:3], want["v_viewmat"][:3]))
        pose_err = max(pose_err, rel_inf(grads["viewmat"][:3], want["v_viewmat"][:3]))
        hip_vs_f32 = max(hip_vs_f32, rel_inf(grads["viewmat"][:3], want32["v_viewmat"][:3]))
        floor32 = max(floor32, rel_inf(want32["v_viewmat"][:3], want["v_viewmat"][:3]))
    errs = dict(render_rel_median=float(qs[0]), render_rel_p99=float(qs[1]), depth_rel=depth_rel, alpha_abs=alpha_abs,
                v_viewmat=pose_err, v_viewmat_vs_f32_oracle=hip_vs_f32, v_viewmat_f32_oracle_vs_f64=floor32,
                upstream_seeds=float(len(upstreams)))
    if det is not None:
        errs["v_viewmat_deterministic"] = det_err
        del det
    for name in grad_names:  # (of the last upstream)
        a, b = grads[name].cpu().double().numpy().reshape(-1), want["v_" + name].reshape(-1)
        errs["v_" + name] = float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
    report(tag, flipped, **errs)
    assert floor32 < FLOOR32_MAX, f"{tag}: the float32 floor itself is {floor32:.2e}"
    kind = "X" if N > 1_100_000 else ("sigma1" if tag.startswith("R") and "sigma=1.0" in tag else "subpixel")
    assert pose_err < pose_grad_bound(floor32, kind), f"{tag}: pose gradient {pose_err:.2e} (float32 floor {floor32:.2e})"
    for name in grad_names:  # relative L2 over all Gaussians; a splat whose own alpha sits on 1/255 at a nearly opaque
        assert errs["v_" + name] < 5e-3, (tag, name, errs["v_" + name])  # pixel switches without moving the pixel
    return errs


def _depth_upstream(H, W, seed=1):
    """The bench's upstream gradient (same generator, dtype and seed: gsplatloc_amd.synthetic.depth_upstream)."""
    from gsplatloc_amd.synthetic import depth_upstream
    return depth_upstream(H, W, seed).double()


@pytest.mark.parametrize("N,W,H,sigma_px,order,max_flipped", [
    (1_000_000, 1200, 680, 1.0, "random", 5e-3),    # workload R, the bench headline
    (1_000_000, 1200, 680, 0.0, "raster", 5e-3),    # R in the reference's regime (as-coded scales, depth-frame order)
    (5_000_000, 1920, 1080, 1.0, "random", 5e-3),   # workload X of BASELINE.json configs[4]
])
def test_config_R_and_X_render_and_gradients_match_the_c_oracle(N, W, H, sigma_px, order, max_flipped):
    from gsplatloc_amd.synthetic import perturbed_pose, random_scene

    sc = random_scene(N, W, H, sigma_px=sigma_px, order=order)
    V = torch.linalg.inv(perturbed_pose())
    _context_vs_c_oracle(f"{'X' if N > 1_000_000 else 'R'} N={N} {W}x{H} sigma={sigma_px} {order}", sc, V, W, H,
                         [_depth_upstream(H, W, seed=k) for k in (1, 2, 3)], max_flipped)


def _tum_like_frame(W=640, H=480, hole_frac=0.08, seed=3):
    """One Gaussian per pixel of a depth frame, with rectangular patches of invalid (zero) depth as a TUM frame
    has them: depth_to_points leaves those points at the camera origin (/root/reference/src/data/Image.py:29-35,
    geometry.py:138-161) and the near plane culls them."""
    from gsplatloc_amd.synthetic import SH_C0, frame_pair
    from oracle import tracker_oracle as T

    fp = frame_pair(W, H, rot_deg=0.4, trans=0.015, seed=seed)
    depth = fp["depth0"].clone()
    g = torch.Generator().manual_seed(seed)
    n_holes = int(hole_frac * W * H / (24 * 18))
    for _ in range(n_holes):
        x0, y0 = int(torch.randint(0, W - 24, (1,), generator=g)), int(torch.randint(0, H - 18, (1,), generator=g))
        depth[y0:y0 + 18, x0:x0 + 24] = 0.0
    pts = T.depth_to_points(depth, fp["K"])
    scales = torch.zeros_like(pts)
    valid = depth.reshape(-1) > 0
    scales[valid] = T.init_gs_scales(pts[valid], as_coded=True)  # kNN of the valid points (as-coded: ~1e-5 m)
    scales[~valid] = 1e-6
    N = pts.shape[0]
    sh = torch.zeros(N, 4, 3)
    sh[:, 0] = (fp["rgb"] - 0.5) / SH_C0
    sc = dict(means=pts, quats=torch.tensor([1.0, 0, 0, 0]).repeat(N, 1), scales=scales, opacities=torch.ones(N), sh=sh,
              K=fp["K"])
    return sc, fp, int(valid.sum())


def test_config_T_depth_frame_with_invalid_pixels():
    sc, fp, n_valid = _tum_like_frame()
    W, H = 640, 480
    assert sc["means"].shape[0] == W * H and n_valid < W * H
    V = torch.linalg.inv(fp["c2w1"])
    _context_vs_c_oracle(f"T N={W * H} ({n_valid} valid) {W}x{H}", sc, V, W, H,
                         [_depth_upstream(H, W, seed=k) for k in (5, 6, 7)], 5e-3)


def test_config_S_tracker_200_iterations():
    """GraphTracker on config S against the oracle tracker (C rasterizer, float64): the first iterations step by
    step, then the whole 200-iteration run through its invariants and its read-outs."""
    import gsplatloc_amd.my_gsplat as M
    from gsplatloc_amd.graph_tracker import GraphTracker
    from gsplatloc_amd.my_gsplat.geometry import depth_to_points
    from gsplatloc_amd.synthetic import frame_pair
    from oracle import tracker_oracle as T

    dev = torch.device("cuda")
    W, H, iters, first = 640, 480, 200, 25
    fp = frame_pair(W, H, rot_deg=0.3, trans=0.01)
    K = fp["K"].to(dev)
    pts0 = depth_to_points(fp["depth0"].to(dev), K)[::3].contiguous()  # 102 400 Gaussians
    rgb = fp["rgb"].to(dev)[::3].contiguous()
    assert pts0.shape[0] == 102_400
    pts1 = depth_to_points(fp["depth1"].to(dev), K)
    scales = M.init_gs_scales(pts0)
    src = M.compute_depth_gt(pts1, fp["rgb"].to(dev), K[None], torch.eye(4, device=dev)[None], H, W)
    cfg = M.TrackerConfig(max_steps=iters, min_step=100, patience=200)
    gt = GraphTracker(pts0.shape[0], W, H, cfg, device=dev, poll=50)
    gt.load_frame(pts0, rgb, scales, src, fp["c2w0"].to(dev), fp["c2w1"].to(dev), K)
    res = gt.run()
    assert res.steps == iters
    # the same frame through the oracle tracker for the first iterations (same schedule: gamma follows max_steps)
    res_o = T.track_frame(pts0.cpu().double(), scales.cpu().double(), rgb.cpu().double(),
                          src.reshape(1, H, W, 1).cpu().double(), fp["K"].double(), W, H, fp["c2w0"].double(),
                          fp["c2w1"].double(), max_steps=iters, min_step=100, engine="c", threads=THREADS,
                          stop_after=first)
    lg, lo = torch.tensor(res.losses[:first], dtype=torch.float64), torch.tensor(res_o.losses, dtype=torch.float64)
    rel = ((lg - lo).abs() / lo).max().item()
    report("S tracker, first %d iterations" % first, 0.0, loss_rel=rel, loss0_rel=abs(float(lg[0] - lo[0]) / float(lo[0])))
    # iteration 0 has no optimiser history: the loss itself must agree to the image tolerance;
    # later iterations pass through Adam, whose g / sqrt(v) normalisation turns a 1e-4 difference of a small
    # gradient entry into a different step, and the difference compounds over the iterations: 3e-3 over 25
    assert abs(float(lg[0] - lo[0])) < 1e-4 * float(lo[0])
    assert rel < 3e-3, rel
    # invariants of the full run
    losses = torch.tensor(res.losses)
    assert torch.isfinite(losses).all() and float(losses[-20:].mean()) < 0.35 * float(losses[0])
    e0 = M.calculate_translation_error(fp["c2w0"], fp["c2w1"])
    assert res.best_eT < 0.1 * e0, (res.best_eT, e0)
    assert res.best_loss == pytest.approx(float(losses[101:].min()), rel=1e-6)  # min-loss read-out after step 100


def test_config_X_fp16_staged_compositing():
    """BASELINE.json configs[4]: 5 M random Gaussians, 1920x1080, "fp16 compositing" -- the compositing kernels gather
    32-byte half-precision records (centre float32; conic, depth, opacity, colour half), transmittance and every
    accumulator float32.  Checked against the float64 oracle at the precision half records allow
    (11 significant bits: 3e-3 relative, stated here), and against the float32-staged HIP path."""
    from gsplatloc_amd.context import RenderContext
    from gsplatloc_amd.synthetic import perturbed_pose, random_scene
    from oracle import c_oracle as C

    dev = torch.device("cuda")
    N, W, H = 5_000_000, 1920, 1080
    sc = random_scene(N, W, H, sigma_px=1.0, order="random")
    V = torch.linalg.inv(perturbed_pose())
    cpu = [sc[k] for k in ("means", "quats", "scales", "opacities", "sh")]
    inp = tuple(t.to(dev).contiguous() for t in cpu) + (V.to(dev).contiguous(), sc["K"].to(dev).contiguous())
    v = _depth_upstream(H, W)
    out = {}
    for staging in ("fp16", "fp32"):
        ctx = RenderContext(N, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=dev, full_grads=False, staging=staging)
        n_is = ctx.calibrate(*inp)
        render, alphas = ctx.forward(*inp)
        g = ctx.backward(v.float().to(dev).contiguous(), torch.zeros(H, W, 1, device=dev), full=False)
        torch.cuda.synchronize()
        ctx.check_capacity()
        out[staging] = (render.cpu().double(), alphas.cpu().double(), g["viewmat"].cpu().double().clone(), n_is)
        del ctx
    r16, a16, g16, n16 = out["fp16"]
    r32, a32, g32, n32 = out["fp32"]
    assert n16 == n32  # binning reads the float32 records: same lists
    want = C.rasterization(*cpu, V, sc["K"], W, H, sh_degree=1, render_mode="RGB+ED", precision="f64", threads=THREADS)
    ref_r, ref_a = torch.from_numpy(want["render"]), torch.from_numpy(want["alphas"])[..., None]
    bad = 1.0 - agreeing_pixels(r16, a16, ref_r, ref_a, rtol=3e-3, atol=3e-4).double().mean().item()
    drel = ((r16[..., 3] - ref_r[..., 3]).abs() / ref_r[..., 3].abs().clamp(min=1e-3))
    pose = rel_inf(g16[:3], g32[:3])
    report("X fp16-staged", bad, depth_rel_mean=float(drel.mean()), depth_rel_p99=float(torch.quantile(drel.reshape(-1)[::37], 0.99)),
           v_viewmat_vs_fp32_staging=pose)
    assert bad < 5e-3, bad
    assert float(drel.mean()) < 5e-4
    assert pose < 2e-2, pose  # the gradient of a render whose records carry 5e-4 relative rounding


@pytest.mark.parametrize("workload", ["S", "T", "D", "R"])
def test_tracker_loss_pose_gradient_at_config_sizes(workload):
    """north_star's pose-gradient statement as the tracker poses it: d loss / d viewmat of GsplatLoc's depth + Sobel
    loss (/root/reference/src/my_gsplat/gs_trainer_total.py:105-150) on the frames of configs S (102 400 Gaussians) and
    T (307 200, invalid depths), 640x480, and -- round 4 -- at the metric's size: D (816 000 Gaussians of a 1200x680
    depth frame) and R (1 M random Gaussians, 1200x680); HIP (render, fused loss kernel, backward) against the oracle
    (C rasterizer in float64 + the restated loss under autograd).  1e-4 of the largest entry is north_star's figure;
    what is measured, and the float32 floor next to it, is printed and bounded (tests/parity.py)."""
    from gsplatloc_amd._lib import check, current_stream, load_library, ptr
    from gsplatloc_amd.context import RenderContext
    from gsplatloc_amd.synthetic import depth_frame_scene, perturbed_pose, random_scene
    from oracle import tracker_oracle as T

    dev = torch.device("cuda")
    if workload == "R":
        W, H = 1200, 680
        sc = random_scene(1_000_000, W, H, sigma_px=1.0, device=dev)
        sc["viewmat"] = torch.linalg.inv(perturbed_pose()).to(dev).contiguous()
    else:
        W, H = (1200, 680) if workload == "D" else (640, 480)
        sc = depth_frame_scene(W, H, stride=3 if workload == "S" else 1, holes=workload == "T", device=dev)
    N = sc["means"].shape[0]
    inp = (sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["sh"], sc["viewmat"], sc["K"].contiguous())
    # target depth: the same cloud seen from a slightly different pose (any fixed image would do)
    ctx = RenderContext(N, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=dev, full_grads=False)
    ctx.calibrate(*inp)
    V2 = sc["viewmat"].clone()
    V2[0, 3] += 0.004
    render, _ = ctx.forward(*inp[:5], V2.contiguous(), inp[6])
    gt = render[..., 3].clone()
    render, _ = ctx.forward(*inp)
    lib = load_library()
    ws_bytes = lib.gsl_loss_ws_bytes(W, H)
    ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=dev)
    v_render = torch.zeros(H, W, 4, device=dev)
    partials = torch.zeros(lib.gsl_loss_n_partials(W, H, 0, H) * 2, device=dev)
    check(lib.gsl_tracking_loss(ptr(ctx.render), 4, ptr(gt), W, H, 0, H, 0.8, 0.2, ptr(v_render), ptr(partials), None,
                                ptr(ws), ws_bytes, current_stream()), "gsl_tracking_loss")
    got = ctx.backward(v_render, torch.zeros(H, W, 1, device=dev), full=False)["viewmat"].cpu().double()
    loss_g = float(0.8 * partials.view(-1, 2)[:, 0].sum() + 0.2 * partials.view(-1, 2)[:, 1].sum()) / (W * H)
    # oracle: float64 C rasterizer under autograd + the restated loss
    cpu = [sc[k].cpu().double() for k in ("means", "quats", "scales", "opacities", "sh")]
    grad, total = {}, {}
    for precision in ("f64", "f32"):  # f32: the float32 floor of this quantity, from the oracle's own float32 build
        Vo = sc["viewmat"].cpu().double().requires_grad_()
        renders, _ = T._CRasterize.apply(Vo, *cpu, sc["K"].cpu().double(), W, H, THREADS, precision)
        total[precision], _, _ = T.tracking_loss(renders[..., 3:4], gt.cpu().double()[None, ..., None])
        total[precision].backward()
        grad[precision] = Vo.grad[:3].clone()
    err, floor32 = rel_inf(got[:3], grad["f64"]), rel_inf(grad["f32"], grad["f64"])
    report(f"tracker-loss pose gradient, {workload}: {N} Gaussians {W}x{H}", 0.0,
           loss_rel=abs(loss_g - float(total["f64"])) / float(total["f64"]), v_viewmat=err,
           v_viewmat_vs_f32_oracle=rel_inf(got[:3], grad["f32"]), v_viewmat_f32_oracle_vs_f64=floor32)
    assert abs(loss_g - float(total["f64"])) < 1e-4 * float(total["f64"])
    # an L1 loss has sign(d - g) in its gradient: where the rendered and the target depth cross, float32 rounding of d
    # flips the sign of that pixel's term, so the gradient carries a float32 floor well above the loss's own.  The
    # bound follows the measured floor but is capped, and the floor itself is bounded (tests/parity.py).
    assert floor32 < FLOOR32_MAX, floor32
    assert err < pose_grad_bound(floor32, "subpixel"), (err, floor32)


def test_long_tile_list_is_split_over_workgroups_and_matches_the_oracle():
    """VERDICT r2 'missing' 1: the pile.  A TUM-like frame seen from a camera that has moved BACKWARDS: the invalid
    (zero-depth) points, which all sit at the previous camera's origin (/root/reference/src/data/Image.py:29-35,
    my_gsplat/geometry.py:138-161), pass the near plane and land on one spot: ~23 k entries in one tile list.  That list
    is cut into segments composited by separate workgroups (segment transmittances; raster_px.hip / raster_g16.hip).
    Render, alpha and every gradient against the float64 oracle, and against the single-workgroup path of round 2
    (GSLOC_LONG_LISTS=0)."""
    import os

    from gsplatloc_amd.context import RenderContext
    from gsplatloc_amd.synthetic import depth_frame_scene

    dev = torch.device("cuda")
    W, H = 640, 480
    sc = depth_frame_scene(W, H, stride=1, holes=True, device=dev, pile=True)
    N = sc["means"].shape[0]
    inp = (sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["sh"], sc["viewmat"], sc["K"].contiguous())
    v = _depth_upstream(H, W, seed=9).float().to(dev).contiguous()
    va = torch.zeros(H, W, 1, device=dev)
    out = {}
    for mode in ("1", "0"):
        os.environ["GSLOC_LONG_LISTS"] = mode
        try:
            ctx = RenderContext(N, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=dev, full_grads=True)
            ctx.calibrate(*inp)
        finally:
            os.environ.pop("GSLOC_LONG_LISTS", None)
        longest = int((ctx.offs[1:] - ctx.offs[:-1]).max())
        assert longest > 20_000, longest
        assert (ctx.long_min > 0) == (mode == "1")
        render, alphas = ctx.forward(*inp)
        g = ctx.backward(v, va, full=True)
        torch.cuda.synchronize()
        ctx.check_capacity()
        n = int(ctx.n_is.item())
        out[mode] = (render.clone(), alphas.clone(), ctx.last_ids.clone(), g["viewmat"].clone(), g["means"].clone(),
                     ctx.flatten_ids[:n].clone(), ctx.offs.clone())
    (r1, a1, l1, gv1, gm1, ids1, offs1), (r0, a0, l0, gv0, gm0, ids0, offs0) = out["1"], out["0"]
    # the long list is sorted by several workgroups (one segment per wave in registers + merge passes): bit for bit the
    # order of the single-workgroup sort
    assert torch.equal(offs1, offs0) and torch.equal(ids1, ids0)
    # same per-pixel sequence of composited splats; only the association of the transmittance product differs
    assert float((l1 != l0).float().mean()) < 1e-4
    assert float((r1 - r0).abs().max()) < 1e-4 and float((a1 - a0).abs().max()) < 1e-5
    assert rel_inf(gv1[:3], gv0[:3]) < 1e-4, rel_inf(gv1[:3], gv0[:3])
    assert float((gm1 - gm0).norm() / gm0.norm()) < 1e-4
    # and the split path against the oracle, like every other configuration
    cpu = {k: sc[k].cpu() for k in ("means", "quats", "scales", "opacities", "sh", "K")}
    errs = _context_vs_c_oracle(f"pile N={N} 640x480, longest tile list {longest}", cpu, sc["viewmat"].cpu(), W, H,
                                [_depth_upstream(H, W, seed=k) for k in (9, 10, 11)], 5e-3)
    assert errs["v_viewmat"] < 8e-4


def test_several_long_lists_of_different_lengths_match_the_single_workgroup_path():
    """Three piles of different sizes in three tiles (2 600, 5 001 and 9 777 extra points at three spots in front of the
    camera -- lengths that are no multiple of the 128-entry compositing segment or of the 512-key sort run) on top of
    a depth frame: the map of (tile, segment) pairs, the multi-workgroup sort and the segment compositing with several
    long tiles at once.  Lists bit for bit those of the single-workgroup path (GSLOC_LONG_LISTS=0), image and gradients
    to the association of the float32 transmittance product."""
    import os

    from gsplatloc_amd.context import RenderContext
    from gsplatloc_amd.synthetic import depth_frame_scene

    dev = torch.device("cuda")
    W, H = 640, 480
    sc = depth_frame_scene(W, H, stride=2, holes=False, device=dev)
    K = sc["K"]
    fx, fy, cx, cy = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])
    g = torch.Generator().manual_seed(4)
    c2w = torch.linalg.inv(sc["viewmat"].cpu().double())
    extra = []
    for n, (u, v_, z) in ((2600, (100.3, 90.2, 1.7)), (5001, (333.1, 250.9, 2.4)), (9777, (555.5, 401.7, 0.9))):
        du = torch.rand(n, generator=g, dtype=torch.float64) * 2.0 - 1.0
        dv = torch.rand(n, generator=g, dtype=torch.float64) * 2.0 - 1.0
        zz = z + 0.3 * torch.rand(n, generator=g, dtype=torch.float64)
        cam = torch.stack([(u + du - cx) / fx * zz, (v_ + dv - cy) / fy * zz, zz], -1)
        extra.append((cam @ c2w[:3, :3].T + c2w[:3, 3]).float())
    extra = torch.cat(extra).to(dev)
    n_extra = extra.shape[0]
    means = torch.cat([sc["means"], extra]).contiguous()
    N = means.shape[0]
    quats = torch.tensor([1.0, 0, 0, 0], device=dev).repeat(N, 1).contiguous()
    scales = torch.cat([sc["scales"], torch.full((n_extra, 3), 2e-3, device=dev)]).contiguous()
    opac = torch.cat([sc["opacities"], torch.full((n_extra,), 0.02, device=dev)]).contiguous()  # (low: long walks)
    sh = torch.cat([sc["sh"], torch.zeros(n_extra, 4, 3, device=dev)]).contiguous()
    sh[-n_extra:, 0] = torch.rand(n_extra, 3, generator=g).to(dev)
    inp = (means, quats, scales, opac, sh, sc["viewmat"], K.contiguous())
    v = _depth_upstream(H, W, seed=9).float().to(dev).contiguous()
    va = torch.zeros(H, W, 1, device=dev)
    out = {}
    for mode in ("1", "0"):
        os.environ["GSLOC_LONG_LISTS"] = mode
        try:
            ctx = RenderContext(N, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=dev, full_grads=True)
            ctx.calibrate(*inp)
        finally:
            os.environ.pop("GSLOC_LONG_LISTS", None)
        sizes = ctx.offs[1:] - ctx.offs[:-1]
        assert (ctx.long_min > 0) == (mode == "1")
        if mode == "1":
            assert int((sizes > ctx.long_min).sum()) >= 3, sizes.sort(descending=True).values[:6]
        for _ in range(2):
            render, alphas = ctx.forward(*inp)
            gr = ctx.backward(v, va, full=True)
        torch.cuda.synchronize()
        ctx.check_capacity()
        n = int(ctx.n_is.item())
        out[mode] = (render.clone(), alphas.clone(), gr["viewmat"].clone(), gr["means"].clone(),
                     ctx.flatten_ids[:n].clone(), ctx.offs.clone())
    (r1, a1, gv1, gm1, ids1, offs1), (r0, a0, gv0, gm0, ids0, offs0) = out["1"], out["0"]
    assert torch.equal(offs1, offs0) and torch.equal(ids1, ids0)
    assert float((r1 - r0).abs().max()) < 1e-4 and float((a1 - a0).abs().max()) < 1e-5
    assert rel_inf(gv1[:3], gv0[:3]) < 2e-4, rel_inf(gv1[:3], gv0[:3])
    assert float((gm1 - gm0).norm() / gm0.norm()) < 2e-4


def test_long_list_workspace_overflow_is_flagged_not_overrun():
    """A frame that needs more (tile, segment) pairs than the long-list workspace holds (a pile that jumps onto a tile
    corner appears in up to four lists at once): tiles whose segments do not fit are left out as a whole and the need is
    reported -- nothing is indexed past the workspace (round 3: the first GraphTracker run on a pile frame faulted
    exactly there) -- and after grow_long() the result is the one of a context that had room from the start."""
    from gsplatloc_amd.context import RenderContext
    from gsplatloc_amd.synthetic import depth_frame_scene

    dev = torch.device("cuda")
    W, H = 640, 480
    sc = depth_frame_scene(W, H, stride=1, holes=True, device=dev, pile=True)
    N = sc["means"].shape[0]
    inp = (sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["sh"], sc["viewmat"], sc["K"].contiguous())
    v = _depth_upstream(H, W, seed=9).float().to(dev).contiguous()
    va = torch.zeros(H, W, 1, device=dev)
    ref = RenderContext(N, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=dev, full_grads=True)
    ref.calibrate(*inp)
    assert ref.long_min > 0
    r_ref, _ = ref.forward(*inp)
    g_ref = ref.backward(v, va, full=True)["viewmat"].clone()
    r_ref = r_ref.clone()
    ctx = RenderContext(N, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=dev, full_grads=True)
    ctx.calibrate(*inp)
    need = int(ctx.long_ws[:16].view(torch.int32)[0].item()) if False else None  # (n_seg is written by the forward)
    # shrink the workspace below the pile's 46 segments
    ctx.max_seg = 20
    ctx.long_ws_bytes = ctx.lib.gsl_long_ws_bytes(ctx.max_seg)
    ctx.long_ws = torch.zeros(ctx.long_ws_bytes, dtype=torch.uint8, device=dev)
    guard = torch.full((1 << 20,), 0x5A, dtype=torch.uint8, device=dev)  # (allocated right after: a canary, not a proof)
    ctx.forward(*inp)
    ctx.backward(v, va, full=True)
    torch.cuda.synchronize()
    assert bool((guard == 0x5A).all())
    need = ctx.long_overflowed()
    assert need >= 46, need
    with pytest.raises(RuntimeError, match="long tile lists need"):
        ctx.check_capacity()
    ctx.grow_long(need)
    r, _ = ctx.forward(*inp)
    g = ctx.backward(v, va, full=True)["viewmat"]
    torch.cuda.synchronize()
    assert ctx.long_overflowed() == 0
    assert torch.equal(r, r_ref) and rel_inf(g[:3], g_ref[:3]) < 1e-5


@pytest.mark.parametrize("mode", ["ED", "RGB", "D"])
def test_long_list_split_in_the_other_render_modes(mode):
    """The pile frame in the one-channel and colour-only modes (other template instances of the long-list kernels and
    of the per-quadrant backward): split over workgroups against the single-workgroup walk."""
    import os

    from gsplatloc_amd.context import RenderContext
    from gsplatloc_amd.synthetic import depth_frame_scene

    dev = torch.device("cuda")
    W, H = 640, 480
    sc = depth_frame_scene(W, H, stride=1, holes=True, device=dev, pile=True)
    N = sc["means"].shape[0]
    inp = (sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["sh"], sc["viewmat"], sc["K"].contiguous())
    D = {"ED": 1, "D": 1, "RGB": 3}[mode]
    g = torch.Generator().manual_seed(4)
    v = torch.randn(H, W, D, generator=g).to(dev)
    va = torch.randn(H, W, 1, generator=g).to(dev)
    out = {}
    for flag in ("1", "0"):
        os.environ["GSLOC_LONG_LISTS"] = flag
        try:
            ctx = RenderContext(N, W, H, mode, sh_degree=1, K_sh=4, device=dev, full_grads=True)
            ctx.calibrate(*inp)
        finally:
            os.environ.pop("GSLOC_LONG_LISTS", None)
        assert (ctx.long_min > 0) == (flag == "1")
        # the general backward in both (the tiny-splat backward does not take colour-only upstream differently, but the
        # point here is the long-list variant of the per-quadrant kernel)
        ctx.use_general_backward()
        render, alphas = ctx.forward(*inp)
        gr = ctx.backward(v, va, full=True)
        torch.cuda.synchronize()
        ctx.check_capacity()
        out[flag] = (render.clone(), alphas.clone(), gr["viewmat"].clone(), gr["means"].clone())
    (r1, a1, gv1, gm1), (r0, a0, gv0, gm0) = out["1"], out["0"]
    assert float((r1 - r0).abs().max()) < 2e-4 * max(1.0, float(r0.abs().max())) and float((a1 - a0).abs().max()) < 1e-5
    assert rel_inf(gv1[:3], gv0[:3]) < 2e-4, rel_inf(gv1[:3], gv0[:3])
    assert float((gm1 - gm0).norm() / gm0.norm()) < 2e-4
