"""Host-side mirror of the reference's model.py on the CPU: the pose module against an independent
restatement of the reference's arithmetic (two separate Adam optimisers, kornia formulas from the oracle), and
GSModel's wiring with the oracle rasterizer substituted for the HIP one (tests may use the oracle)."""
import pytest
import torch

from gsplatloc_amd.my_gsplat import model as M
from oracle import gsplat_oracle as G
from oracle import tracker_oracle as TO


def _pose(seed=0):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(4, generator=g)
    R = TO.quaternion_to_rotation_matrix(q / q.norm())
    T = torch.eye(4)
    T[:3, :3] = R
    T[:3, 3] = torch.randn(3, generator=g)
    return T


def test_pose_module_matches_two_adam_restatement():
    """model.py:27-116 as the reference codes it: one Adam per parameter, lr/weight decay per CameraConfig."""
    cfg = M.CameraConfig()
    T0 = _pose(1)
    mod = M.CameraOptModule_quat_tans(T0.clone(), config=cfg)
    q = TO.rotation_matrix_to_quaternion(T0[:3, :3].contiguous()).clone().requires_grad_()
    t = T0[:3, 3].clone().requires_grad_()
    opts = [torch.optim.Adam([q], lr=cfg.quat_lr, weight_decay=cfg.quat_opt_reg),
            torch.optim.Adam([t], lr=cfg.trans_lr, weight_decay=cfg.trans_opt_reg)]
    gamma = 0.2 ** (1 / 40)
    sched_a = [torch.optim.lr_scheduler.ExponentialLR(o, gamma) for o in mod.optimizers]
    sched_b = [torch.optim.lr_scheduler.ExponentialLR(o, gamma) for o in opts]
    g = torch.Generator().manual_seed(2)
    for _ in range(40):
        w = torch.randn(4, 4, generator=g)
        mod.optimizer_clean()
        (mod() * w).sum().backward()
        mod.optimizer_step()
        for o in opts:
            o.zero_grad(set_to_none=True)
        c2w = torch.eye(4)
        c2w = torch.cat([torch.cat([TO.quaternion_to_rotation_matrix(q / q.norm()), t[:, None]], 1), c2w[3:]], 0)
        (c2w * w).sum().backward()
        for o in opts:
            o.step()
        for s in sched_a + sched_b:
            s.step()
    torch.testing.assert_close(mod.quat_cur.detach(), q.detach(), rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(mod.t_cur.detach(), t.detach(), rtol=1e-6, atol=1e-7)
    assert mod().shape == (4, 4) and torch.equal(mod()[3], torch.tensor([0.0, 0, 0, 1]))


def test_pose_module_optimizers_index_like_the_reference_trainer():
    """gs_trainer_total.py:65-72 builds one ExponentialLR over camera_opt.optimizers[0] and one over
    camera_opt.optimizers[1]: the module must expose two Adam objects, quaternion first (model.py:93-116)."""
    cfg = M.CameraConfig()
    mod = M.CameraOptModule_quat_tans(_pose(5), config=cfg)
    assert len(mod.optimizers) == 2
    schedulers = [torch.optim.lr_scheduler.ExponentialLR(mod.optimizers[0], gamma=0.2 ** (1.0 / 200)),
                  torch.optim.lr_scheduler.ExponentialLR(mod.optimizers[1], gamma=0.2 ** (1.0 / 200))]
    q_opt, t_opt = mod.optimizers
    assert q_opt.param_groups[0]["params"][0] is mod.quat_cur and t_opt.param_groups[0]["params"][0] is mod.t_cur
    assert q_opt.param_groups[0]["lr"] == cfg.quat_lr and q_opt.param_groups[0]["weight_decay"] == cfg.quat_opt_reg
    assert t_opt.param_groups[0]["lr"] == cfg.trans_lr and t_opt.param_groups[0]["weight_decay"] == cfg.trans_opt_reg
    assert q_opt.param_groups[0]["name"] == "quat" and t_opt.param_groups[0]["name"] == "trans"
    mod.optimizer_clean()
    mod().sum().backward()
    mod.optimizer_step()
    for s in schedulers:
        s.step()
    assert abs(q_opt.param_groups[0]["lr"] - cfg.quat_lr * 0.2 ** (1.0 / 200)) < 1e-12


def test_pose_module_update_and_prediction():
    T0, T1 = _pose(3), _pose(4)
    mod = M.CameraOptModule_quat_tans(T0.clone())
    torch.testing.assert_close(mod(), T0, rtol=1e-5, atol=1e-6)
    q0, t0 = mod.quat_cur.detach().clone(), mod.t_cur.detach().clone()
    with torch.no_grad():
        mod.quat_cur += 0.01
        mod.t_cur += torch.tensor([0.1, 0.0, -0.2])
    q1, t1 = mod.quat_cur.detach().clone(), mod.t_cur.detach().clone()
    mod.update_pose()  # constant velocity from the pose at construction
    want_q = q1 + (q1 - q0)
    torch.testing.assert_close(mod.quat_cur.detach(), want_q / want_q.norm())
    torch.testing.assert_close(mod.t_cur.detach(), t1 + (t1 - t0))
    assert torch.equal(mod.prev_quat, q1) and torch.equal(mod.prev_t, t1)
    old = mod.optimizers[0]
    mod.update_pose(T1)
    torch.testing.assert_close(mod(), T1, rtol=1e-5, atol=1e-6)
    assert mod.optimizers[0] is not old and len(mod.optimizers[0].state) == 0
    with pytest.raises(ValueError, match="fake new pose"):
        mod.update_pose(torch.eye(3))


def test_gsmodel_renders_through_the_rasterization_signature(monkeypatch):
    calls = []

    def oracle_rasterization(**kw):
        calls.append(kw)
        return G.rasterization(kw["means"], kw["quats"], kw["scales"], kw["opacities"], kw["colors"], kw["viewmats"],
                               kw["Ks"], kw["width"], kw["height"], sh_degree=kw["sh_degree"],
                               near_plane=kw["near_plane"], far_plane=kw["far_plane"], render_mode=kw["render_mode"])

    monkeypatch.setattr(M, "rasterization", oracle_rasterization)
    g = torch.Generator().manual_seed(5)
    n, W, H = 300, 48, 32
    pts = torch.stack([torch.rand(n, generator=g) * 2 - 1, torch.rand(n, generator=g) * 1.2 - 0.6,
                       torch.rand(n, generator=g) * 2 + 2], 1)
    rgb = torch.rand(n, 3, generator=g)
    scales = torch.full((n, 3), 0.03)
    K = torch.tensor([[40.0, 0, 23.5], [0, 40.0, 15.5], [0, 0, 1]])
    model = M.GSModel(pts, rgb, scales=scales)
    assert len(model) == n and model.sh0.shape == (n, 1, 3) and model.shN.shape == (n, 3, 3)
    assert torch.equal(model.quats, torch.tensor([1.0, 0, 0, 0]).repeat(n, 1))
    c2w = torch.eye(4)[None]
    colors, alphas, _ = model(c2w, K[None], W, H)
    assert colors.shape == (1, H, W, 4) and alphas.shape == (1, H, W, 1)
    kw = calls[-1]
    assert kw["render_mode"] == "RGB+ED" and kw["rasterize_mode"] == "classic" and kw["packed"] is False
    assert torch.all(kw["opacities"] == 1.0)  # sigmoid(logit(1.0))
    torch.testing.assert_close(kw["colors"][:, 0], (rgb - 0.5) / 0.28209479177387814)
    assert float(kw["colors"][:, 1:].abs().max()) == 0.0
    # the frame constants are built once and rebuilt when an input is modified or replaced
    first = calls[-1]["opacities"]
    model(c2w, K[None], W, H, render_mode="ED")
    assert calls[-1]["opacities"] is first and calls[-1]["render_mode"] == "ED"
    model.opacities.fill_(0.0)
    model(c2w, K[None], W, H)
    assert calls[-1]["opacities"] is not first and torch.all(calls[-1]["opacities"] == 0.5)
    model.sh0 = model.sh0 * 0.5
    model(c2w, K[None], W, H)
    torch.testing.assert_close(calls[-1]["colors"][:, 0], 0.5 * (rgb - 0.5) / 0.28209479177387814)


def test_loss_mirror_matches_the_oracle_restatement():
    from gsplatloc_amd.my_gsplat import loss as L

    g = torch.Generator().manual_seed(6)
    a = torch.rand(1, 37, 53, 1, generator=g, dtype=torch.float64) * 3 + 1
    b = a + 0.1 * torch.randn(1, 37, 53, 1, generator=g, dtype=torch.float64)
    for kind in ("l1", "mse"):
        assert float(L.compute_depth_loss(a, b, loss_type=kind)) == pytest.approx(
            float((a - b).abs().mean() if kind == "l1" else ((a - b) ** 2).mean()), rel=1e-12)
    assert float(L.compute_silhouette_loss(a, b)) == pytest.approx(float(TO.compute_silhouette_loss(a, b)), rel=1e-12)
    x = torch.rand(2, 3, 9, 11, generator=g, dtype=torch.float64)
    torch.testing.assert_close(L.sobel(x), TO.sobel(x), rtol=1e-12, atol=0)
    # hand-checked value: a unit step edge, normalised Sobel responds with 4/8 on both sides of the step
    step = torch.zeros(1, 1, 5, 6, dtype=torch.float64)
    step[..., 3:] = 1.0
    e = L.sobel(step, eps=0.0)
    assert torch.allclose(e[0, 0, :, 2:4], torch.full((5, 2), 0.5, dtype=torch.float64)) and float(e[0, 0, :, 0].max()) == 0.0
    with pytest.raises(ValueError, match="Use 'mse' or 'l1'"):
        L.compute_depth_loss(a, b, loss_type="huber")
    with pytest.raises(ValueError, match="or 'huber'"):
        L.compute_silhouette_loss(a, b, loss_type="huber")


def test_normal_consistency_term_mirror_oracle_and_tracker_weighting():
    """The term the reference defines and keeps switched off (loss.py:62-101, gs_trainer_total.py:138-143): the host
    mirror, the oracle restatement and the weighting inside tracking_loss agree; as coded, the cosine runs along
    each image row per component (dim=1 of [H,W,3])."""
    import torch.nn.functional as F
    from gsplatloc_amd.my_gsplat import loss as L
    from gsplatloc_amd.my_gsplat.trainer import PoseTracker, TrackerConfig
    from gsplatloc_amd.synthetic import replica_intrinsics

    g = torch.Generator().manual_seed(8)
    H, W = 23, 31
    K = replica_intrinsics(W, H, dtype=torch.float64)
    a = torch.rand(H, W, generator=g, dtype=torch.float64) * 2 + 1
    b = a + 0.05 * torch.randn(H, W, generator=g, dtype=torch.float64)
    a[4:7, 9:15] = 0.0
    na, no = L.depth_to_normal(a, K), TO.depth_to_normal(a, K)
    torch.testing.assert_close(na, no, rtol=1e-12, atol=1e-14)
    assert torch.allclose(na.norm(dim=-1)[a > 0][:50], torch.ones(50, dtype=torch.float64))
    # a fronto-parallel plane: dx x dy with x to the right and y down points along +z, as coded
    flat = TO.depth_to_normal(torch.full((5, 6), 2.0, dtype=torch.float64), K)
    assert torch.allclose(flat, torch.tensor([0.0, 0.0, 1.0], dtype=torch.float64).expand(5, 6, 3), atol=1e-12)
    want = 1 - F.cosine_similarity(no, TO.depth_to_normal(b, K), dim=1).mean()
    assert float(L.compute_normal_consistency_loss(a, b, K=K)) == pytest.approx(float(want), rel=1e-12)
    assert float(TO.compute_normal_consistency_loss(a, b, K)) == pytest.approx(float(want), rel=1e-12)
    assert want.shape == () and F.cosine_similarity(no, no, dim=1).shape == (H, 3)
    d, gt = a[None, ..., None].clone().requires_grad_(), b[None, ..., None]
    cfg = TrackerConfig(depth_lambda=0.7, normal_lambda=0.1)
    total, dl, sl = PoseTracker(cfg).tracking_loss(d, gt, K)
    t_o, dl_o, sl_o = TO.tracking_loss(d, gt, 0.7, 0.1, K)
    m = (a != 0).double()
    nl = TO.compute_normal_consistency_loss(a * m, b * m, K)
    assert float(total.detach()) == pytest.approx(float((0.7 * dl_o + 0.2 * sl_o + 0.1 * nl).detach()), rel=1e-12)
    assert float(total.detach()) == pytest.approx(float(t_o.detach()), rel=1e-12)
    total.backward()
    assert torch.isfinite(d.grad).all() and float(d.grad.abs().max()) > 0


def test_geometry_mirror_on_cpu(monkeypatch):
    from gsplatloc_amd.my_gsplat import geometry as Geo

    def oracle_rasterization(**kw):
        return G.rasterization(kw["means"], kw["quats"], kw["scales"], kw["opacities"], kw["colors"], kw["viewmats"],
                               kw["Ks"], kw["width"], kw["height"], sh_degree=kw["sh_degree"],
                               near_plane=kw["near_plane"], far_plane=kw["far_plane"], render_mode=kw["render_mode"])

    monkeypatch.setattr(Geo, "rasterization", oracle_rasterization)
    g = torch.Generator().manual_seed(7)
    H, W = 30, 44
    K = torch.tensor([[45.0, 0, 21.5], [0, 45.0, 14.5], [0, 0, 1]])
    depth = torch.rand(H, W, generator=g) * 0.3 + 2.0
    pts = Geo.depth_to_points(depth, K)
    torch.testing.assert_close(pts, TO.depth_to_points(depth, K))
    assert Geo.depth_to_points(depth, K, include_homogeneous=True).shape == (H * W, 4)
    # pixel (u, v) = (3, 2): index 2 * W + 3
    z = depth[2, 3]
    torch.testing.assert_close(pts[2 * W + 3], torch.stack([(3 - 21.5) / 45 * z, (2 - 14.5) / 45 * z, z]))
    scales = Geo.init_gs_scales(pts)
    torch.testing.assert_close(scales, TO.init_gs_scales(pts, as_coded=True))
    assert scales.shape == (H * W, 3) and torch.equal(scales[:, 0], scales[:, 2])
    # a fronto-parallel plane: normals are -z ... +z up to the sign convention of cross(dx, dy) = +z
    flat = Geo.depth_to_normal(torch.full((H, W), 2.5), K)
    torch.testing.assert_close(flat, torch.tensor([0.0, 0.0, 1.0]).expand(H, W, 3))
    # the target depth of the tracker: ED render of the cloud from its own camera reproduces the depth image
    rgb = torch.rand(H * W, 3, generator=g)
    vv, uu = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    ramp = 2.0 + 0.002 * uu + 0.003 * vv  # a tilted plane
    ed = Geo.compute_depth_gt(Geo.depth_to_points(ramp, K), rgb, K[None], torch.eye(4)[None], H, W)
    assert ed.shape == (H, W)
    inner = (slice(2, H - 2), slice(2, W - 2))
    assert float((ed[inner] - ramp[inner]).abs().max()) < 5e-3  # blurred over ~1 px of a 0.003/px slope
    R = TO.quaternion_to_rotation_matrix(torch.tensor([0.9, 0.1, -0.3, 0.2]) / torch.tensor([0.9, 0.1, -0.3, 0.2]).norm())
    T = Geo.construct_full_pose(R, torch.tensor([0.1, 0.2, 0.3]))
    torch.testing.assert_close(Geo.transform_points(T, pts), pts @ R.T + torch.tensor([0.1, 0.2, 0.3]))


def test_pose_tracker_host_loop_matches_the_oracle_tracker(monkeypatch):
    """PoseTracker(engine="autograd") -- the host side of Runner.train's per-frame body -- with the oracle
    rasterizer substituted, against the oracle's own tracker loop: same loss trajectory, early-stop
    bookkeeping and final pose (float32 on both sides)."""
    from gsplatloc_amd.my_gsplat import geometry as Geo
    from gsplatloc_amd.my_gsplat import trainer as Tr

    def oracle_rasterization(**kw):
        return G.rasterization(kw["means"], kw["quats"], kw["scales"], kw["opacities"], kw["colors"], kw["viewmats"],
                               kw["Ks"], kw["width"], kw["height"], sh_degree=kw["sh_degree"],
                               near_plane=kw["near_plane"], far_plane=kw["far_plane"], render_mode=kw["render_mode"])

    monkeypatch.setattr(M, "rasterization", oracle_rasterization)
    monkeypatch.setattr(Geo, "rasterization", oracle_rasterization)
    H, W = 24, 32
    K = torch.tensor([[30.0, 0, 15.5], [0, 30.0, 11.5], [0, 0, 1]])
    vv, uu = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    depth = 2.0 + 0.01 * uu + 0.02 * vv + 0.15 * torch.sin(uu / 3.0) * torch.cos(vv / 4.0)
    pts = Geo.depth_to_points(depth, K)
    g = torch.Generator().manual_seed(8)
    rgb = torch.rand(H * W, 3, generator=g)
    scales = torch.full((H * W, 3), 0.04)
    gt_c2w = torch.eye(4)
    target = Geo.compute_depth_gt(pts, rgb, K[None], gt_c2w[None], H, W).reshape(1, H, W, 1)
    init = torch.eye(4)
    init[:3, :3] = TO.quaternion_to_rotation_matrix(torch.tensor([1.0, 0.004, -0.003, 0.002]) / torch.tensor([1.0, 0.004, -0.003, 0.002]).norm())
    init[:3, 3] = torch.tensor([0.01, -0.008, 0.006])
    steps = 12
    cfg = Tr.TrackerConfig(max_steps=steps, min_step=3, patience=4)
    mine = Tr.PoseTracker(cfg, engine="autograd").track_frame(pts, rgb, target, init.clone(), gt_c2w, K, W, H,
                                                             scales=scales)
    ref = TO.track_frame(pts, scales, rgb, target, K, W, H, init.clone(), gt_c2w, max_steps=steps, patience=4,
                         min_step=3)
    assert mine.steps == ref.steps and len(mine.losses) == len(ref.losses)
    torch.testing.assert_close(torch.tensor(mine.losses), torch.tensor(ref.losses), rtol=2e-5, atol=1e-8)
    assert mine.best_loss == pytest.approx(ref.best_loss, rel=2e-5)
    assert mine.best_eT == pytest.approx(ref.best_eT, rel=1e-3, abs=1e-7)
    torch.testing.assert_close(mine.final_c2w, ref.final_c2w, rtol=1e-5, atol=1e-6)
    assert mine.losses[-1] < mine.losses[0]
