"""The C restatement (oracle/csrc/gsplat_oracle.c) against the autograd oracle (oracle/gsplat_oracle.py) in
float64: same images, same intersection count, and its HAND-DERIVED backward against gradients nobody derived
by hand -- every input and the view matrix, all render modes, SH degree 0-3, anisotropic and antialiased.
Also the float32 build (the CPU baseline of bench.py) against the float64 one."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as C
from oracle import gsplat_oracle as G
from tests.scenes import random_scene, sh_from_rgb, small_pose


@pytest.fixture(scope="module", autouse=True)
def _built():
    C.build()


def _scene(N, W, H, sh_degree, aniso, opacity, sigma_px, seed):
    sc = random_scene(N, W, H, seed=seed, sigma_px=sigma_px, aniso=aniso, opacity=opacity)
    g = torch.Generator().manual_seed(seed + 1)
    if sh_degree is None:
        colors = sc["rgbs"]
    else:
        K = (sh_degree + 1) ** 2
        colors = torch.zeros(N, K, 3, dtype=torch.float64)
        colors[:, 0] = sh_from_rgb(sc["rgbs"])[:, 0] if sh_from_rgb(sc["rgbs"]).dim() == 3 else sh_from_rgb(sc["rgbs"])
        if K > 1:
            colors[:, 1:] = 0.3 * torch.randn(N, K - 1, 3, generator=g, dtype=torch.float64)
    V = torch.linalg.inv(small_pose(1.5, 0.05, seed=seed))
    return sc, colors, V


def _autograd(sc, colors, V, W, H, sh_degree, mode, aa, v_render, v_alphas):
    ins = [sc[k].clone().requires_grad_() for k in ("means", "quats", "scales", "opacities")]
    col = colors.clone().requires_grad_()
    Vg = V.clone().requires_grad_()
    rc, ra, meta = G.rasterization(*ins, col, Vg[None], sc["K"][None], W, H, sh_degree=sh_degree, render_mode=mode,
                                   rasterize_mode="antialiased" if aa else "classic")
    ((rc[0] * v_render).sum() + (ra[0, ..., 0] * v_alphas).sum()).backward()
    return rc[0].detach(), ra[0, ..., 0].detach(), [t.grad for t in ins], col.grad, Vg.grad, meta


@pytest.mark.parametrize("mode,sh_degree,aniso,aa,opacity,sigma_px", [
    ("RGB+ED", 1, False, False, None, 1.0),          # the reference's call
    ("RGB+ED", 3, True, False, (0.2, 0.95), 2.0),
    ("ED", None, False, False, None, 0.0),           # geometry.py:117-132, as-coded tiny splats
    ("RGB", None, True, True, (0.3, 1.0), 1.5),      # direct colours, antialiased
    ("RGB+D", 2, True, True, (0.1, 0.9), 3.0),
    ("D", None, False, False, (0.5, 1.0), 1.0),
    ("RGB+ED", 0, False, False, (0.4, 1.0), 1.2),
])
def test_c_oracle_matches_autograd_oracle(mode, sh_degree, aniso, aa, opacity, sigma_px):
    N, W, H = 400, 57, 41  # not multiples of the tile size
    use_sh = sh_degree if mode.startswith("RGB") else None
    sc, colors, V = _scene(N, W, H, use_sh, aniso, opacity, sigma_px, seed=11)
    if not mode.startswith("RGB"):
        colors = torch.zeros(N, 3, dtype=torch.float64)  # ignored by depth-only modes
    D = (3 if mode.startswith("RGB") else 0) + (1 if mode not in ("RGB",) else 0)
    g = torch.Generator().manual_seed(5)
    v_render = torch.randn(H, W, D, generator=g, dtype=torch.float64)
    v_alphas = torch.randn(H, W, generator=g, dtype=torch.float64)
    rc, ra, gin, gcol, gV, meta = _autograd(sc, colors, V, W, H, use_sh, mode, aa, v_render, v_alphas)
    out = C.rasterization(sc["means"], sc["quats"], sc["scales"], sc["opacities"], colors, V, sc["K"], W, H,
                          sh_degree=use_sh, render_mode=mode, v_render=v_render, v_alphas=v_alphas, antialiased=aa,
                          threads=4)
    assert out["n_isects"] == meta["flatten_ids"].numel() and out["n_isects"] > N // 2
    np.testing.assert_allclose(out["render"], rc.numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(out["alphas"], ra.numpy(), rtol=1e-10, atol=1e-12)

    def close(a, b, what):
        b = b.numpy()
        scale = max(np.abs(b).max(), 1e-30)
        assert np.abs(a - b).max() <= 1e-9 * scale, (what, np.abs(a - b).max(), scale)

    close(out["v_means"], gin[0], "means")
    close(out["v_quats"], gin[1], "quats")
    close(out["v_scales"], gin[2], "scales")
    close(out["v_opacities"], gin[3], "opacities")
    close(out["v_viewmat"][:3], gV[:3], "viewmat")
    if mode.startswith("RGB"):
        close(out["v_colors"], gcol, "colors")


def test_stage_functions_and_binning_order():
    """Intersection ids in the oracle's order (tile, float32 depth bits, Gaussian id), duplicates included."""
    import ctypes

    lib = C.load("f64")
    N, W, H = 300, 64, 48
    sc, _, V = _scene(N, W, H, None, False, None, 2.0, seed=3)
    sc["means"][10:14] = sc["means"][10]  # identical depth: order by id
    radii, m2, dep, con, _ = G.fully_fused_projection(sc["means"], sc["quats"], sc["scales"], V[None], sc["K"][None], W, H)
    tpg, ids, fids = G.isect_tiles(m2, radii, dep, 16, 4, 3)
    offs = G.isect_offset_encode(ids, 1, 4, 3)
    m2n, rn, dn = (np.ascontiguousarray(t[0].numpy()) for t in (m2, radii, dep))
    tp = np.zeros(N, np.int32)
    total = lib.gso_isect(m2n.ctypes.data, rn.ctypes.data, dn.ctypes.data, N, 16, 4, 3, tp.ctypes.data, 0, None, None, None)
    assert total == fids.numel() and (tp == tpg[0].numpy()).all()
    my_ids, my_f, my_o = np.zeros(total, np.int64), np.zeros(total, np.int32), np.zeros(12, np.int32)
    assert lib.gso_isect(m2n.ctypes.data, rn.ctypes.data, dn.ctypes.data, N, 16, 4, 3, None, total, my_ids.ctypes.data,
                         my_f.ctypes.data, my_o.ctypes.data) == total
    assert (my_f == fids.numpy()).all() and (my_ids == ids.numpy()).all() and (my_o == offs.reshape(-1).numpy()).all()
    # the projection stage on its own, culled entries zeroed
    r2, mm, dd, cc = np.zeros(N, np.int32), np.zeros((N, 2)), np.zeros(N), np.zeros((N, 3))
    a = [np.ascontiguousarray(sc[k].numpy()) for k in ("means", "quats", "scales")]
    Vn, Kn = np.ascontiguousarray(V.numpy()), np.ascontiguousarray(sc["K"].numpy())
    assert lib.gso_project_fwd(*(x.ctypes.data for x in a), Vn.ctypes.data, Kn.ctypes.data, N, W, H, 0.3, 0.01, 1e10, 0.0,
                               r2.ctypes.data, mm.ctypes.data, dd.ctypes.data, cc.ctypes.data, None) == 0
    assert (r2 == rn).all()
    np.testing.assert_allclose(mm, m2n, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(cc, con[0].numpy(), rtol=1e-11, atol=1e-12)
    assert lib.gso_project_fwd(None, None, None, Vn.ctypes.data, Kn.ctypes.data, 5, W, H, 0.3, 0.01, 1e10, 0.0, None, None,
                               None, None, None) == -1
    del ctypes


def test_float32_build_tracks_the_float64_build():
    N, W, H = 3000, 160, 120
    sc, colors, V = _scene(N, W, H, 1, False, None, 1.0, seed=21)
    g = torch.Generator().manual_seed(9)
    v_render = torch.zeros(H, W, 4, dtype=torch.float64)
    v_render[..., 3] = torch.randn(H, W, generator=g, dtype=torch.float64)
    kw = dict(sh_degree=1, render_mode="RGB+ED", v_render=v_render)
    args = (sc["means"], sc["quats"], sc["scales"], sc["opacities"], colors, V, sc["K"], W, H)
    hi = C.rasterization(*args, precision="f64", threads=4, **kw)
    lo = C.rasterization(*args, precision="f32", threads=4, **kw)
    assert lo["render"].dtype == np.float32 and abs(lo["n_isects"] - hi["n_isects"]) <= 2
    bad = np.abs(lo["render"] - hi["render"]) > 1e-5 + 1e-4 * np.abs(hi["render"])
    assert bad.mean() < 3e-3  # threshold-sitting pixels (alpha < 1/255, T <= 1e-4) flip between precisions
    gh, gl = hi["v_viewmat"][:3], lo["v_viewmat"][:3]
    assert np.abs(gl - gh).max() < 2e-3 * np.abs(gh).max()


@pytest.mark.parametrize("seed", range(8))
def test_binning_agrees_on_adversarial_inputs(seed):
    """Tile rectangles, counts, keys and offsets of the two oracles on inputs chosen to sit on the clamps: centres
    outside the image on every side, radii from 1 px to larger than the image, exact tile boundaries, equal depths."""
    lib = C.load("f64")
    g = torch.Generator().manual_seed(100 + seed)
    N, W, H, ts = 400, 150 + 7 * seed, 90 + 5 * seed, 16
    tw, th = (W + ts - 1) // ts, (H + ts - 1) // ts
    m2 = torch.stack([torch.rand(N, generator=g, dtype=torch.float64) * (W + 120) - 60,
                      torch.rand(N, generator=g, dtype=torch.float64) * (H + 120) - 60], -1)
    m2[:40] = torch.round(m2[:40] / ts) * ts                      # centres exactly on tile boundaries
    radii = torch.randint(1, 40, (N,), generator=g, dtype=torch.int32)
    radii[40:50] = 400                                            # wider than the image
    radii[50:70] = 0                                              # culled
    dep = torch.rand(N, generator=g, dtype=torch.float64) * 5 + 0.5
    dep[70:90] = dep[70]                                          # ties resolve by Gaussian id
    tpg, ids, fids = G.isect_tiles(m2[None], radii[None], dep[None], ts, tw, th)
    offs = G.isect_offset_encode(ids, 1, tw, th)
    a = [np.ascontiguousarray(x.numpy()) for x in (m2, radii, dep)]
    tp = np.zeros(N, np.int32)
    total = lib.gso_isect(a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, N, ts, tw, th, tp.ctypes.data, 0, None,
                          None, None)
    assert total == fids.numel() and (tp == tpg[0].numpy()).all()
    my_ids, my_f, my_o = np.zeros(total, np.int64), np.zeros(total, np.int32), np.zeros(tw * th, np.int32)
    assert lib.gso_isect(a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, N, ts, tw, th, None, total,
                         my_ids.ctypes.data, my_f.ctypes.data, my_o.ctypes.data) == total
    assert (my_f == fids.numpy()).all() and (my_ids == ids.numpy()).all() and (my_o == offs.reshape(-1).numpy()).all()
    assert lib.gso_isect(a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, N, ts, tw, th, None, total - 1,
                         my_ids.ctypes.data, my_f.ctypes.data, my_o.ctypes.data) == -1  # capacity too small


def test_frustum_clamp_branches_and_culling_agree():
    """Gaussians centred outside 1.3x the field of view but large enough to reach the image exercise the clamped
    branch of the EWA Jacobian and its vjp; others sit behind the camera, beyond the far plane or below
    radius_clip.  Both oracles must cull the same set and agree on every gradient."""
    from tests.scenes import frustum_clamp_scene

    sc = frustum_clamp_scene()
    W, H, V, K, kw = sc["W"], sc["H"], sc["V"], sc["K"], sc["kw"]
    tanx, tany = sc["tan"]
    ins = [sc[k].clone().requires_grad_() for k in ("means", "quats", "scales", "opacities", "rgbs")]
    Vg = V.clone().requires_grad_()
    rc, ra, meta = G.rasterization(*ins, Vg[None], K[None], W, H, render_mode="RGB+D", **kw)
    radii = meta["radii"][0]
    assert (radii[12:] == 0).all() and (radii[:12] > 0).sum() >= 8      # the three special ones are culled
    mc = sc["means"] @ V[:3, :3].T + V[:3, 3]
    clamped = ((mc[:, 0] / mc[:, 2]).abs() > 1.3 * tanx) | ((mc[:, 1] / mc[:, 2]).abs() > 1.3 * tany)
    assert int((clamped[:12] & (radii[:12] > 0)).sum()) >= 3             # the clamped branch is really taken
    ((rc[0] * sc["v_render"]).sum() + (ra[0, ..., 0] * sc["v_alphas"]).sum()).backward()
    out = C.rasterization(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["rgbs"], V, K, W, H,
                          render_mode="RGB+D", v_render=sc["v_render"], v_alphas=sc["v_alphas"], threads=2, **kw)
    np.testing.assert_allclose(out["render"], rc[0].detach().numpy(), rtol=1e-10, atol=1e-12)
    for name, got, want in (("means", out["v_means"], ins[0].grad), ("quats", out["v_quats"], ins[1].grad),
                            ("scales", out["v_scales"], ins[2].grad), ("opacities", out["v_opacities"], ins[3].grad),
                            ("colors", out["v_colors"], ins[4].grad), ("viewmat", out["v_viewmat"][:3], Vg.grad[:3])):
        want = want.numpy()
        assert np.abs(got - want).max() <= 1e-9 * max(np.abs(want).max(), 1e-30), name
