"""GPU parity: every HIP stage operator (through the C ABI) against the CPU oracle.

Tolerance (north_star): rendered depth and pose gradients within 1e-4 relative.
Integer outputs (radii, tile counts, intersection keys, offsets) must be bit-exact
when both sides are given the same fp32 inputs.  Discrete compositing decisions
(alpha < 1/255, T <= 1e-4) can flip for a pixel that sits on a threshold, so image
comparisons allow a small fraction of outlier pixels and say so.
"""
import math

import numpy as np
import pytest
import torch

from oracle import gsplat_oracle as G
from tests.parity import POSE_GRAD_TOL, agreeing_pixels, report
from tests.scenes import random_scene, sh_from_rgb, small_pose

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _gpu():
    import gsplatloc_amd as A
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return A


def mostly_close(a, b, rtol=1e-4, atol=1e-5, max_bad_frac=0.0, what=""):
    a = torch.as_tensor(a).detach().cpu().double()
    b = torch.as_tensor(b).detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    bad = (a - b).abs() > (atol + rtol * b.abs())
    frac = bad.double().mean().item() if bad.numel() else 0.0
    assert frac <= max_bad_frac, f"{what}: {frac:.2e} of elements differ (max |d|={float((a - b).abs().max()):.3e})"


def rel_inf(a, b):
    a = torch.as_tensor(a).detach().cpu().double()
    b = torch.as_tensor(b).detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def _scene32(N=2000, W=160, H=120, **kw):
    sc = random_scene(N, W, H, dtype=torch.float32, **kw)
    return sc


@pytest.mark.parametrize("aniso", [False, True])
def test_projection_fwd_bwd(aniso):
    A = _gpu()
    sc = _scene32(3000, aniso=aniso, sigma_px=1.5)
    W, H = sc["W"], sc["H"]
    c2w = small_pose(2.0, 0.05, dtype=torch.float32)
    V = torch.linalg.inv(c2w)[None]
    Ks = sc["K"][None]
    # oracle (fp32 on CPU)
    mo, qo, so, Vo = (x.clone().requires_grad_() for x in (sc["means"], sc["quats"], sc["scales"], V))
    r_o, m2_o, d_o, c_o, cp_o = G.fully_fused_projection(mo, qo, so, Vo, Ks, W, H, calc_compensations=True)
    # HIP
    mg, qg, sg, Vg = (x.to(DEV).clone().requires_grad_() for x in (sc["means"], sc["quats"], sc["scales"], V))
    r_g, m2_g, d_g, c_g, cp_g = A.fully_fused_projection(mg, None, qg, sg, Vg, Ks.to(DEV), W, H,
                                                         calc_compensations=True)
    agree = (r_g.cpu() == r_o)
    assert agree.float().mean() > 0.999, "radii must match except at ceil() borderlines"
    keep = agree & (r_o > 0)
    assert keep.sum() > 1000
    for a, b, nm in ((m2_g, m2_o, "means2d"), (d_g, d_o, "depths"), (c_g, c_o, "conics"), (cp_g, cp_o, "comp")):
        mostly_close(a.cpu()[keep], b[keep], rtol=1e-4, atol=1e-6, what=nm)
    # culled entries are zero-filled
    culled = m2_g.cpu()[r_g.cpu() == 0]
    assert culled.numel() == 0 or float(culled.abs().max()) == 0.0
    gen = torch.Generator().manual_seed(11)
    vm2, vd, vc, vcp = (torch.randn(x.shape, generator=gen) * keep.float().reshape(keep.shape + (1,) * (x.dim() - 2))
                        for x in (m2_o, d_o, c_o, cp_o))
    ((m2_o * vm2).sum() + (d_o * vd).sum() + (c_o * vc).sum() + (cp_o * vcp).sum()).backward()
    ((m2_g * vm2.to(DEV)).sum() + (d_g * vd.to(DEV)).sum() + (c_g * vc.to(DEV)).sum()
     + (cp_g * vcp.to(DEV)).sum()).backward()
    assert rel_inf(Vg.grad[0, :3], Vo.grad[0, :3]) < 1e-4, "v_viewmat"
    assert float(Vg.grad[0, 3].abs().max()) == 0.0
    mostly_close(mg.grad, mo.grad, rtol=2e-3, atol=1e-3 * float(mo.grad.abs().max()), what="v_means")
    mostly_close(sg.grad, so.grad, rtol=2e-3, atol=1e-3 * float(so.grad.abs().max()), what="v_scales")
    mostly_close(qg.grad, qo.grad, rtol=2e-3, atol=1e-3 * float(qo.grad.abs().max()) + 1e-7, what="v_quats")


def test_projection_pose_only_matches_full():
    A = _gpu()
    sc = _scene32(2000, sigma_px=1.0)
    V = torch.linalg.inv(small_pose(1.0, 0.02, dtype=torch.float32))[None].to(DEV)
    K = sc["K"][None].to(DEV)
    out = {}
    for full in (True, False):
        m = sc["means"].to(DEV).requires_grad_(full)
        Vg = V.clone().requires_grad_()
        r, m2, d, c, _ = A.fully_fused_projection(m, None, sc["quats"].to(DEV), sc["scales"].to(DEV), Vg, K, 160, 120)
        ((m2 ** 2).sum() + (d * 0.3).sum() + c.sum()).backward()
        out[full] = Vg.grad.clone()
    # two template instantiations of one kernel: the compiler may contract FMAs differently
    assert rel_inf(out[True], out[False]) < 1e-6, "pose-only mode must agree with the full backward"


@pytest.mark.parametrize("shape", [(160, 120), (100, 70), (33, 17)])
def test_binning_bit_exact(shape):
    A = _gpu()
    W, H = shape
    sc = _scene32(4000, W, H, sigma_px=3.0)
    V = torch.linalg.inv(small_pose(1.0, 0.02, dtype=torch.float32))[None]
    r, m2, d, c, _ = G.fully_fused_projection(sc["means"], sc["quats"], sc["scales"], V, sc["K"][None], W, H)
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    for sort in (True, False):
        tpg_o, ids_o, fl_o = G.isect_tiles(m2, r, d, 16, tw, th, sort=sort)
        tpg_g, ids_g, fl_g = A.isect_tiles(m2.to(DEV), r.to(DEV), d.to(DEV), 16, tw, th, sort=sort)
        assert torch.equal(tpg_g.cpu(), tpg_o)
        assert torch.equal(ids_g.cpu(), ids_o), f"isect_ids sort={sort}"
        assert torch.equal(fl_g.cpu(), fl_o), f"flatten_ids sort={sort}"
    off_o = G.isect_offset_encode(ids_o if sort else ids_o, 1, tw, th)
    tpg_o, ids_o, fl_o = G.isect_tiles(m2, r, d, 16, tw, th, sort=True)
    off_o = G.isect_offset_encode(ids_o, 1, tw, th)
    off_g = A.isect_offset_encode(ids_o.to(DEV), 1, tw, th)
    assert torch.equal(off_g.cpu(), off_o)


def test_binning_multi_camera_and_empty():
    A = _gpu()
    W, H = 96, 64
    sc = _scene32(1500, W, H, sigma_px=2.0)
    Vs = torch.stack([torch.linalg.inv(small_pose(a, 0.02, seed=s, dtype=torch.float32)) for a, s in ((0.5, 1), (3.0, 2))])
    Ks = sc["K"][None].repeat(2, 1, 1)
    r, m2, d, c, _ = G.fully_fused_projection(sc["means"], sc["quats"], sc["scales"], Vs, Ks, W, H)
    tw, th = 6, 4
    tpg_o, ids_o, fl_o = G.isect_tiles(m2, r, d, 16, tw, th)
    tpg_g, ids_g, fl_g = A.isect_tiles(m2.to(DEV), r.to(DEV), d.to(DEV), 16, tw, th)
    assert torch.equal(ids_g.cpu(), ids_o) and torch.equal(fl_g.cpu(), fl_o) and torch.equal(tpg_g.cpu(), tpg_o)
    assert torch.equal(A.isect_offset_encode(ids_g, 2, tw, th).cpu(), G.isect_offset_encode(ids_o, 2, tw, th))
    # nothing visible
    r0 = torch.zeros_like(r)
    tpg, ids, fl = A.isect_tiles(m2.to(DEV), r0.to(DEV), d.to(DEV), 16, tw, th)
    assert ids.numel() == 0 and fl.numel() == 0 and int(tpg.sum()) == 0
    off = A.isect_offset_encode(ids, 2, tw, th)
    assert off.shape == (2, th, tw) and int(off.abs().sum()) == 0


@pytest.mark.parametrize("kernel", ["wave", "wg"])
@pytest.mark.parametrize("N", [5, 63, 200, 256, 257, 700, 1024, 1025, 2048, 2049, 3000, 4097, 8192, 12289, 20000, 70001])
def test_binning_long_tile_list_uses_global_sort(N, kernel, monkeypatch):
    """One tile list of every size class of the sort: a wave's registers (4, 8, 16, 32 keys per lane, at and around the
    class boundaries), and beyond 2048 entries the block-wise long-list sort with whole and partial last blocks -- by
    the one-tile-per-wave kernel and by the one-tile-per-workgroup kernel (quarters sorted by the waves, merged in LDS;
    GSL_DEV_TILE_SORT forces either, the library picks by the mean list length)."""
    A = _gpu()
    monkeypatch.setenv("GSL_DEV_TILE_SORT", kernel)
    g = torch.Generator().manual_seed(3)
    m2 = (torch.rand(1, N, 2, generator=g) * 14 + 1).float()
    r = torch.full((1, N), 1, dtype=torch.int32)
    d = (torch.rand(1, N, generator=g) * 5 + 0.5).float()
    d[0, ::7] = d[0, 3]  # depth ties: order falls back to the Gaussian index
    tpg_o, ids_o, fl_o = G.isect_tiles(m2, r, d, 16, 2, 2)
    tpg_g, ids_g, fl_g = A.isect_tiles(m2.to(DEV), r.to(DEV), d.to(DEV), 16, 2, 2)
    assert ids_o.numel() == N
    assert torch.equal(ids_g.cpu(), ids_o) and torch.equal(fl_g.cpu(), fl_o)


def _stage_inputs(W, H, N, sigma_px, opacity, D, seed=5):
    sc = _scene32(N, W, H, sigma_px=sigma_px, opacity=opacity)
    V = torch.linalg.inv(small_pose(0.5, 0.01, dtype=torch.float32))[None]
    r, m2, d, c, _ = G.fully_fused_projection(sc["means"], sc["quats"], sc["scales"], V, sc["K"][None], W, H)
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    _, ids, fl = G.isect_tiles(m2, r, d, 16, tw, th)
    offs = G.isect_offset_encode(ids, 1, tw, th)
    gen = torch.Generator().manual_seed(seed)
    cols = torch.rand(1, N, D, generator=gen)
    cols[..., -1] = d  # last channel = depth, as in RGB+ED
    return sc, m2, c, cols, sc["opacities"][None].contiguous(), offs, fl


@pytest.mark.parametrize("D,opacity,sigma_px", [(1, None, 1.0), (4, None, 0.0), (4, (0.1, 0.9), 2.5), (3, (0.3, 1.0), 6.0),
                                                 (7, (0.2, 0.8), 1.5)])
def test_rasterize_fwd_bwd(D, opacity, sigma_px):
    A = _gpu()
    W, H, N = 100, 70, 3000
    sc, m2, c, cols, opa, offs, fl = _stage_inputs(W, H, N, sigma_px, opacity, D)
    bg = torch.rand(1, D, generator=torch.Generator().manual_seed(9)) if D == 3 else None
    # float64 oracle on the SAME fp32 inputs
    ins_o = [x.double().clone().requires_grad_() for x in (m2, c, cols, opa)]
    rc_o, ra_o = G.rasterize_to_pixels(*ins_o, W, H, 16, offs, fl, backgrounds=bg.double() if bg is not None else None)
    ins_g = [x.to(DEV).clone().requires_grad_() for x in (m2, c, cols, opa)]
    rc_g, ra_g = A.rasterize_to_pixels(*ins_g, W, H, 16, offs.to(DEV), fl.to(DEV),
                                       backgrounds=bg.to(DEV) if bg is not None else None)
    assert rc_g.shape == (1, H, W, D) and ra_g.shape == (1, H, W, 1)
    # a pixel exactly on the alpha / T thresholds may differ: allow 0.2 % outliers
    mostly_close(rc_g, rc_o, rtol=1e-4, atol=1e-5, max_bad_frac=2e-3, what="render_colors")
    mostly_close(ra_g, ra_o, rtol=1e-4, atol=1e-5, max_bad_frac=2e-3, what="render_alphas")
    gen = torch.Generator().manual_seed(21)
    v_c = torch.randn(rc_o.shape, generator=gen)
    v_a = torch.randn(ra_o.shape, generator=gen)
    ok = agreeing_pixels(rc_g, ra_g, rc_o, ra_o)  # flip-aware, see tests/parity.py
    v_c, v_a = v_c * ok[..., None], v_a * ok[..., None]
    ((rc_o * v_c.double()).sum() + (ra_o * v_a.double()).sum()).backward()
    ((rc_g * v_c.to(DEV)).sum() + (ra_g * v_a.to(DEV)).sum()).backward()
    for g_t, o_t, nm in zip(ins_g, ins_o, ("v_means2d", "v_conics", "v_colors", "v_opacities")):
        scale = float(o_t.grad.abs().max())
        mostly_close(g_t.grad, o_t.grad, rtol=1e-3, atol=1e-4 * scale, max_bad_frac=1e-2, what=nm)
        # the sum over the Gaussians cancels to a few per cent of sum |entries|: float32 rounding of the entries
        # (1e-5 relative each) shows up magnified in the total
        assert rel_inf(g_t.grad.sum(1), o_t.grad.sum(1)) < 5e-4, nm + " (summed)"


def test_rasterize_empty_and_edge_tiles():
    A = _gpu()
    W, H = 37, 21  # ragged: partial tiles on both edges
    m2 = torch.tensor([[[36.2, 20.1], [0.3, 0.2], [18.0, 10.0]]])
    con = torch.tensor([[[0.5, 0.0, 0.5]] * 3])
    col = torch.tensor([[[1.0], [2.0], [3.0]]])
    opa = torch.tensor([[0.9, 0.8, 0.7]])
    r = torch.tensor([[4, 4, 4]], dtype=torch.int32)
    d = torch.tensor([[1.0, 2.0, 3.0]])
    tw, th = 3, 2
    _, ids, fl = G.isect_tiles(m2, r, d, 16, tw, th)
    offs = G.isect_offset_encode(ids, 1, tw, th)
    rc_o, ra_o = G.rasterize_to_pixels(m2.double(), con.double(), col.double(), opa.double(), W, H, 16, offs, fl)
    rc_g, ra_g = A.rasterize_to_pixels(m2.to(DEV), con.to(DEV), col.to(DEV), opa.to(DEV), W, H, 16, offs.to(DEV),
                                       fl.to(DEV))
    mostly_close(rc_g, rc_o, what="edge colors")
    mostly_close(ra_g, ra_o, what="edge alphas")
    # no intersections at all
    e_ids = torch.empty(0, dtype=torch.int64, device=DEV)
    offs0 = A.isect_offset_encode(e_ids, 1, tw, th)
    rc, ra = A.rasterize_to_pixels(m2.to(DEV), con.to(DEV), col.to(DEV), opa.to(DEV), W, H, 16, offs0,
                                   torch.empty(0, dtype=torch.int32, device=DEV))
    assert float(rc.abs().max()) == 0 and float(ra.abs().max()) == 0


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_spherical_harmonics(deg):
    A = _gpu()
    gen = torch.Generator().manual_seed(4)
    M, K = 1000, 16
    dirs = torch.randn(2, M, 3, generator=gen)
    coeffs = torch.randn(2, M, K, 3, generator=gen)
    masks = torch.rand(2, M, generator=gen) > 0.2
    do, co = dirs.double().requires_grad_(), coeffs.double().requires_grad_()
    out_o = G.spherical_harmonics(deg, do, co, masks)
    dg, cg = dirs.to(DEV).requires_grad_(), coeffs.to(DEV).requires_grad_()
    out_g = A.spherical_harmonics(deg, dg, cg, masks.to(DEV))
    mostly_close(out_g, out_o, rtol=1e-4, atol=1e-5, what="sh colors")
    v = torch.randn(out_o.shape, generator=gen)
    (out_o * v.double()).sum().backward()
    (out_g * v.to(DEV)).sum().backward()
    mostly_close(cg.grad, co.grad, rtol=1e-4, atol=1e-5, what="v_coeffs")
    v_dirs_o = do.grad if do.grad is not None else torch.zeros_like(do)  # degree 0 ignores dirs
    mostly_close(dg.grad, v_dirs_o, rtol=1e-3, atol=1e-4, what="v_dirs")


@pytest.mark.parametrize("fused", ["fused", "staged"])
@pytest.mark.parametrize("mode,sh", [("RGB+ED", 1), ("ED", 1), ("RGB", None), ("D", None), ("RGB+D", 1)])
def test_rasterization_end_to_end(mode, sh, fused, monkeypatch):
    """The exact keyword call of model.py:195-213 / geometry.py:117-132, through the fused
    pipeline and through the stage operators."""
    A = _gpu()
    monkeypatch.setenv("GSLOC_DISABLE_FUSED", "0" if fused == "fused" else "1")
    W, H, N = 160, 120, 6000
    sc = _scene32(N, W, H, sigma_px=1.2, opacity=(0.4, 1.0))
    c2w = small_pose(0.5, 0.01, dtype=torch.float32)
    colors = sh_from_rgb(sc["rgbs"]) if sh is not None else sc["rgbs"]
    kw = dict(sh_degree=sh, width=W, height=H, packed=False, absgrad=False, sparse_grad=False, far_plane=1e10,
              near_plane=1e-2, render_mode=mode, rasterize_mode="classic")
    V = torch.linalg.inv(c2w)[None]
    Vo = V.double().clone().requires_grad_()
    rc_o, ra_o, _ = G.rasterization(sc["means"].double(), sc["quats"].double(), sc["scales"].double(),
                                    sc["opacities"].double(), colors.double(), Vo, sc["K"].double()[None], **kw)
    Vg = V.to(DEV).clone().requires_grad_()
    rc_g, ra_g, meta = A.rasterization(means=sc["means"].to(DEV), quats=sc["quats"].to(DEV),
                                       scales=sc["scales"].to(DEV), opacities=sc["opacities"].to(DEV),
                                       colors=colors.to(DEV), viewmats=Vg, Ks=sc["K"][None].to(DEV), **kw)
    assert rc_g.shape == rc_o.shape and ra_g.shape == ra_o.shape
    mostly_close(rc_g, rc_o, rtol=1e-4, atol=2e-5, max_bad_frac=3e-3, what="render " + mode)
    mostly_close(ra_g, ra_o, rtol=1e-4, atol=2e-5, max_bad_frac=3e-3, what="alpha " + mode)
    for k in ("radii", "means2d", "depths", "conics", "opacities", "tiles_per_gauss", "isect_ids", "flatten_ids",
              "isect_offsets", "tile_width", "tile_height", "width", "height", "tile_size", "n_cameras"):
        assert k in meta
    gen = torch.Generator().manual_seed(2)
    v = torch.randn(rc_o.shape, generator=gen)
    # flip-aware (tests/parity.py): pixels whose forward disagrees carry no upstream gradient on either side
    ok = agreeing_pixels(rc_g, ra_g, rc_o, ra_o)
    v = v * ok[..., None]
    (rc_o * v.double()).sum().backward()
    (rc_g * v.to(DEV)).sum().backward()
    err = rel_inf(Vg.grad[0, :3], Vo.grad[0, :3])
    report(f"end-to-end {mode} {fused}", 1.0 - ok.double().mean().item(), v_viewmat=err)
    assert err < POSE_GRAD_TOL, f"v_viewmat {mode}: {err:.2e}"


@pytest.mark.parametrize("mode,sh_deg,aa", [("RGB+ED", 1, False), ("RGB+ED", 3, False), ("RGB", None, False),
                                            ("ED", 1, False), ("RGB+D", 2, True), ("D", None, True)])
def test_fused_full_gradients(mode, sh_deg, aa):
    """Fused pipeline: every input gradient (means, quats, scales, opacities, colours/SH, viewmat)
    against float64 autograd of the oracle; the loss touches every output channel and alpha."""
    A = _gpu()
    W, H, N = 128, 80, 2500
    sc = _scene32(N, W, H, sigma_px=1.5, opacity=(0.3, 1.0), aniso=True)
    gen = torch.Generator().manual_seed(17)
    if sh_deg is None:
        colors = sc["rgbs"]
    else:
        colors = torch.randn(N, (sh_deg + 1) ** 2, 3, generator=gen) * 0.3
    V = torch.linalg.inv(small_pose(1.0, 0.03, dtype=torch.float32))[None]
    kw = dict(sh_degree=sh_deg, width=W, height=H, packed=False, render_mode=mode,
              rasterize_mode="antialiased" if aa else "classic")
    names = ("means", "quats", "scales", "opacities")
    ins_o = [sc[k].double().clone().requires_grad_() for k in names] + [colors.double().clone().requires_grad_(),
                                                                         V.double().clone().requires_grad_()]
    rc_o, ra_o, _ = G.rasterization(*ins_o[:5], ins_o[5], sc["K"].double()[None], **kw)
    ins_g = [sc[k].to(DEV).clone().requires_grad_() for k in names] + [colors.to(DEV).clone().requires_grad_(),
                                                                       V.to(DEV).clone().requires_grad_()]
    rc_g, ra_g, meta = A.rasterization(*ins_g[:5], viewmats=ins_g[5], Ks=sc["K"][None].to(DEV), **kw)
    assert "Q0" not in meta and meta["means2d"].shape == (1, N, 2) and meta["conics"].shape == (1, N, 3)
    mostly_close(rc_g, rc_o, rtol=1e-4, atol=2e-5, max_bad_frac=3e-3, what="render")
    mostly_close(ra_g, ra_o, rtol=1e-4, atol=2e-5, max_bad_frac=3e-3, what="alpha")
    v_c = torch.randn(rc_o.shape, generator=gen)
    v_a = torch.randn(ra_o.shape, generator=gen)
    # flip-aware (tests/parity.py): threshold-sitting pixels carry no upstream gradient on either side
    ok = agreeing_pixels(rc_g, ra_g, rc_o, ra_o)
    v_c, v_a = v_c * ok[..., None], v_a * ok[..., None]
    ((rc_o * v_c.double()).sum() + (ra_o * v_a.double()).sum()).backward()
    ((rc_g * v_c.to(DEV)).sum() + (ra_g * v_a.to(DEV)).sum()).backward()
    err = rel_inf(ins_g[5].grad[0, :3], ins_o[5].grad[0, :3])
    report(f"fused full gradients {mode} sh={sh_deg} aa={aa}", 1.0 - ok.double().mean().item(), v_viewmat=err)
    assert err < POSE_GRAD_TOL, f"v_viewmat: {err:.2e}"
    for g_t, o_t, nm in zip(ins_g[:5], ins_o[:5], names + ("colors",)):
        if o_t.grad is None:
            assert g_t.grad is None or float(g_t.grad.abs().max()) == 0.0, nm
            continue
        # per-Gaussian gradients: element-wise 1e-3 relative with a floor of 1e-4 of the largest entry (a float32 sum
        # of ~50 signed terms; a splat whose own alpha sits on 1/255 at a nearly opaque pixel switches on or off
        # without moving the pixel: a bounded fraction of entries), and 1e-4 on the sum over Gaussians
        scale = float(o_t.grad.abs().max())
        mostly_close(g_t.grad, o_t.grad, rtol=1e-3, atol=1e-4 * scale, max_bad_frac=1e-2, what="v_" + nm)
        assert rel_inf(g_t.grad.sum(0), o_t.grad.sum(0)) < 5e-4, nm + " (summed)"  # cancelling sum, see above


def test_fused_tile_strip_matches_full_render():
    """Rendering tile rows [ty0,ty1) only reproduces those rows of the full render bit for bit, and the
    strips' pose gradients add up to the full one (screen-tile parallelism)."""
    A = _gpu()
    from gsplatloc_amd.fused import fused_rasterization
    W, H, N = 160, 120, 5000
    sc = _scene32(N, W, H, sigma_px=1.3, opacity=(0.4, 1.0))
    sh = sh_from_rgb(sc["rgbs"]).to(DEV)
    V0 = torch.linalg.inv(small_pose(0.5, 0.01, dtype=torch.float32)).to(DEV)
    args = [sc[k].to(DEV) for k in ("means", "quats", "scales", "opacities")]
    gen = torch.Generator().manual_seed(8)
    v = torch.randn(H, W, 4, generator=gen).to(DEV)

    def run(rows):
        Vg = V0.clone().requires_grad_()
        r, a, m = fused_rasterization(*args, sh, Vg, sc["K"].to(DEV), W, H, sh_degree=1, render_mode="RGB+ED",
                                      tile_rows=rows)
        (r * v).sum().backward()
        return r.detach(), a.detach(), Vg.grad.clone()

    th = (H + 15) // 16
    r_full, a_full, g_full = run(None)
    g_sum = torch.zeros_like(g_full)
    for rows in ((0, 3), (3, 4), (4, th)):
        r, a, g = run(rows)
        y0, y1 = rows[0] * 16, min(rows[1] * 16, H)
        assert torch.equal(r[y0:y1], r_full[y0:y1]) and torch.equal(a[y0:y1], a_full[y0:y1])
        assert float(r[:y0].abs().max() if y0 else 0) == 0 and float(r[y1:].abs().max() if y1 < H else 0) == 0
        g_sum += g
    assert rel_inf(g_sum, g_full) < 1e-5


@pytest.mark.parametrize("mode,full", [("RGB+ED", True), ("ED", False), ("RGB+ED", False), ("RGB", True)])
def test_tiny_splat_backward_matches_general_backward(mode, full, monkeypatch):
    """RenderContext picks the tiny-splat backward (4x4 record slabs folded inside the projection backward, no atomics) when r_cull < 2 px;
    its gradients must agree with the general compositing backward (quadrant walk, MFMA pixel sums) on the same
    render."""
    A = _gpu()
    from gsplatloc_amd.context import RenderContext
    W, H, N = 200, 136, 30000
    sc = _scene32(N, W, H, sigma_px=0.0)   # scales -> 0: every splat is the 0.3 px^2 blur (radius 2)
    sh = sh_from_rgb(sc["rgbs"]).to(DEV)
    ins = [sc[k].to(DEV) for k in ("means", "quats", "scales", "opacities")] + [sh]
    V = torch.linalg.inv(small_pose(0.5, 0.01, dtype=torch.float32)).to(DEV).contiguous()
    K = sc["K"].to(DEV).contiguous()
    gen = torch.Generator().manual_seed(3)
    D = {"RGB+ED": 4, "ED": 1, "RGB": 3}[mode]
    v = torch.randn(H, W, D, generator=gen).to(DEV)
    if mode == "RGB+ED" and not full:
        v[..., :3] = 0  # the tracker's situation: only the depth channel carries a gradient
    va = torch.randn(H, W, 1, generator=gen).to(DEV)
    out = {}
    for which in ("tiny", "general"):  # ("auto" would pick the general backward: the scene's Gaussians are in random order)
        monkeypatch.setenv("GSLOC_BWD", which)
        rc = RenderContext(N, W, H, mode, sh_degree=1, K_sh=4, device=DEV, full_grads=full)
        rc.calibrate(*ins, V, K)
        assert rc.tiny == (which == "tiny")
        for _ in range(2):  # twice: the slabs / rows must come back clean
            rc.forward(*ins, V, K)
            g = rc.backward(v, va, full=full)
        rc.check_capacity()
        out[which] = {k: (t.clone() if t is not None else None) for k, t in g.items()}
    assert rel_inf(out["tiny"]["viewmat"], out["general"]["viewmat"]) < 2e-5
    if full:
        for k in ("means", "scales", "opacities", "colors"):
            scale = float(out["general"][k].abs().max())
            mostly_close(out["tiny"][k], out["general"][k], rtol=1e-3, atol=1e-5 * scale, max_bad_frac=1e-3, what=k)


@pytest.mark.parametrize("kernel", ["wave", "wg"])
def test_binned_projection_gives_the_lists_of_two_pass_binning(kernel, monkeypatch):
    """(Both tile-sort kernels.)  RenderContext bins directly from the projection kernel once calibrate() knows the tile sizes (no scatter pass);
    the sorted lists, offsets, render and gradients are those of the count -> scan -> scatter path, bit for bit
    (integer work) -- also for a strip, for the deterministic mode's sorted keys, and when a pose change makes the
    lists longer than at calibration.  A tile that outgrows its bin is flagged, never silently truncated."""
    _gpu()
    from gsplatloc_amd.context import RenderContext
    monkeypatch.setenv("GSL_DEV_TILE_SORT", kernel)
    W, H, N = 260, 200, 40000
    sc = _scene32(N, W, H, sigma_px=1.3, opacity=(0.3, 1.0))
    sh = sh_from_rgb(sc["rgbs"]).to(DEV)
    ins = [sc[k].to(DEV) for k in ("means", "quats", "scales", "opacities")] + [sh]
    V0 = torch.linalg.inv(small_pose(0.5, 0.01, dtype=torch.float32)).to(DEV).contiguous()
    V1 = torch.linalg.inv(small_pose(0.8, 0.02, dtype=torch.float32)).to(DEV).contiguous()
    K = sc["K"].to(DEV).contiguous()
    th = (H + 15) // 16
    v = torch.randn(H, W, 4, generator=torch.Generator().manual_seed(2)).to(DEV)
    va = torch.zeros(H, W, 1, device=DEV)
    for rows, det in ((None, False), ((3, 9), False), (None, True)):
        rc = RenderContext(N, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=DEV, tile_rows=rows, deterministic=det)
        rc.calibrate(*ins, V0, K)
        assert rc.bins is not None and rc.bin_cap > 0
        got = {}
        for name, V in (("calibrated pose", V0), ("moved pose", V1)):
            for binned in (True, False):
                if not binned:
                    keep = (rc.bins, rc.bin_cap)
                    rc.bins, rc.bin_cap = None, 0
                rc.forward(*ins, V, K)
                g = rc.backward(v, va)
                n = rc.check_capacity()
                got[binned] = dict(n=n, offs=rc.offs.clone(), ids=rc.flatten_ids[:n].clone(), render=rc.render.clone(),
                                   keys=rc.keys[:n].clone() if det else None,
                                   vm=g["viewmat"].clone(), means=g["means"].clone())
                if not binned:
                    rc.bins, rc.bin_cap = keep
            a, b = got[True], got[False]
            assert a["n"] == b["n"] > 0 and torch.equal(a["offs"], b["offs"]) and torch.equal(a["ids"], b["ids"]), name
            assert torch.equal(a["render"], b["render"])
            if det:
                assert torch.equal(a["keys"], b["keys"])
                assert torch.equal(a["vm"], b["vm"]) and torch.equal(a["means"], b["means"])
            else:
                assert float((a["vm"] - b["vm"]).abs().max()) <= 1e-5 * float(b["vm"].abs().max())
        # shrink the bins below the longest list: flagged with that length, and the lists hold what fitted
        longest = int((rc.offs[1:] - rc.offs[:-1]).max())
        rc._alloc_bins(longest // 2)
        rc.forward(*ins, V1, K)
        assert rc.bins_overflowed() == longest
        assert int((rc.offs[1:] - rc.offs[:-1]).max()) == longest // 2
        with pytest.raises(RuntimeError, match="outgrew its bin"):
            rc.check_capacity()
        rc.grow_bins(longest)
        rc.forward(*ins, V1, K)
        assert rc.bins_overflowed() == 0 and rc.check_capacity() == got[False]["n"]
        assert torch.equal(rc.flatten_ids[:got[False]["n"]], got[False]["ids"])


@pytest.mark.parametrize("case", ["all_culled", "single_gaussian", "odd_image", "huge_splats", "mostly_offscreen"])
def test_render_context_edge_cases(case):
    """RenderContext (binned projection, register sort, fused backward) on the inputs a tracker can meet at the edges:
    nothing visible, one Gaussian, an image smaller than a tile grid cell and not a multiple of 16, splats that cover
    every tile, a cloud that is mostly off screen -- render, alpha and the pose gradient against the float64 oracle."""
    _gpu()
    from gsplatloc_amd.context import RenderContext
    gen = torch.Generator().manual_seed(11)
    W, H, N, sigma, op = 96, 64, 600, 1.5, (0.3, 1.0)
    if case == "odd_image":
        W, H, N = 37, 21, 150
    elif case == "single_gaussian":
        N = 1
    elif case == "huge_splats":
        N, sigma, op = 40, 60.0, (0.05, 0.3)
    sc = _scene32(N, W, H, sigma_px=sigma, opacity=op)
    V = torch.linalg.inv(small_pose(0.7, 0.02, dtype=torch.float32))
    if case == "all_culled":
        sc["means"][:, 2] = -sc["means"][:, 2]          # behind the camera
    elif case == "mostly_offscreen":
        sc["means"][:, 0] += 0.8 * sc["means"][:, 2]    # shifted right: most of the cloud leaves the image
    sh = sh_from_rgb(sc["rgbs"])
    ins = [sc[k] for k in ("means", "quats", "scales", "opacities")] + [sh]
    Vo = V.double()[None].clone().requires_grad_()
    r_o, a_o, _ = G.rasterization(*[t.double() for t in ins], Vo, sc["K"].double()[None], W, H, sh_degree=1,
                                  render_mode="RGB+ED")
    rc = RenderContext(N, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=DEV, full_grads=True)
    dev_in = [t.to(DEV).contiguous() for t in ins] + [V.to(DEV).contiguous(), sc["K"].to(DEV).contiguous()]
    n = rc.calibrate(*dev_in)
    render, alphas = rc.forward(*dev_in)
    assert rc.check_capacity() == n
    if case == "all_culled":
        assert n == 0 and float(render.abs().max()) == 0.0 and float(alphas.abs().max()) == 0.0
    else:
        assert n > 0
    if case == "huge_splats":
        assert int((rc.offs[1:] - rc.offs[:-1]).min()) > 0   # every tile has a list
    mostly_close(render[None], r_o, rtol=1e-4, atol=2e-5, max_bad_frac=3e-3, what="render")
    mostly_close(alphas[None], a_o, rtol=1e-4, atol=2e-5, max_bad_frac=3e-3, what="alpha")
    ok = agreeing_pixels(render[None], alphas[None], r_o, a_o)
    v_c = torch.randn(r_o.shape, generator=gen) * ok[..., None]
    v_a = torch.randn(a_o.shape, generator=gen) * ok[..., None]
    loss_o = (r_o * v_c.double()).sum() + (a_o * v_a.double()).sum()
    if loss_o.requires_grad:   # (nothing visible: the oracle's outputs do not depend on the pose at all)
        loss_o.backward()
    g = rc.backward(v_c[0].float().to(DEV).contiguous(), v_a[0].float().to(DEV).contiguous())
    want = Vo.grad[0, :3] if Vo.grad is not None else torch.zeros(3, 4, dtype=torch.float64)
    if float(want.abs().max()) == 0.0:
        assert float(g["viewmat"].abs().max()) == 0.0 and float(g["means"].abs().max()) == 0.0
    else:
        err = rel_inf(g["viewmat"][:3], want)
        report(f"RenderContext edge case {case}", 1.0 - ok.double().mean().item(), v_viewmat=err)
        assert err < POSE_GRAD_TOL, err
    assert torch.isfinite(g["viewmat"]).all() and torch.isfinite(g["means"]).all()


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_render_context_against_the_oracle(seed):
    """Randomised configurations of the fused path (render mode, SH degree, anti-aliasing, anisotropy, splat size,
    opacity range, image size, tile-row strip, record staging, deterministic mode, near plane) against float64 autograd
    of the oracle: render, alpha, pose gradient (1e-4, flip-aware) and the summed Gaussian gradients."""
    _gpu()
    from gsplatloc_amd.context import RenderContext
    rng = np.random.default_rng(1000 + seed)
    W, H = int(rng.integers(20, 200)), int(rng.integers(20, 140))
    N = int(rng.integers(50, 3000))
    mode = ["RGB+ED", "ED", "RGB", "D", "RGB+D"][int(rng.integers(0, 5))]
    rgb = mode.startswith("RGB")
    sh_deg = int(rng.integers(0, 4)) if rgb else 1
    aa = bool(rng.integers(0, 2))
    sigma = float(rng.choice([0.0, 0.6, 1.2, 2.5, 5.0]))
    op = (float(rng.uniform(0.05, 0.5)), float(rng.uniform(0.6, 1.0)))
    th = (H + 15) // 16
    rows = None
    if rng.integers(0, 3) == 0 and th >= 3:
        r0 = int(rng.integers(0, th - 1))
        rows = (r0, int(rng.integers(r0 + 1, th + 1)))
    det = bool(rng.integers(0, 3) == 0)
    near = float(rng.choice([0.01, 1.5]))
    sc = _scene32(N, W, H, sigma_px=sigma, opacity=op, aniso=bool(rng.integers(0, 2)), seed=100 + seed)
    gen = torch.Generator().manual_seed(seed)
    K_sh = (sh_deg + 1) ** 2
    colors = torch.randn(N, K_sh, 3, generator=gen) * 0.4 if rgb else None
    V = torch.linalg.inv(small_pose(float(rng.uniform(0.0, 2.0)), float(rng.uniform(0.0, 0.05)), dtype=torch.float32))
    kw = dict(sh_degree=sh_deg if rgb else None, width=W, height=H, packed=False, render_mode=mode, near_plane=near,
              rasterize_mode="antialiased" if aa else "classic")
    names = ("means", "quats", "scales", "opacities")
    ins_o = [sc[k].double().clone().requires_grad_() for k in names]
    col_o = colors.double().clone().requires_grad_() if rgb else sc["rgbs"].double()
    Vo = V.double()[None].clone().requires_grad_()
    r_o, a_o, _ = G.rasterization(*ins_o, col_o, Vo, sc["K"].double()[None], **kw)
    rc = RenderContext(N, W, H, mode, sh_degree=sh_deg if rgb else None, K_sh=K_sh if rgb else 0, device=DEV,
                       near_plane=near, antialiased=aa, tile_rows=rows, deterministic=det, full_grads=True)
    dev_in = [sc[k].to(DEV).contiguous() for k in names] + [colors.to(DEV).contiguous() if rgb else None,
                                                             V.to(DEV).contiguous(), sc["K"].to(DEV).contiguous()]
    rc.calibrate(*dev_in)
    render, alphas = rc.forward(*dev_in)
    rc.check_capacity()
    y0, y1 = (0, H) if rows is None else (rows[0] * 16, min(rows[1] * 16, H))
    tag = f"fuzz {seed}: {mode} sh={sh_deg} aa={aa} sigma={sigma} {W}x{H} N={N} rows={rows} det={det} near={near}"
    mostly_close(render[None, y0:y1], r_o[:, y0:y1], rtol=1e-4, atol=2e-5, max_bad_frac=5e-3, what=tag + " render")
    mostly_close(alphas[None, y0:y1], a_o[:, y0:y1], rtol=1e-4, atol=2e-5, max_bad_frac=5e-3, what=tag + " alpha")
    ok = agreeing_pixels(render[None], alphas[None], r_o, a_o)
    ok[:, :y0] = False
    ok[:, y1:] = False
    v_c = torch.randn(r_o.shape, generator=gen) * ok[..., None]
    v_a = torch.randn(a_o.shape, generator=gen) * ok[..., None]
    loss_o = (r_o * v_c.double()).sum() + (a_o * v_a.double()).sum()
    if not loss_o.requires_grad:
        return
    loss_o.backward()
    g = rc.backward(v_c[0].float().to(DEV).contiguous(), v_a[0].float().to(DEV).contiguous())
    want = Vo.grad[0, :3]
    if float(want.abs().max()) > 0:
        err = rel_inf(g["viewmat"][:3], want)
        report(tag, 1.0 - ok[:, y0:y1].double().mean().item(), v_viewmat=err)
        assert err < POSE_GRAD_TOL, (tag, err)
    for nm, o_t in zip(names, ins_o):
        if o_t.grad is not None and float(o_t.grad.abs().max()) > 0 and nm != "quats":
            assert rel_inf(g[nm].sum(0), o_t.grad.sum(0)) < 2e-3, (tag, nm)


@pytest.mark.parametrize("mode,full", [("RGB+ED", True), ("ED", False), ("RGB", True)])
def test_deterministic_backward_is_bit_reproducible(mode, full):
    """RenderContext(deterministic=True): no float atomics in the backward (per-wave moment rows summed in wave order,
    one gradient row per intersection, a Gaussian's rows found by binary search and added in tile order), so two runs
    give bit-identical gradients; and they agree with the default (atomic) backward to rounding."""
    _gpu()
    from gsplatloc_amd.context import RenderContext
    W, H, N = 200, 136, 30000
    sc = _scene32(N, W, H, sigma_px=1.0, opacity=(0.3, 1.0))
    sh = sh_from_rgb(sc["rgbs"]).to(DEV)
    ins = [sc[k].to(DEV) for k in ("means", "quats", "scales", "opacities")] + [sh]
    V = torch.linalg.inv(small_pose(0.5, 0.01, dtype=torch.float32)).to(DEV).contiguous()
    K = sc["K"].to(DEV).contiguous()
    gen = torch.Generator().manual_seed(3)
    D = {"RGB+ED": 4, "ED": 1, "RGB": 3}[mode]
    v = torch.randn(H, W, D, generator=gen).to(DEV)
    va = torch.randn(H, W, 1, generator=gen).to(DEV)
    runs = []
    for det in (True, True, False):
        rc = RenderContext(N, W, H, mode, sh_degree=1, K_sh=4, device=DEV, full_grads=full, deterministic=det)
        rc.calibrate(*ins, V, K)
        assert not rc.tiny
        for _ in range(2):
            rc.forward(*ins, V, K)
            g = rc.backward(v, va, full=full)
        rc.check_capacity()
        runs.append({k: t.clone() for k, t in g.items() if t is not None})
    for k in runs[0]:
        assert torch.equal(runs[0][k], runs[1][k]), k          # bit-identical from run to run
        scale = float(runs[2][k].abs().max())
        if k == "quats":   # isotropic Gaussians: this gradient is rounding noise of a cancelling sum
            continue
        mostly_close(runs[0][k], runs[2][k], rtol=1e-3, atol=1e-5 * scale, max_bad_frac=1e-3, what=k)
    assert rel_inf(runs[0]["viewmat"], runs[2]["viewmat"]) < 2e-5


def test_tiny_backward_reports_a_splat_that_outgrew_its_slab(monkeypatch):
    """A context calibrated on pixel-sized splats whose scales then grow (sigma_px = 1: 8+ px wide): the tiny
    backward raises its sticky device flag instead of dropping gradient silently; the caller switches the context
    to the general backward and gets the right gradient."""
    _gpu()
    from gsplatloc_amd.context import RenderContext
    monkeypatch.setenv("GSLOC_BWD", "tiny")  # (the scene's Gaussians are in random order: "auto" would not pick it)
    W, H, N = 96, 64, 3000
    small, big = _scene32(N, W, H, sigma_px=0.0), _scene32(N, W, H, sigma_px=1.0)
    sh = sh_from_rgb(small["rgbs"]).to(DEV)
    V = torch.linalg.inv(small_pose(0.5, 0.01, dtype=torch.float32)).to(DEV).contiguous()
    K = small["K"].to(DEV).contiguous()
    ins = lambda sc: [sc[k].to(DEV) for k in ("means", "quats", "scales", "opacities")] + [sh, V, K]  # noqa: E731
    v, va = torch.ones(H, W, 4, device=DEV), torch.zeros(H, W, 1, device=DEV)
    rc = RenderContext(N, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=DEV, full_grads=False)
    rc.calibrate(*ins(small))
    assert rc.tiny and not rc.tiny_overflowed()
    rc.forward(*ins(big))
    rc.backward(v, va, full=False)
    assert rc.tiny_overflowed()
    with pytest.raises(RuntimeError, match="outgrew"):
        rc.check_capacity()
    rc.use_general_backward()
    rc.vacc.zero_()  # rows the interrupted tiny pass may have left
    rc.forward(*ins(big))
    got = rc.backward(v, va, full=False)["viewmat"].clone()
    ref = RenderContext(N, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=DEV, full_grads=False)
    ref.calibrate(*ins(big))
    assert not ref.tiny
    ref.forward(*ins(big))
    want = ref.backward(v, va, full=False)["viewmat"]
    assert rel_inf(got, want) < 1e-5


def test_legacy_pair_on_the_gpu():
    """project_gaussians / rasterize_gaussians (north_star's legacy operator pair, IDX:14774 / 14765) over the HIP
    stage operators against the ORACLE (oracle.gsplat_oracle.rasterization, float64 autograd) on the same inputs --
    not against another HIP path: images to the image tolerance, gradients to the view matrix
    and to the means flip-aware (tests/parity.py).  The glue itself is also covered on the CPU by
    tests/test_legacy_cpu.py."""
    import gsplat
    from oracle import gsplat_oracle as G
    from tests.scenes import random_scene, small_pose

    N, W, H = 4000, 200, 150
    sc = random_scene(N, W, H, seed=5, sigma_px=2.0, aniso=True, opacity=(0.3, 1.0), dtype=torch.float32)
    Vc = torch.linalg.inv(small_pose(1.0, 0.03, dtype=torch.float32))
    V = Vc.cuda()
    cu = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in sc.items()}
    fx, fy, cx, cy = (float(sc["K"][0, 0]), float(sc["K"][1, 1]), float(sc["K"][0, 2]), float(sc["K"][1, 2]))
    m1, V1 = cu["means"].clone().requires_grad_(), V.clone().requires_grad_()
    xys, depths, radii, conics, comp, hit, cov3d = gsplat.project_gaussians(
        m1, cu["scales"], 1.0, cu["quats"], V1, fx, fy, cx, cy, H, W, 16)
    img, alpha = gsplat.rasterize_gaussians(xys, depths, radii, conics, hit, cu["rgbs"], cu["opacities"][:, None], H, W, 16,
                                            return_alpha=True)
    assert cov3d.shape == (N, 6)
    # the oracle on the same float32 inputs, evaluated in float64
    m2, V2 = sc["means"].double().clone().requires_grad_(), Vc.double().clone().requires_grad_()
    ro, ao, meta = G.rasterization(m2, sc["quats"].double(), sc["scales"].double(), sc["opacities"].double(),
                                   sc["rgbs"].double(), V2[None], sc["K"].double()[None], W, H, render_mode="RGB")
    # (a radius is ceil(3 sqrt(lambda)): float32 against float64 may differ by one on a borderline splat)
    assert int((radii.cpu() != meta["radii"][0].to(torch.int32)).sum()) <= 2
    mostly_close(img, ro[0], rtol=1e-4, atol=2e-5, max_bad_frac=1e-3, what="legacy image vs oracle")
    mostly_close(alpha, ao[0, ..., 0], rtol=1e-4, atol=2e-5, max_bad_frac=1e-3, what="legacy alpha vs oracle")
    ok = agreeing_pixels(img, alpha, ro[0], ao[0])
    flipped = 1.0 - ok.double().mean().item()
    w = torch.linspace(0.5, 1.5, img.numel(), dtype=torch.float64).reshape(img.shape) * ok[..., None]
    (img * w.float().cuda()).sum().backward()
    (ro[0] * w).sum().backward()
    ev, em = rel_inf(V1.grad[:3], V2.grad[:3]), rel_inf(m1.grad, m2.grad)
    report("legacy pair vs float64 oracle", flipped, v_viewmat=ev, v_means=em)
    assert flipped < 1e-3
    assert ev < POSE_GRAD_TOL, ev
    assert em < 1e-3, em  # per-Gaussian gradients: largest entry of a single splat, float32 against float64


@pytest.mark.parametrize("kernel", ["wave", "wg"])
@pytest.mark.parametrize("seed", range(4))
def test_hip_binning_on_adversarial_inputs(seed, kernel, monkeypatch):
    """The HIP tile binning (stage operators) on the inputs of tests/test_c_oracle.py's adversarial case: centres
    outside the image, radii larger than the image, exact tile boundaries, equal depths.  Bit-exact against the
    oracle, with either tile-sort kernel."""
    import gsplatloc_amd as A
    from oracle import gsplat_oracle as G

    monkeypatch.setenv("GSL_DEV_TILE_SORT", kernel)

    g = torch.Generator().manual_seed(100 + seed)
    N, W, H, ts = 400, 150 + 7 * seed, 90 + 5 * seed, 16
    tw, th = (W + ts - 1) // ts, (H + ts - 1) // ts
    m2 = torch.stack([torch.rand(N, generator=g) * (W + 120) - 60, torch.rand(N, generator=g) * (H + 120) - 60], -1)
    m2[:40] = torch.round(m2[:40] / ts) * ts
    radii = torch.randint(1, 40, (N,), generator=g, dtype=torch.int32)
    radii[40:50] = 400
    radii[50:70] = 0
    dep = torch.rand(N, generator=g) * 5 + 0.5
    dep[70:90] = dep[70]
    tpg, ids, fids = G.isect_tiles(m2[None], radii[None], dep[None], ts, tw, th)
    offs = G.isect_offset_encode(ids, 1, tw, th)
    t2, i2, f2 = A.isect_tiles(m2[None].cuda(), radii[None].cuda(), dep[None].cuda(), ts, tw, th)
    o2 = A.isect_offset_encode(i2, 1, tw, th)
    assert torch.equal(t2.cpu(), tpg) and torch.equal(i2.cpu(), ids) and torch.equal(f2.cpu(), fids)
    assert torch.equal(o2.cpu(), offs)


def test_hip_frustum_clamp_branches():
    """The clamped branch of the EWA Jacobian (centres beyond 1.3x the half field of view) and the near / far /
    radius_clip culls through the HIP path, against the float64 oracle."""
    import gsplatloc_amd as A
    from oracle import gsplat_oracle as G
    from tests.scenes import frustum_clamp_scene

    sc = frustum_clamp_scene()
    W, H, kw = sc["W"], sc["H"], sc["kw"]
    ins = [sc[k].clone().requires_grad_() for k in ("means", "quats", "scales", "opacities", "rgbs")]
    Vo = sc["V"].clone().requires_grad_()
    ro, ao, mo = G.rasterization(*ins, Vo[None], sc["K"][None], W, H, render_mode="RGB+D", **kw)
    gin = [sc[k].float().cuda().requires_grad_() for k in ("means", "quats", "scales", "opacities", "rgbs")]
    Vg = sc["V"].float().cuda().requires_grad_()
    rg, ag, mg = A.rasterization(*gin, viewmats=Vg[None], Ks=sc["K"].float().cuda()[None], width=W, height=H,
                                 render_mode="RGB+D", packed=False, **kw)
    assert torch.equal(mg["radii"][0].cpu(), mo["radii"][0])
    ok = agreeing_pixels(rg[0], ag[0], ro[0], ao[0])  # flip-aware, tests/parity.py
    v_r, v_a = sc["v_render"] * ok[..., None], sc["v_alphas"] * ok
    ((ro[0] * v_r).sum() + (ao[0, ..., 0] * v_a).sum()).backward()
    ((rg[0] * v_r.float().cuda()).sum() + (ag[0, ..., 0] * v_a.float().cuda()).sum()).backward()
    bad = (rg[0].cpu().double() - ro[0].detach()).abs() > 2e-5 + 1e-4 * ro[0].detach().abs()
    assert float(bad.double().mean()) < 3e-3
    for got, want in zip(gin + [Vg], ins + [Vo]):
        a, b = got.grad.cpu().double(), want.grad
        if a.shape == (4, 4):
            a, b = a[:3], b[:3]
        assert float((a - b).abs().max()) <= 2e-4 * float(b.abs().max()) + 1e-9


@pytest.mark.gpu
def test_a_non_positive_near_plane_is_clamped_to_the_smallest_normal_depth():
    """The per-tile sorts compare (depth bits, index) keys as doubles, which holds for positive finite normal depths; the
    projection entry points clamp the depth window to [FLT_MIN, FLT_MAX] (include/gsloc_hip.h, "Sort keys").  A scene with
    Gaussians behind the camera rendered with near_plane = -1 must equal the render with near_plane = FLT_MIN bit for bit
    (nothing at z <= 0 is projected), and every tile list must come out ascending in (depth, index)."""
    import gsplatloc_amd as A
    g = torch.Generator().manual_seed(11)
    N, W, H = 3000, 160, 120
    means = torch.rand(N, 3, generator=g) * torch.tensor([4.0, 3.0, 6.0]) - torch.tensor([2.0, 1.5, 2.0])  # z in [-2, 4]
    quats = torch.tensor([1.0, 0, 0, 0]).repeat(N, 1)
    scales = torch.full((N, 3), 0.03)
    opac = torch.full((N,), 0.8)
    rgbs = torch.rand(N, 3, generator=g)
    K = torch.tensor([[120.0, 0, 79.5], [0, 120.0, 59.5], [0, 0, 1]])
    V = torch.eye(4)
    outs = {}
    for near in (-1.0, 1.17549435e-38):
        r, a, meta = A.rasterization(means.cuda(), quats.cuda(), scales.cuda(), opac.cuda(), rgbs.cuda(), viewmats=V.cuda()[None],
                                     Ks=K.cuda()[None], width=W, height=H, near_plane=near, render_mode="RGB+ED", packed=False,
                                     sh_degree=None)
        total = int(meta["tiles_per_gauss"].sum())
        outs[near] = (r.cpu(), a.cpu(), meta["radii"][0].cpu(), meta["flatten_ids"][:total].cpu(),
                      meta["isect_offsets"].cpu(), meta["depths"][0].cpu())
    a, b = outs[-1.0], outs[1.17549435e-38]
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    radii, fids, offs, depths = a[2], a[3].long(), a[4].reshape(-1).long(), a[5]
    assert int((radii[means[:, 2] <= 0] > 0).sum()) == 0 and int((radii > 0).sum()) > 500
    ends = torch.cat([offs[1:], torch.tensor([fids.numel()])])
    for s0, e0 in zip(offs.tolist(), ends.tolist()):
        d = depths[fids[s0:e0]]
        key = (d.view(torch.int32).long() << 32) | fids[s0:e0]
        assert bool((key[1:] > key[:-1]).all())


@pytest.mark.gpu
def test_wide_features_are_composited_in_channel_chunks():
    """gsplat.rasterization with more feature channels than one compositing kernel takes (32 feature
    channels + the depth channel of RGB+ED = 33): the general path splits the channels into chunks of channel_chunk,
    with a background colour, and every channel agrees with the float64 oracle."""
    import gsplatloc_amd as A

    N, W, H, Dc = 1500, 96, 64, 32
    sc = random_scene(N, W, H, seed=11, sigma_px=1.5, opacity=(0.3, 0.9), dtype=torch.float32)
    g = torch.Generator().manual_seed(12)
    feats = torch.rand(N, Dc, generator=g)
    bg = torch.rand(1, Dc, generator=g)
    V = torch.linalg.inv(small_pose(0.5, 0.01, dtype=torch.float32))[None]
    kw = dict(sh_degree=None, width=W, height=H, render_mode="RGB+ED")
    ro, ao, _ = G.rasterization(sc["means"].double(), sc["quats"].double(), sc["scales"].double(), sc["opacities"].double(),
                                feats.double(), V.double(), sc["K"].double()[None], backgrounds=bg.double(), **kw)
    for chunk in (32, 7):
        rg, ag, _ = A.rasterization(sc["means"].to(DEV), sc["quats"].to(DEV), sc["scales"].to(DEV), sc["opacities"].to(DEV),
                                    feats.to(DEV), V.to(DEV), sc["K"][None].to(DEV), backgrounds=bg.to(DEV),
                                    channel_chunk=chunk, **kw)
        assert rg.shape == (1, H, W, Dc + 1)
        mostly_close(rg, ro, rtol=1e-4, atol=2e-5, max_bad_frac=2e-3, what=f"33 channels, chunk {chunk}")
        mostly_close(ag, ao, rtol=1e-4, atol=2e-5, max_bad_frac=2e-3, what=f"alpha, chunk {chunk}")


def test_backward_without_the_forwards_hit_lists_matches_the_one_with():
    """The compositing backward walks the (block, entry) pairs the forward recorded (isect_hits); given NULL it scans
    the tile lists itself and tests every splat's alpha >= 1/255 disc against its quadrant's blocks.  Same gradients
    (the geometric test only adds pairs in which no pixel passes the alpha test)."""
    _gpu()
    from gsplatloc_amd.context import RenderContext
    W, H, N = 250, 190, 30000
    for sigma_px, aniso in ((1.3, False), (2.5, True)):
        sc = random_scene(N, W, H, seed=21, sigma_px=sigma_px, aniso=aniso, opacity=(0.3, 1.0), dtype=torch.float32)
        sh = sh_from_rgb(sc["rgbs"]).to(DEV)
        ins = [sc[k].to(DEV) for k in ("means", "quats", "scales", "opacities")] + [sh]
        V = torch.linalg.inv(small_pose(0.5, 0.01, dtype=torch.float32)).to(DEV).contiguous()
        K = sc["K"].to(DEV).contiguous()
        v = torch.randn(H, W, 4, generator=torch.Generator().manual_seed(2)).to(DEV)
        va = torch.randn(H, W, 1, generator=torch.Generator().manual_seed(3)).to(DEV)
        rc = RenderContext(N, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=DEV)
        rc.calibrate(*ins, V, K)
        assert not rc.tiny
        rc.forward(*ins, V, K)
        a = {k: t.clone() for k, t in rc.backward(v, va).items()}
        keep = (rc.hits, rc.hit_counts)
        rc.hits = rc.hit_counts = None
        rc.forward(*ins, V, K)
        b = {k: t.clone() for k, t in rc.backward(v, va).items()}
        rc.hits, rc.hit_counts = keep
        rc.check_capacity()
        for k in ("viewmat", "means", "scales", "opacities", "colors"):
            assert rel_inf(b[k], a[k]) < 2e-5, (sigma_px, k, rel_inf(b[k], a[k]))


def test_cached_drop_in_call_recomputes_recalibrates_and_owns_its_outputs():
    """gsplat.rasterization keeps one RenderContext per call signature (fused.py).  The cases the reference's loop never
    produces but a drop-in must survive: (1) two forwards before the first one's backward (the node re-runs its own
    forward); (2) a scene that outgrows the buffers measured on the first call (re-measured, the call repeated);
    (3) outputs and gradients are fresh tensors (a later call does not change an earlier result); all against the
    allocate-per-call path."""
    _gpu()
    import os

    import gsplatloc_amd as A
    from gsplatloc_amd.fused import clear_context_cache

    W, H, N = 200, 150, 6000
    sc = random_scene(N, W, H, seed=31, sigma_px=1.5, opacity=(0.3, 1.0), dtype=torch.float32)
    sh = sh_from_rgb(sc["rgbs"]).to(DEV)
    base = dict(quats=sc["quats"].to(DEV), opacities=sc["opacities"].to(DEV), colors=sh, Ks=sc["K"][None].to(DEV), width=W,
                height=H, sh_degree=1, render_mode="RGB+ED")
    means, scales = sc["means"].to(DEV), sc["scales"].to(DEV)
    Va = torch.linalg.inv(small_pose(0.5, 0.01, dtype=torch.float32)).to(DEV)
    Vb = torch.linalg.inv(small_pose(1.5, 0.05, dtype=torch.float32)).to(DEV)

    def call(V, scale_factor=1.0, cached=True):
        os.environ["GSLOC_DROPIN_CACHE"] = "1" if cached else "0"
        try:
            Vg = V.clone().requires_grad_()
            m = means.clone().requires_grad_()
            r, a, _ = A.rasterization(means=m, scales=scales * scale_factor, viewmats=Vg[None], **base)
            return r, a, Vg, m
        finally:
            os.environ.pop("GSLOC_DROPIN_CACHE", None)

    w = torch.linspace(0.5, 1.5, H * W * 4, device=DEV).reshape(1, H, W, 4)
    clear_context_cache()
    # reference results, allocate-per-call
    ref = {}
    for name, V, sf in (("a", Va, 1.0), ("b", Vb, 1.0), ("big", Va, 4.0)):
        r, a, Vg, m = call(V, sf, cached=False)
        (r * w).sum().backward()
        ref[name] = (r.detach().clone(), Vg.grad.clone(), m.grad.clone())
    # (1) + (3): forward a, forward b, THEN backward a, then backward b
    ra, aa, Vga, ma = call(Va)
    ra_copy = ra.detach().clone()
    rb, ab, Vgb, mb = call(Vb)
    assert torch.equal(ra.detach(), ra_copy), "a later call overwrote an earlier call's output"
    (ra * w).sum().backward()
    (rb * w).sum().backward()
    for got, want in (((ra, Vga, ma), ref["a"]), ((rb, Vgb, mb), ref["b"])):
        assert torch.equal(got[0].detach(), want[0])  # same kernels, same lists: bit-identical images
        assert rel_inf(got[1].grad, want[1]) < 1e-5 and rel_inf(got[2].grad, want[2]) < 1e-5
    # (2) four times larger splats: ~16 x the intersections of the calibration call
    rg, ag, Vgg, mg = call(Va, 4.0)
    (rg * w).sum().backward()
    assert torch.equal(rg.detach(), ref["big"][0])
    assert rel_inf(Vgg.grad, ref["big"][1]) < 1e-5 and rel_inf(mg.grad, ref["big"][2]) < 1e-5
    clear_context_cache()


@pytest.mark.parametrize("sigma_px,N,bwd", [(0.0, 60000, "tiny"), (1.3, 30000, "auto"), (2.2, 40000, "auto")])
def test_forward_that_sorts_its_own_bins_matches_the_separate_sort_launch(sigma_px, N, bwd, monkeypatch):
    """RenderContext(sort_in_forward=True): gsl_fused_raster_fwd(..., sort_bins) does gsl_fused_bin's work per tile (no
    sort launch) and the compositing backward clears the tile counters.  Offsets, lists, image bit for bit those of the
    separate launch; gradients equal; a forward nobody back-propagates is followed by a correct one (the context zeroes
    the counters); an outgrown bin is flagged.  Bins of <= 1024 keys (LK 2) and of 1025..2048 keys (LK 3); tiny and
    general backward."""
    _gpu()
    from gsplatloc_amd.context import RenderContext
    monkeypatch.setenv("GSLOC_BWD", bwd)
    monkeypatch.setenv("GSLOC_SORT_IN_FORWARD", "force")  # (RenderContext itself stops at bins of 1024 keys)
    W, H = 260, 200
    sc = _scene32(N, W, H, sigma_px=sigma_px, opacity=(0.3, 1.0))
    sh = sh_from_rgb(sc["rgbs"]).to(DEV)
    ins = [sc[k].to(DEV) for k in ("means", "quats", "scales", "opacities")] + [sh]
    V0 = torch.linalg.inv(small_pose(0.5, 0.01, dtype=torch.float32)).to(DEV).contiguous()
    V1 = torch.linalg.inv(small_pose(0.8, 0.02, dtype=torch.float32)).to(DEV).contiguous()
    K = sc["K"].to(DEV).contiguous()
    v = torch.randn(H, W, 4, generator=torch.Generator().manual_seed(2)).to(DEV)
    va = torch.zeros(H, W, 1, device=DEV)
    got = {}
    for sif in (False, True):
        rc = RenderContext(N, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=DEV, sort_in_forward=sif)
        rc.calibrate(*ins, V0, K)
        assert rc.sorts_in_forward() == sif and rc.tiny == (bwd == "tiny")
        if sigma_px > 2.0:
            assert rc.bin_cap > 1024
        elif sigma_px > 0:
            assert rc.bin_cap <= 1024
        out = []
        for V in (V0, V1, V1):
            rc.forward(*ins, V, K)
            g = rc.backward(v, va)
            torch.cuda.synchronize()
            n = rc.check_capacity()
            out.append(dict(n=n, offs=rc.offs.clone(), ids=rc.flatten_ids[:n].clone(), render=rc.render.clone(),
                            vm=g["viewmat"].clone(), means=g["means"].clone()))
        # forward, forward (nobody back-propagated the first), backward
        rc.forward(*ins, V0, K)
        rc.forward(*ins, V1, K)
        g = rc.backward(v, va)
        torch.cuda.synchronize()
        n = rc.check_capacity()
        out.append(dict(n=n, offs=rc.offs.clone(), ids=rc.flatten_ids[:n].clone(), render=rc.render.clone(),
                        vm=g["viewmat"].clone(), means=g["means"].clone()))
        got[sif] = out
        if sif:  # an outgrown bin is flagged by the sorting forward too
            longest = int((rc.offs[1:] - rc.offs[:-1]).max())
            rc._alloc_bins(longest // 2)
            rc.forward(*ins, V1, K)
            assert rc.bins_overflowed() == longest
    for a, b in zip(got[False], got[True]):
        assert a["n"] == b["n"] > 0 and torch.equal(a["offs"], b["offs"]) and torch.equal(a["ids"], b["ids"])
        assert torch.equal(a["render"], b["render"])
        assert float((a["vm"] - b["vm"]).abs().max()) <= 1e-5 * float(a["vm"].abs().max())
        assert float((a["means"] - b["means"]).abs().max()) <= 1e-4 * float(a["means"].abs().max())
    assert torch.equal(got[True][1]["render"], got[True][3]["render"])
