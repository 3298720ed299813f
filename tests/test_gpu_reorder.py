"""Tile-order placement of randomly ordered Gaussians (RenderContext(reorder=True)).

The context stores its own copies of the per-Gaussian inputs sorted by the tile of the projected centre; the sort key's
low word stays the ORIGINAL index, so depth ties break as in the caller's order.  Claim under test: against the
unpermuted run the tile lists are the same lists with every id relabelled (order_ids[list id] == the plain run's id),
offsets, image, alpha and last_ids bit-identical, gradients equal up to the summation order of the float atomics --
on a fronto-parallel wall on which EVERY Gaussian has the same depth (the whole order comes from the tie-break), through
every sort of the library: one wave per tile, one tile per workgroup, the forward that sorts its own bin, two-pass
binning, and the multi-workgroup sort of a long list.  Semantics restated: SURVEY.md A.2 (isect_tiles sorts by
(depth bits, Gaussian id)); reference inputs are static per frame: /root/reference/src/my_gsplat/model.py:137-175."""
import pytest
import torch

from tests.scenes import sh_from_rgb, small_pose

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _wall(N, W, H, sigma_px, seed=3, z=2.0, pile=0):
    """N Gaussians on the plane z = const seen by an identity camera: every depth is bit-identical.  Random order."""
    g = torch.Generator().manual_seed(seed)
    fx = 0.5 * W
    cx, cy = (W - 1) / 2.0, (H - 1) / 2.0
    u = torch.rand(N, generator=g) * W
    v = torch.rand(N, generator=g) * H
    if pile:  # `pile` of them on one spot: one tile list far longer than the rest
        u[:pile] = 0.37 * W + torch.rand(pile, generator=g) * 3.0
        v[:pile] = 0.61 * H + torch.rand(pile, generator=g) * 3.0
        shuffle = torch.randperm(N, generator=g)
        u, v = u[shuffle], v[shuffle]
    zz = torch.full((N,), z)
    means = torch.stack([(u - cx) / fx * zz, (v - cy) / fx * zz, zz], -1).float().contiguous()
    quats = torch.tensor([1.0, 0, 0, 0]).repeat(N, 1).contiguous()
    scales = (max(sigma_px, 1e-3) * zz / fx)[:, None].repeat(1, 3).float().contiguous()
    opac = (0.05 + 0.5 * torch.rand(N, generator=g)).float()
    if pile:
        opac = opac * 0.02 + 0.004  # so that the pile's list is walked deep
    sh = sh_from_rgb(torch.rand(N, 3, generator=g)).float().contiguous()
    K = torch.tensor([[fx, 0, cx], [0, fx, cy], [0, 0, 1]]).float()
    return means, quats, scales, opac.contiguous(), sh, K


def _run(ins, V, K, W, H, v, va, reorder, **kw):
    from gsplatloc_amd.context import RenderContext
    N = ins[0].shape[0]
    rc = RenderContext(N, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=DEV, reorder=reorder, **kw)
    rc.calibrate(*ins, V, K)
    assert (rc.order_ids is not None) == reorder
    outs = []
    for it in range(2):  # the second iteration runs on the binned path with the buffers calibrate() sized
        rc.forward(*ins, V, K)
        g = rc.backward(v, va)
        torch.cuda.synchronize()
        n = rc.check_capacity()
        g = rc.grads_in_input_order(g)
        ids = rc.flatten_ids[:n].long()
        if reorder:
            ids = rc.order_ids.long()[ids]  # storage slot -> the caller's index
        outs.append(dict(n=n, offs=rc.offs.clone(), ids=ids, render=rc.render.clone(), alphas=rc.alphas.clone(),
                         last=rc.last_ids.clone(), vm=g["viewmat"].clone(), means=g["means"].clone(),
                         colors=g["colors"].clone(), opac=g["opacities"].clone()))
    return rc, outs


def _same(a, b):
    assert a["n"] == b["n"] > 0
    assert torch.equal(a["offs"], b["offs"])
    assert torch.equal(a["ids"], b["ids"]), "the relabelled lists differ from the unpermuted run's lists"
    assert torch.equal(a["render"], b["render"]) and torch.equal(a["alphas"], b["alphas"])
    assert torch.equal(a["last"], b["last"])
    for k, tol in (("vm", 2e-5), ("means", 2e-4), ("colors", 2e-4), ("opac", 2e-4)):
        assert float((a[k] - b[k]).abs().max()) <= tol * float(a[k].abs().max()), k


@pytest.mark.parametrize("variant", ["wave_sort", "wg_sort", "sort_in_forward", "two_pass"])
def test_depth_ties_on_a_wall_keep_their_order_under_tile_order_placement(variant, monkeypatch):
    W, H, N, sigma = 320, 240, 40000, 1.2
    kw = {}
    if variant == "wave_sort":
        monkeypatch.setenv("GSL_DEV_TILE_SORT", "wave")
    elif variant == "wg_sort":
        monkeypatch.setenv("GSL_DEV_TILE_SORT", "wg")
    elif variant == "sort_in_forward":
        monkeypatch.setenv("GSLOC_SORT_IN_FORWARD", "force")
        kw["sort_in_forward"] = True
    elif variant == "two_pass":
        monkeypatch.setenv("GSLOC_BINNING", "two-pass")
    means, quats, scales, opac, sh, K = _wall(N, W, H, sigma)
    ins = [t.to(DEV) for t in (means, quats, scales, opac, sh)]
    K = K.to(DEV).contiguous()
    V = torch.eye(4, device=DEV)  # identity camera: all depths equal, bit for bit
    gen = torch.Generator().manual_seed(5)
    v = torch.randn(H, W, 4, generator=gen).to(DEV)
    va = torch.randn(H, W, 1, generator=gen).to(DEV)
    rc0, plain = _run(ins, V, K, W, H, v, va, False, **kw)
    rc1, placed = _run(ins, V, K, W, H, v, va, True, **kw)
    assert int(torch.unique(rc0.Q0[rc0.Q1[:, 3] > 0][:, 2]).numel()) == 1, "the wall must tie every depth"
    if variant == "sort_in_forward":
        assert rc0.sorts_in_forward() and rc1.sorts_in_forward()
    if variant == "two_pass":
        assert rc0.bins is None and rc1.bins is None
    for a, b in zip(plain, placed):
        _same(a, b)
    # the placement really is a tile order: consecutive storage slots sit in the same or the next tile
    t = torch.floor(rc1.Q0[:, 0] / 16) + rc1.tw * torch.floor(rc1.Q0[:, 1] / 16)
    vis = rc1.Q1[:, 3] > 0
    assert bool((t[vis][1:] >= t[vis][:-1]).all())
    # a moved camera (ties resolved by depth now) still matches
    V2 = torch.linalg.inv(small_pose(0.7, 0.02, dtype=torch.float32)).to(DEV).contiguous()
    for rc in (rc0, rc1):
        rc.forward(*ins, V2, K)
    torch.cuda.synchronize()
    assert torch.equal(rc0.render, rc1.render) and torch.equal(rc0.last_ids, rc1.last_ids)


def test_long_list_sorted_by_several_workgroups_under_tile_order_placement():
    """A pile on the wall: one tile list of > 4 x the mean and > 2048 entries goes through gsl_long_sort's merge passes,
    whose last pass relabels."""
    W, H, N = 320, 240, 30000
    means, quats, scales, opac, sh, K = _wall(N, W, H, 0.8, pile=6000)
    ins = [t.to(DEV) for t in (means, quats, scales, opac, sh)]
    K = K.to(DEV).contiguous()
    V = torch.eye(4, device=DEV)
    gen = torch.Generator().manual_seed(6)
    v = torch.randn(H, W, 4, generator=gen).to(DEV)
    va = torch.zeros(H, W, 1, device=DEV)
    rc0, plain = _run(ins, V, K, W, H, v, va, False)
    rc1, placed = _run(ins, V, K, W, H, v, va, True)
    assert rc0.long_min > 0 and rc1.long_min > 0, "the pile must switch the long-list split on"
    assert int((rc0.offs[1:] - rc0.offs[:-1]).max()) > 2048
    for a, b in zip(plain, placed):
        _same(a, b)


def test_placement_follows_new_or_modified_input_tensors():
    """The context's storage-order copies are re-gathered when the caller passes other tensors or writes in place."""
    W, H, N = 200, 160, 20000
    means, quats, scales, opac, sh, K = _wall(N, W, H, 1.0, seed=9)
    ins = [t.to(DEV) for t in (means, quats, scales, opac, sh)]
    K = K.to(DEV).contiguous()
    V = torch.linalg.inv(small_pose(0.4, 0.01, dtype=torch.float32)).to(DEV).contiguous()
    from gsplatloc_amd.context import RenderContext
    rcs = []
    for reorder in (False, True):
        rc = RenderContext(N, W, H, "ED", sh_degree=None, device=DEV, reorder=reorder)
        rc.calibrate(*ins[:4], None, V, K)
        rcs.append(rc)
    moved = [t.clone() for t in ins]
    moved[0][:, 0] += 0.01  # new tensors
    for rc in rcs:
        rc.forward(*moved[:4], None, V, K)
    torch.cuda.synchronize()
    assert torch.equal(rcs[0].render, rcs[1].render)
    moved[0][:, 1] -= 0.02  # same tensors, written in place
    for rc in rcs:
        rc.forward(*moved[:4], None, V, K)
    torch.cuda.synchronize()
    assert torch.equal(rcs[0].render, rcs[1].render)
    assert not torch.equal(rcs[1]._placed[0], ins[0][rcs[1]._perm64])


def test_autograd_through_a_placed_context_returns_gradients_in_the_callers_order():
    W, H, N = 200, 160, 20000
    means, quats, scales, opac, sh, K = _wall(N, W, H, 1.1, seed=11)
    K = K.to(DEV).contiguous()
    V = torch.linalg.inv(small_pose(0.4, 0.01, dtype=torch.float32)).to(DEV).contiguous()
    from gsplatloc_amd.context import RenderContext
    w = torch.randn(H, W, 4, generator=torch.Generator().manual_seed(1)).to(DEV)
    grads = []
    for reorder in (False, True):
        ins = [t.to(DEV).clone().requires_grad_() for t in (means, quats, scales, opac, sh)]
        Vg = V.clone().requires_grad_()
        rc = RenderContext(N, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=DEV, reorder=reorder)
        rc.calibrate(*[t.detach() for t in ins], V, K)
        render, alphas = rc.render_autograd(*ins, Vg, K)
        (render * w).sum().backward()
        torch.cuda.synchronize()
        grads.append([t.grad.clone() for t in ins] + [Vg.grad.clone()])
    for a, b in zip(*grads):  # (+1e-7: v_quats of isotropic splats is cancellation noise of ~1e-8)
        assert float((a - b).abs().max()) <= 2e-4 * float(a.abs().max()) + 1e-7


def test_zero_colour_gradients_are_not_stored_and_stale_ones_are_cleared():
    """gsl_fused_project_bwd(..., v_colors_state): under a depth-only upstream gradient (GsplatLoc's loss,
    /root/reference/src/my_gsplat/gs_trainer_total.py:111-123) no Gaussian has a colour gradient and the 48 bytes of zeros per
    Gaussian are not stored while the context's buffer is known to be zero.  The state must follow the buffer: a backward
    WITH a colour gradient writes it and marks the buffer dirty; the next depth-only backward must clear those values again."""
    from gsplatloc_amd.context import RenderContext
    W, H, N = 200, 160, 20000
    means, quats, scales, opac, sh, K = _wall(N, W, H, 1.1, seed=13)
    ins = [t.to(DEV) for t in (means, quats, scales, opac, sh)]
    K = K.to(DEV).contiguous()
    V = torch.linalg.inv(small_pose(0.4, 0.01, dtype=torch.float32)).to(DEV).contiguous()
    gen = torch.Generator().manual_seed(2)
    v_depth = torch.zeros(H, W, 4)
    v_depth[..., 3] = torch.randn(H, W, generator=gen)
    v_rgb = torch.randn(H, W, 4, generator=gen)
    v_depth, v_rgb = v_depth.to(DEV), v_rgb.to(DEV)
    va = torch.zeros(H, W, 1, device=DEV)
    rc = RenderContext(N, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=DEV, reorder=False)
    rc.calibrate(*ins, V, K)
    ref = RenderContext(N, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=DEV, reorder=False)
    ref.calibrate(*ins, V, K)
    ref.vc_state = None  # the reference context always stores every colour gradient

    def run(ctx, v):
        ctx.forward(*ins, V, K)
        g = ctx.backward(v, va)
        torch.cuda.synchronize()
        return {k: t.clone() for k, t in g.items()}

    assert int(rc.vc_state) == 1
    rc.v_colors.fill_(7.0)          # what a skipped store would leave behind if the state were wrong ...
    rc.vc_state.zero_()             # ... so tell the kernel the buffer is of unknown content
    a = run(rc, v_depth)
    assert float(a["colors"].abs().max()) == 0.0 and int(rc.vc_state) == 1   # everything stored once, then known zero
    rc.v_colors[5] = 3.0            # (a skipped store leaves this: proves the zero stores are really skipped now)
    b = run(rc, v_depth)
    assert float(b["colors"][5].abs().max()) == 3.0 and int(rc.vc_state) == 1
    rc.v_colors[5] = 0.0
    c, c_ref = run(rc, v_rgb), run(ref, v_rgb)
    assert float(c["colors"].abs().max()) > 0 and int(rc.vc_state) == 0      # dirty: a real colour gradient was written
    assert torch.equal(c["colors"] != 0, c_ref["colors"] != 0)
    assert float((c["colors"] - c_ref["colors"]).abs().max()) <= 2e-4 * float(c_ref["colors"].abs().max())
    d, d_ref = run(rc, v_depth), run(ref, v_depth)
    assert float(d["colors"].abs().max()) == 0.0, "stale colour gradients survived a depth-only backward"
    assert int(rc.vc_state) == 1
    for k in ("means", "scales", "opacities", "viewmat"):
        assert float((d[k] - d_ref[k]).abs().max()) <= 2e-4 * float(d_ref[k].abs().max()) + 1e-12, k


def test_strips_of_placed_gaussians_reproduce_the_full_frame():
    """The N > 1 path of bench.py / GraphTracker on one GPU: two tile-row strips, each keeping only the Gaussians within its
    guard band AND placing them in tile order, against the full frame rendered from the caller's (random) order: the
    strips' pixels bit-identical, the two pose gradients adding up to the full frame's."""
    from gsplatloc_amd.context import RenderContext
    from gsplatloc_amd.parallel import gaussians_for_strip
    W, H, N = 320, 240, 60000
    means, quats, scales, opac, sh, K = _wall(N, W, H, 1.2, seed=17)
    ins = [t.to(DEV) for t in (means, quats, scales, opac, sh)]
    K = K.to(DEV).contiguous()
    V = torch.linalg.inv(small_pose(0.5, 0.015, dtype=torch.float32)).to(DEV).contiguous()
    gen = torch.Generator().manual_seed(3)
    v = torch.randn(H, W, 4, generator=gen).to(DEV)
    va = torch.randn(H, W, 1, generator=gen).to(DEV)
    full = RenderContext(N, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=DEV, reorder=False, full_grads=False)
    full.calibrate(*ins, V, K)
    full.forward(*ins, V, K)
    g_full = full.backward(v, va, full=False)["viewmat"].clone()
    torch.cuda.synchronize()
    th = (H + 15) // 16
    total = torch.zeros_like(g_full)
    for rows in ((0, 6), (6, th)):
        idx = gaussians_for_strip(full.Q0[:, 0:2], full.radii, rows)
        loc = [t[idx].contiguous() for t in ins]
        rc = RenderContext(int(idx.numel()), W, H, "RGB+ED", sh_degree=1, K_sh=4, device=DEV, tile_rows=rows, reorder=True,
                           full_grads=False)
        rc.calibrate(*loc, V, K)
        assert rc.order_ids is not None
        rc.forward(*loc, V, K)
        g = rc.backward(v, va, full=False)["viewmat"].clone()
        torch.cuda.synchronize()
        rc.check_capacity()
        r0, r1 = rows[0] * 16, min(rows[1] * 16, H)
        assert torch.equal(rc.render[r0:r1], full.render[r0:r1]) and torch.equal(rc.alphas[r0:r1], full.alphas[r0:r1])
        total += g
    assert float((total - g_full).abs().max()) <= 2e-5 * float(g_full.abs().max())
