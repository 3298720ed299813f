"""Out-of-bounds write detector for the RenderContext launch sequence.

Every buffer the kernels may write is re-homed between two 64 KiB canary regions; after a few
forward/backward passes (general and tiny-splat backward, whole frame and tile strips, sizes that are not
multiples of the workgroup or tile size) the canaries must be untouched.  The caching allocator packs
tensors side by side, so without this an overrun would corrupt a neighbour silently.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

GUARD = 64 * 1024
CANARY = 0xA5

WRITABLE = ["radii", "Q0", "Q1", "Q2", "comps", "offs", "n_is", "ws", "render", "alphas", "last_ids", "vacc",
            "v_viewmat", "v_means", "v_quats", "v_scales", "v_opacities", "v_colors", "keys", "flatten_ids", "hits", "hit_counts", "trec",
            "vcT", "long_ws"]


def _rehome(ctx):
    homes = {}
    for name in WRITABLE:
        t = getattr(ctx, name, None)
        if t is None:
            continue
        nbytes = t.numel() * t.element_size()
        pad = (-nbytes) % 256
        raw = torch.full((GUARD + nbytes + pad + GUARD,), CANARY, dtype=torch.uint8, device=t.device)
        body = raw[GUARD:GUARD + nbytes].view(t.dtype).view(t.shape)
        body.copy_(t)
        setattr(ctx, name, body)
        homes[name] = (raw, nbytes)
    return homes


def _check(homes):
    bad = []
    for name, (raw, nbytes) in homes.items():
        lo = raw[:GUARD]
        hi = raw[GUARD + nbytes:]
        if not bool((lo == CANARY).all()):
            bad.append(f"{name}: write BEFORE the buffer ({int((lo != CANARY).sum())} bytes)")
        if not bool((hi == CANARY).all()):
            first = int((hi != CANARY).nonzero()[0])
            bad.append(f"{name}: write PAST the end (+{first} bytes, {int((hi != CANARY).sum())} bytes touched)")
    assert not bad, "; ".join(bad)


@pytest.mark.parametrize("N,W,H,sigma_px,rows,mode,full", [
    (50_000, 640, 480, 1.0, None, "RGB+ED", True),
    (50_001, 333, 217, 0.0, None, "RGB+ED", True),       # tiny-splat backward, ragged sizes
    (30_011, 640, 480, 2.5, (7, 19), "RGB+ED", True),     # strip, general backward
    (30_011, 640, 480, 0.0, (0, 11), "ED", False),        # strip at the top edge, tiny, pose-only
    (30_011, 640, 470, 1.0, (19, 30), "ED", False),       # strip at the bottom edge, partial last tile row
    (200_000, 1200, 680, 1.0, (22, 43), "RGB+ED", True),  # the rank-1-of-2 strip of the benchmark frame
])
def test_no_write_outside_buffers(N, W, H, sigma_px, rows, mode, full, monkeypatch):
    from gsplatloc_amd.context import RenderContext
    from gsplatloc_amd.synthetic import perturbed_pose, random_scene

    dev = torch.device("cuda")
    if sigma_px == 0.0:  # the tiny-splat backward, which "auto" keeps for Gaussians in pixel order (these are not)
        monkeypatch.setenv("GSLOC_BWD", "tiny")
    sc = random_scene(N, W, H, sigma_px=sigma_px, device=dev)
    viewmat = torch.linalg.inv(perturbed_pose()).to(dev).contiguous()
    K = sc["K"].contiguous()
    ctx = RenderContext(N, W, H, mode, sh_degree=1, K_sh=4, device=dev, tile_rows=rows, full_grads=full)
    inp = (sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["sh"], viewmat, K)
    ctx.calibrate(*inp, headroom=1.05)
    assert ctx.tiny == (sigma_px == 0.0)
    homes = _rehome(ctx)
    g = torch.Generator().manual_seed(3)
    v = torch.randn(H, W, ctx.D, generator=g).to(dev)
    va = torch.randn(H, W, 1, generator=g).to(dev)
    for _ in range(3):
        ctx.forward(*inp)
        ctx.backward(v, va, full=full)
    torch.cuda.synchronize()
    ctx.check_capacity()
    _check(homes)
    assert torch.isfinite(ctx.v_viewmat).all()


def test_no_write_outside_buffers_with_a_long_tile_list():
    """The same canaries around a frame whose invalid points pile up in one tile (long-list split active)."""
    from gsplatloc_amd.context import RenderContext
    from gsplatloc_amd.synthetic import depth_frame_scene

    dev = torch.device("cuda")
    W, H = 640, 480
    sc = depth_frame_scene(W, H, stride=1, holes=True, device=dev, pile=True)
    N = sc["means"].shape[0]
    ctx = RenderContext(N, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=dev, full_grads=True)
    inp = (sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["sh"], sc["viewmat"], sc["K"].contiguous())
    ctx.calibrate(*inp, headroom=1.05)
    assert ctx.long_min > 0
    homes = _rehome(ctx)
    g = torch.Generator().manual_seed(3)
    v = torch.randn(H, W, ctx.D, generator=g).to(dev)
    va = torch.randn(H, W, 1, generator=g).to(dev)
    for _ in range(3):
        ctx.forward(*inp)
        ctx.backward(v, va, full=True)
    torch.cuda.synchronize()
    ctx.check_capacity()
    _check(homes)
    assert torch.isfinite(ctx.v_viewmat).all()


@pytest.mark.parametrize("sigma_px,mode,bwd", [(1.0, "RGB+ED", "auto"), (1.0, "ED", "auto"), (0.0, "RGB+ED", "tiny"),
                                               (0.0, "ED", "tiny"), (1.0, "RGB", "auto")])
def test_results_do_not_depend_on_stale_lds(sigma_px, mode, bwd, monkeypatch):
    """The compositing kernels' straight-line trips let a lane without a candidate read SOME staged slot and multiply
    it by an exact zero -- so every slot must hold finite numbers written in the same batch.  LDS is filled with NaNs
    (then with +Inf, then with zeros) by gsl_dev_poison_lds before each pass: the images must be bit-identical and the
    gradients finite and equal up to the order of the float atomics."""
    from gsplatloc_amd._lib import check, current_stream, load_library
    from gsplatloc_amd.context import RenderContext
    from gsplatloc_amd.synthetic import perturbed_pose, random_scene

    dev = torch.device("cuda")
    monkeypatch.setenv("GSLOC_BWD", bwd)
    N, W, H = 40_003, 333, 217
    sc = random_scene(N, W, H, sigma_px=sigma_px, device=dev)
    viewmat = torch.linalg.inv(perturbed_pose()).to(dev).contiguous()
    ctx = RenderContext(N, W, H, mode, sh_degree=1, K_sh=4, device=dev, full_grads=True)
    inp = (sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["sh"], viewmat, sc["K"].contiguous())
    ctx.calibrate(*inp)
    assert ctx.tiny == (bwd == "tiny")
    g = torch.Generator().manual_seed(5)
    v = torch.randn(H, W, ctx.D, generator=g).to(dev)
    va = torch.randn(H, W, 1, generator=g).to(dev)
    lib = load_library()
    out = []
    for pattern in (0xFFFFFFFF, 0x7F800000, 0):
        check(lib.gsl_dev_poison_lds(pattern, current_stream()), "gsl_dev_poison_lds")
        ctx.forward(*inp)
        check(lib.gsl_dev_poison_lds(pattern, current_stream()), "gsl_dev_poison_lds")
        grads = ctx.backward(v, va, full=True)
        torch.cuda.synchronize()
        ctx.check_capacity()
        out.append((ctx.render.clone(), ctx.alphas.clone(), {k: t.clone() for k, t in grads.items() if t is not None}))
    for r, a, gr in out[:2]:
        assert torch.equal(r, out[2][0]) and torch.equal(a, out[2][1])
        for k, t in gr.items():
            assert torch.isfinite(t).all(), k
            if k == "quats":  # (isotropic splats: this gradient is cancellation noise around 1e-7, atomics order and all)
                continue
            scale = float(out[2][2][k].abs().max()) + 1e-30
            assert float((t - out[2][2][k]).abs().max()) <= 1e-4 * scale, k


def test_long_list_results_do_not_depend_on_stale_lds():
    """The same with a pile (long-list split: k_long_fwd passes A / B, the segment backward)."""
    from gsplatloc_amd._lib import check, current_stream, load_library
    from gsplatloc_amd.context import RenderContext
    from gsplatloc_amd.synthetic import depth_frame_scene

    dev = torch.device("cuda")
    W, H = 640, 480
    sc = depth_frame_scene(W, H, stride=1, holes=True, device=dev, pile=True)
    N = sc["means"].shape[0]
    ctx = RenderContext(N, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=dev, full_grads=False)
    inp = (sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["sh"], sc["viewmat"], sc["K"].contiguous())
    ctx.calibrate(*inp)
    assert ctx.long_min > 0
    g = torch.Generator().manual_seed(5)
    v = torch.randn(H, W, ctx.D, generator=g).to(dev)
    va = torch.randn(H, W, 1, generator=g).to(dev)
    lib = load_library()
    out = []
    for pattern in (0xFFFFFFFF, 0):
        check(lib.gsl_dev_poison_lds(pattern, current_stream()), "gsl_dev_poison_lds")
        ctx.forward(*inp)
        check(lib.gsl_dev_poison_lds(pattern, current_stream()), "gsl_dev_poison_lds")
        gv = ctx.backward(v, va, full=False)["viewmat"].clone()
        torch.cuda.synchronize()
        ctx.check_capacity()
        out.append((ctx.render.clone(), ctx.alphas.clone(), gv))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    assert torch.isfinite(out[0][2]).all()
    assert float((out[0][2] - out[1][2]).abs().max()) <= 1e-4 * float(out[1][2].abs().max())


@pytest.mark.parametrize("sigma_px,mode", [(0.0, "RGB+ED"), (1.0, "RGB+ED"), (1.0, "ED")])
def test_no_write_outside_buffers_when_the_forward_sorts(sigma_px, mode, monkeypatch):
    """The canaries around a context whose compositing forward sorts its own tile bins (sort_in_forward: the tracker's
    set-up) -- tile_offsets, flatten_ids, n_isects and the flags are then outputs of that kernel."""
    from gsplatloc_amd.context import RenderContext
    from gsplatloc_amd.synthetic import perturbed_pose, random_scene

    dev = torch.device("cuda")
    if sigma_px == 0.0:
        monkeypatch.setenv("GSLOC_BWD", "tiny")
    N, W, H = 50_001, 333, 217
    sc = random_scene(N, W, H, sigma_px=sigma_px, device=dev)
    viewmat = torch.linalg.inv(perturbed_pose()).to(dev).contiguous()
    ctx = RenderContext(N, W, H, mode, sh_degree=1, K_sh=4, device=dev, full_grads=True, sort_in_forward=True)
    inp = (sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["sh"], viewmat, sc["K"].contiguous())
    ctx.calibrate(*inp, headroom=1.05)
    assert ctx.sorts_in_forward()
    homes = _rehome(ctx)
    g = torch.Generator().manual_seed(3)
    v = torch.randn(H, W, ctx.D, generator=g).to(dev)
    va = torch.randn(H, W, 1, generator=g).to(dev)
    for _ in range(3):
        ctx.forward(*inp)
        ctx.backward(v, va, full=True)
    ctx.forward(*inp)  # (and one nobody back-propagates)
    ctx.forward(*inp)
    ctx.backward(v, va, full=True)
    torch.cuda.synchronize()
    ctx.check_capacity()
    _check(homes)
    assert torch.isfinite(ctx.v_viewmat).all()
