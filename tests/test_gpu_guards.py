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
