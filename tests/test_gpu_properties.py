"""Size-independent properties of the hot path at BASELINE.json's full sizes (R: 1 M Gaussians, 1200x680; X: 5 M,
1920x1080, fp16-staged records) -- checks that need no oracle run and hold for any correct implementation:

* conservation: the tile lists hold exactly the (Gaussian, tile) pairs the projection counted (sum of tiles_per_gauss =
  number of intersections = last offset), every Gaussian id in a list is a visible Gaussian, and a Gaussian appears in as
  many lists as it touches tiles;
* sortedness: every tile list is ascending in (depth bits, Gaussian id) -- the order gsplat's isect_tiles produces
  (IDX:14360) and the compositing loop relies on;
* idempotence: the forward has no atomics -- two passes give bit-identical images, and binned / two-pass binning agree;
* bounds: 0 <= alpha <= 1, the expected depth lies inside the depth range of the cloud, no NaN;
* linearity of the backward in the upstream gradient: g(a v1 + b v2) = a g(v1) + b g(v2) up to float32 summation order.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(N, W, H, staging):
    from gsplatloc_amd.context import RenderContext
    from gsplatloc_amd.synthetic import perturbed_pose, random_scene

    dev = torch.device("cuda")
    sc = random_scene(N, W, H, sigma_px=1.0, device=dev)
    V = torch.linalg.inv(perturbed_pose()).to(dev).contiguous()
    ctx = RenderContext(N, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=dev, full_grads=True, staging=staging)
    ctx.tiles_per_gauss = torch.zeros(N, dtype=torch.int32, device=dev)
    inp = (sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["sh"], V, sc["K"].contiguous())
    ctx.calibrate(*inp)
    return ctx, inp, sc


@pytest.mark.parametrize("N,W,H,staging", [(1_000_000, 1200, 680, "fp32"), (5_000_000, 1920, 1080, "fp16")])
def test_properties_at_full_size(N, W, H, staging):
    ctx, inp, sc = _setup(N, W, H, staging)
    dev = sc["means"].device
    ctx.forward(*inp)
    torch.cuda.synchronize()
    n = ctx.check_capacity()
    offs = ctx.offs.long()
    ids = ctx.flatten_ids[:n].long()
    # ---- conservation
    assert int(offs[0]) == 0 and int(offs[-1]) == n and bool((offs[1:] >= offs[:-1]).all())
    assert int(ctx.tiles_per_gauss.long().sum()) == n
    visible = ctx.radii.reshape(N, -1)[:, 0] > 0
    assert bool(visible[ids].all())
    per_gauss = torch.bincount(ids, minlength=N)
    assert torch.equal(per_gauss, ctx.tiles_per_gauss.long())
    # ---- sortedness: (depth bits, id) ascending inside every tile
    tile_of = torch.repeat_interleave(torch.arange(ctx.n_tiles, device=dev), offs[1:] - offs[:-1])
    depth_bits = ctx.Q0[:, 2].contiguous().view(torch.int32).long()[ids]
    # (random-order input: the context stores the Gaussians in tile order; the sort key's id is the CALLER's index)
    assert (ctx.order_ids is not None) == (N <= ctx.REORDER_MAX_N), "1 M randomly ordered Gaussians are placed in tile order"
    key = (depth_bits << 32) | (ctx.order_ids.long()[ids] if ctx.order_ids is not None else ids)
    same_tile = tile_of[1:] == tile_of[:-1]
    assert bool((key[1:][same_tile] > key[:-1][same_tile]).all())
    # ---- idempotence, and binned == two-pass lists / image
    r1, a1, l1 = ctx.render.clone(), ctx.alphas.clone(), ctx.last_ids.clone()
    ctx.forward(*inp)
    torch.cuda.synchronize()
    assert torch.equal(ctx.render, r1) and torch.equal(ctx.alphas, a1) and torch.equal(ctx.last_ids, l1)
    assert torch.equal(ctx.flatten_ids[:n].long(), ids)
    if ctx.bins is not None:
        keep = (ctx.bins, ctx.bin_cap)
        ctx.bins, ctx.bin_cap = None, 0
        ctx.forward(*inp)
        torch.cuda.synchronize()
        assert ctx.check_capacity() == n and torch.equal(ctx.flatten_ids[:n].long(), ids) and torch.equal(ctx.offs.long(), offs)
        assert torch.equal(ctx.render, r1)
        ctx.bins, ctx.bin_cap = keep
        ctx.ws.zero_()
    # ---- bounds
    assert bool(torch.isfinite(r1).all()) and float(a1.min()) >= 0.0 and float(a1.max()) <= 1.0
    z = ctx.Q0[:, 2][visible]
    covered = a1[..., 0] > 0.5
    d = r1[..., 3][covered]
    assert float(d.min()) >= float(z.min()) * (1 - 2e-3) and float(d.max()) <= float(z.max()) * (1 + 2e-3)
    assert float(r1[..., :3].min()) >= -1e-3
    # ---- linearity of the backward
    g = torch.Generator().manual_seed(11)
    v1 = torch.randn(H, W, 4, generator=g).to(dev)
    v2 = torch.randn(H, W, 4, generator=g).to(dev)
    va = torch.zeros(H, W, 1, device=dev)

    def grads(v):
        ctx.forward(*inp)
        out = ctx.backward(v, va, full=True)
        torch.cuda.synchronize()
        return {k: t.clone() for k, t in out.items() if t is not None}

    g1, g2, g12 = grads(v1), grads(v2), grads(0.5 * v1 - 2.0 * v2)
    for k in g12:
        want = 0.5 * g1[k] - 2.0 * g2[k]
        scale = float(want.abs().max()) + 1e-30
        if k == "quats":  # isotropic splats: cancellation noise
            continue
        assert float((g12[k] - want).abs().max()) <= 2e-4 * scale, k
