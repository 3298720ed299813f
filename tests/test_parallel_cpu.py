"""CPU tests of the screen-tile parallel path: strip partition, Gaussian pre-bucketing,
loss sharing with a one-pixel halo, and the single 16-float all-reduce over a 2-rank gloo group.
The per-rank render here is the CPU oracle (test infrastructure); the product's ranks run the HIP
pipeline with tile_rows=... (GPU tests cover that part on one device)."""
import math
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gsplatloc_amd.parallel import all_reduce_pose, gaussians_for_strip, halo_rows, strip_rows, strip_tracking_loss
from oracle import gsplat_oracle as G
from oracle import tracker_oracle as T
from tests.scenes import random_scene, sh_from_rgb, small_pose


def test_strip_rows_balance_and_cover():
    tw, th = 10, 7
    counts = torch.tensor([[5] * tw, [0] * tw, [50] * tw, [5] * tw, [5] * tw, [20] * tw, [1] * tw]).reshape(-1)
    offs = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(counts, 0)]).to(torch.int32)
    for world in (1, 2, 3, 4, 8):
        rows = strip_rows(offs, tw, th, world)
        assert len(rows) == world and rows[0][0] == 0 and rows[-1][1] == th
        assert all(rows[i][1] == rows[i + 1][0] for i in range(world - 1))
        assert all(a <= b for a, b in rows)
    rows = strip_rows(offs, tw, th, 2)
    load = [int(counts.reshape(th, tw)[a:b].sum()) for a, b in rows]
    assert max(load) <= 0.75 * sum(load)
    assert halo_rows((2, 4), th) == (1, 5) and halo_rows((0, 7), th) == (0, 7)


def test_gaussians_for_strip_is_a_superset_of_what_the_strip_needs():
    sc = random_scene(3000, 160, 120, sigma_px=2.0, dtype=torch.float32)
    V = torch.linalg.inv(small_pose(0.5, 0.01, dtype=torch.float32))[None]
    radii, m2, dep, con, _ = G.fully_fused_projection(sc["means"], sc["quats"], sc["scales"], V, sc["K"][None], 160, 120)
    tw, th = 10, 8
    rows = (3, 5)
    idx = gaussians_for_strip(m2[0], radii[0], rows, guard_tiles=1)
    # every Gaussian whose tile rectangle reaches the strip must be kept
    xmin, ymin, xmax, ymax = G._tile_bbox(m2[0], radii[0], 16, tw, th)
    needed = ((radii[0] > 0) & (ymax > rows[0]) & (ymin < rows[1])).nonzero(as_tuple=True)[0]
    assert set(needed.tolist()) <= set(idx.tolist())
    assert idx.numel() < 0.7 * 3000


def _full_loss_and_grad(sc, sh, V, gt, W, H):
    Vg = V.clone().requires_grad_()
    r, a, _ = G.rasterization(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sh, Vg, sc["K"][None], W, H,
                              sh_degree=1, render_mode="RGB+ED")
    total, dl, sl = T.tracking_loss(r[..., 3:4], gt)
    total.backward()
    return float(total), Vg.grad.clone()


def _scene():
    W, H = 96, 80
    sc = random_scene(1500, W, H, sigma_px=2.0, opacity=(0.4, 0.9), dtype=torch.float64)
    sh = sh_from_rgb(sc["rgbs"])
    with torch.no_grad():
        gt, _, _ = G.rasterization(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sh,
                                   torch.eye(4, dtype=torch.float64)[None], sc["K"][None], W, H, sh_degree=1,
                                   render_mode="RGB+ED")
    V = torch.linalg.inv(small_pose(0.4, 0.01))[None]
    return sc, sh, V, gt[..., 3:4], W, H


def _rank_share(rank, world, sc, sh, V, gt, W, H):
    th = math.ceil(H / 16)
    per = math.ceil(th / world)
    rows = (min(rank * per, th), min((rank + 1) * per, th))
    Vg = V.clone().requires_grad_()
    r, a, _ = G.rasterization(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sh, Vg, sc["K"][None], W, H,
                              sh_degree=1, render_mode="RGB+ED")
    # a rank only has its strip + halo: blank everything else to prove nothing else is used
    hr = halo_rows(rows, th)
    depth = r[..., 3:4]
    keep = torch.zeros_like(depth)
    keep[:, hr[0] * 16:min(hr[1] * 16, H)] = 1.0
    total, dl, sl = strip_tracking_loss(depth * keep, gt, rows, H)
    if total.requires_grad:
        total.backward()
    g = Vg.grad if Vg.grad is not None else torch.zeros_like(V)
    return float(total), g


def test_strip_losses_and_gradients_add_up_single_process():
    sc, sh, V, gt, W, H = _scene()
    L, g = _full_loss_and_grad(sc, sh, V, gt, W, H)
    for world in (2, 3):
        parts = [_rank_share(r, world, sc, sh, V, gt, W, H) for r in range(world)]
        assert abs(sum(p[0] for p in parts) - L) < 1e-12 * max(1.0, abs(L)) + 1e-14
        gs = sum(p[1] for p in parts)
        assert torch.allclose(gs, g, rtol=1e-9, atol=1e-12 * float(g.abs().max()))


def test_strip_shares_of_the_normal_term_add_up():
    """The switched-off normal-consistency term is a mean of row-wise cosines: a rank owns its rows' cosines, and the
    shares (value and depth gradient) add up to the single-GPU term."""
    from gsplatloc_amd.synthetic import replica_intrinsics
    g = torch.Generator().manual_seed(4)
    H, W = 70, 40
    K = replica_intrinsics(W, H, dtype=torch.float64)
    gt = (torch.rand(1, H, W, 1, generator=g, dtype=torch.float64) * 2 + 1)
    depth0 = gt + 0.05 * torch.randn(1, H, W, 1, generator=g, dtype=torch.float64)
    depth0[:, 20:24, 5:12] = 0.0
    d = depth0.clone().requires_grad_()
    total, _, _ = T.tracking_loss(d, gt, 0.7, 0.1, K)
    total.backward()
    th = math.ceil(H / 16)
    for world in (2, 3):
        per = math.ceil(th / world)
        acc, gacc = 0.0, torch.zeros_like(depth0)
        for rank in range(world):
            rows = (min(rank * per, th), min((rank + 1) * per, th))
            dr = depth0.clone().requires_grad_()
            share, _, _ = strip_tracking_loss(dr, gt, rows, H, 0.7, 0.1, K=K)
            share.backward()
            acc += float(share)
            gacc += dr.grad
        assert acc == pytest.approx(float(total), rel=1e-12)
        assert torch.allclose(gacc, d.grad, rtol=1e-9, atol=1e-14)


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    sc, sh, V, gt, W, H = _scene()
    L, g = _rank_share(rank, world, sc, sh, V, gt, W, H)
    buf = torch.zeros(16, dtype=torch.float64)
    buf[:12] = g[0, :3].reshape(-1)
    buf[12] = L
    all_reduce_pose(buf)  # THE collective of the path: one sum of 16 floats
    if rank == 0:
        ret["buf"] = buf.clone()
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_allreduce_rebuilds_the_full_pose_gradient():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    sc, sh, V, gt, W, H = _scene()
    L, g = _full_loss_and_grad(sc, sh, V, gt, W, H)
    buf = ret["buf"]
    assert abs(float(buf[12]) - L) < 1e-12
    assert torch.allclose(buf[:12].reshape(3, 4), g[0, :3], rtol=1e-9, atol=1e-12 * float(g.abs().max()))
