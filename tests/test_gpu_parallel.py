"""GPU (one device, ranks emulated in sequence): strip + halo rendering with pre-bucketed Gaussians and
the shared loss reproduces the single-GPU tracker loss and pose gradient -- with a whole halo tile
row composited, and with only the ONE halo pixel row the 3x3 Sobel needs (what GraphTracker does)."""
import pytest
import torch

from gsplatloc_amd.synthetic import frame_pair

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("pixel_halo", [False, True])
def test_emulated_ranks_match_single_gpu(pixel_halo):
    import gsplatloc_amd.my_gsplat as M
    from gsplatloc_amd.context import RenderContext
    from gsplatloc_amd.parallel import gaussians_for_strip, halo_pixel_rows, halo_rows, strip_rows, strip_tracking_loss
    from oracle import tracker_oracle as T

    W, H = 160, 120
    fp = frame_pair(W, H, rot_deg=0.3, trans=0.01)
    K = fp["K"].to(DEV)
    pts0 = T.depth_to_points(fp["depth0"], fp["K"]).to(DEV)
    pts1 = T.depth_to_points(fp["depth1"], fp["K"]).to(DEV)
    N = pts0.shape[0]
    scales = torch.full((N, 3), 2e-3, device=DEV)
    quats = torch.tensor([1.0, 0, 0, 0], device=DEV).repeat(N, 1)
    opac = torch.ones(N, device=DEV)
    sh = torch.zeros(N, 4, 3, device=DEV)
    sh[:, 0] = M.rgb_to_sh(fp["rgb"].to(DEV))
    gt = M.compute_depth_gt(pts1, fp["rgb"].to(DEV), K[None], torch.eye(4, device=DEV)[None], H, W)[None, ..., None]
    viewmat = torch.linalg.inv(fp["c2w0"]).to(DEV).contiguous()
    trk = M.PoseTracker()

    def run(rows_render, rows_own, idx):
        sub = [t[idx].contiguous() for t in (pts0, quats, scales, opac, sh)] if idx is not None else [pts0, quats, scales, opac, sh]
        px = halo_pixel_rows(rows_own, H) if (pixel_halo and rows_own is not None) else None
        rc = RenderContext(sub[0].shape[0], W, H, "RGB+ED", sh_degree=1, K_sh=4, device=DEV, tile_rows=rows_render,
                           pixel_rows=px, full_grads=False)
        rc.calibrate(*sub, viewmat, K)
        Vg = viewmat.clone().requires_grad_()
        render, _ = rc.render_autograd(*sub, Vg, K)
        depths = render[None, ..., 3:4]
        if rows_own is None:
            total, _, _ = trk.tracking_loss(depths, gt)
        else:
            total, _, _ = strip_tracking_loss(depths, gt, rows_own, H)
        total.backward()
        rc.check_capacity()
        return float(total.detach()), Vg.grad.clone(), rc

    L_full, g_full, rc_full = run(None, None, None)
    th = (H + 15) // 16
    for world in (2, 4):
        strips = strip_rows(rc_full.offs, rc_full.tw, rc_full.th, world)
        L_sum, g_sum, n_sum = 0.0, torch.zeros_like(g_full), 0
        for rows in strips:
            hr = halo_rows(rows, th)
            idx = gaussians_for_strip(rc_full.Q0[:, 0:2], rc_full.radii, hr)
            L, g, _ = run(hr, rows, idx)
            L_sum += L
            g_sum += g
            n_sum += idx.numel()
        assert abs(L_sum - L_full) < 1e-5 * abs(L_full)
        assert float((g_sum - g_full).abs().max() / g_full.abs().max()) < 1e-4
        assert n_sum < world * N  # every bucket is a strict subset (strip + halo + guard band)
