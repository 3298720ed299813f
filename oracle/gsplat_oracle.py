"""Pure-PyTorch CPU restatement of the gsplat 1.3.0 operators GsplatLoc calls.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  PARITY UNPINNED: no
reference binary, fixture or golden vector exists for this path; the algorithm
is restated from the published gsplat 1.3.0 kernels (``gsplat/cuda/csrc``) and
the operator signatures that survive in the reference's symbol index
(``/root/reference/.vscode/PythonImportHelper-v2-Completion.json``, "IDX").

Every function is dtype-generic (float32 or float64) and differentiable through
plain autograd, so the HIP kernels' hand-derived backward passes can be checked
against gradients that were *not* derived by hand.

Reference call sites this file stands in for:
  * ``gsplat.rasterization``      /root/reference/src/my_gsplat/model.py:195-213
                                  /root/reference/src/my_gsplat/geometry.py:117-132
  * operator signatures           IDX:14351 (fully_fused_projection), IDX:14360
                                  (isect_tiles), IDX:14369 (isect_offset_encode),
                                  IDX:14378 (rasterize_to_pixels), IDX:14306
                                  (spherical_harmonics), IDX:14954 (rasterization)
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

# ---- named constants of the restated algorithm -------
# gsplat spells the two alpha thresholds as float32 literals (0.999f, 1.f / 255.f): the float32-representable values
# are used at every precision, so that 1 - alpha at the clamp is 1 - 0.999f as in every float32 implementation
ALPHA_MAX = 0.9990000128746033  # float(np.float32(0.999)): alpha clamp in compositing         (A.3)
ALPHA_MIN = 0.003921568859368563  # float(np.float32(1) / np.float32(255)): contribution cut-off (A.3)
T_STOP = 1e-4  # transmittance early-stop threshold (exclusive)      (A.3)
RADIUS_LAMBDA_FLOOR = 0.01  # max(0.01, b^2-det) under the sqrt       (A.1.6)
FOV_LIM_FACTOR = 1.3  # symmetric frustum clamp of the EWA Jacobian   (A.1.3)
ED_ALPHA_CLAMP = 1e-10  # expected-depth normalisation clamp          (A.3)
SH_C0 = 0.2820947917738781
SH_C1 = 0.48860251190292


# ------------------------------------------------------------------ projection
def quat_to_rotmat(quats: Tensor) -> Tensor:
    """wxyz quaternion (normalised inside) -> rotation matrix [...,3,3]. (A.1.2)"""
    q = quats / quats.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    R = torch.stack(
        [
            1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
            2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
            2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y),
        ],
        dim=-1,
    )
    return R.reshape(quats.shape[:-1] + (3, 3))


def quat_scale_to_covar(quats: Tensor, scales: Tensor) -> Tensor:
    """Sigma = (R S)(R S)^T   (IDX:14315 quat_scale_to_covar_preci, covar only)."""
    R = quat_to_rotmat(quats)
    M = R * scales[..., None, :]
    return M @ M.transpose(-1, -2)


def world_to_cam(means: Tensor, covars: Tensor, viewmat: Tensor) -> Tuple[Tensor, Tensor]:
    """mu_c = R mu + t ; Sigma_c = R Sigma R^T   (IDX:14342), one camera."""
    R = viewmat[:3, :3]
    t = viewmat[:3, 3]
    means_c = means @ R.T + t
    covars_c = R @ covars @ R.T
    return means_c, covars_c


def persp_proj(
    means_c: Tensor, covars_c: Tensor, K: Tensor, width: int, height: int
) -> Tuple[Tensor, Tensor]:
    """Perspective EWA projection (IDX:14333 proj, ortho=False).  (A.1.3)"""
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    x, y, z = means_c.unbind(-1)
    tan_fovx = 0.5 * width / fx
    tan_fovy = 0.5 * height / fy
    lim_x = FOV_LIM_FACTOR * tan_fovx
    lim_y = FOV_LIM_FACTOR * tan_fovy
    rz = 1.0 / z
    rz2 = rz * rz
    tx = z * torch.minimum(lim_x, torch.maximum(-lim_x, x * rz))
    ty = z * torch.minimum(lim_y, torch.maximum(-lim_y, y * rz))
    O = torch.zeros_like(z)
    J = torch.stack(
        [fx * rz, O, -fx * tx * rz2, O, fy * rz, -fy * ty * rz2], dim=-1
    ).reshape(means_c.shape[:-1] + (2, 3))
    cov2d = J @ covars_c @ J.transpose(-1, -2)
    mean2d = torch.stack([fx * x * rz + cx, fy * y * rz + cy], dim=-1)
    return mean2d, cov2d


def _project_valid(means, quats, scales, viewmat, K, width, height, eps2d):
    """Differentiable part of the fused projection for Gaussians known valid."""
    covars = quat_scale_to_covar(quats, scales)
    means_c, covars_c = world_to_cam(means, covars, viewmat)
    mean2d, cov2d = persp_proj(means_c, covars_c, K, width, height)
    det_orig = cov2d[..., 0, 0] * cov2d[..., 1, 1] - cov2d[..., 0, 1] * cov2d[..., 1, 0]
    a = cov2d[..., 0, 0] + eps2d
    c = cov2d[..., 1, 1] + eps2d
    b = cov2d[..., 0, 1]
    det = a * c - b * cov2d[..., 1, 0]
    comp = torch.sqrt(torch.clamp(det_orig / det, min=0.0))
    conics = torch.stack([c / det, -b / det, a / det], dim=-1)
    return mean2d, means_c[..., 2], conics, comp, (a, b, c, det)


def fully_fused_projection(
    means: Tensor,  # [N,3]
    quats: Tensor,  # [N,4] wxyz
    scales: Tensor,  # [N,3]
    viewmats: Tensor,  # [C,4,4] world->camera
    Ks: Tensor,  # [C,3,3]
    width: int,
    height: int,
    eps2d: float = 0.3,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    radius_clip: float = 0.0,
    calc_compensations: bool = False,
):
    """gsplat.fully_fused_projection, packed=False (IDX:14351).  (A.1)

    Returns radii[C,N] int32, means2d[C,N,2], depths[C,N], conics[C,N,3],
    compensations[C,N] or None.  Entries with radii==0 are zero.
    """
    C, N = viewmats.shape[0], means.shape[0]
    dt, dev = means.dtype, means.device
    radii_all = torch.zeros(C, N, dtype=torch.int32, device=dev)
    m2_all, d_all, con_all, comp_all = [], [], [], []
    for ci in range(C):
        V, K = viewmats[ci], Ks[ci]
        with torch.no_grad():
            zc = (means @ V[:3, :3].T + V[:3, 3])[:, 2]
            ok = (zc >= near_plane) & (zc <= far_plane)
            idx = ok.nonzero(as_tuple=True)[0]
            m2, dep, con, comp, (a, b, c, det) = _project_valid(
                means[idx], quats[idx], scales[idx], V, K, width, height, eps2d
            )
            ok2 = det > 0
            bb = 0.5 * (a + c)
            v1 = bb + torch.sqrt(torch.clamp(bb * bb - det, min=RADIUS_LAMBDA_FLOOR))
            radius = torch.ceil(3.0 * torch.sqrt(v1))
            ok2 &= radius > radius_clip
            ok2 &= ~(
                (m2[:, 0] + radius <= 0)
                | (m2[:, 0] - radius >= width)
                | (m2[:, 1] + radius <= 0)
                | (m2[:, 1] - radius >= height)
            )
            idx = idx[ok2]
            radii_all[ci, idx] = radius[ok2].to(torch.int32)
        m2, dep, con, comp, _ = _project_valid(
            means[idx], quats[idx], scales[idx], V, K, width, height, eps2d
        )
        m2_all.append(torch.zeros(N, 2, dtype=dt, device=dev).index_put((idx,), m2))
        d_all.append(torch.zeros(N, dtype=dt, device=dev).index_put((idx,), dep))
        con_all.append(torch.zeros(N, 3, dtype=dt, device=dev).index_put((idx,), con))
        comp_all.append(torch.zeros(N, dtype=dt, device=dev).index_put((idx,), comp))
    comps = torch.stack(comp_all) if calc_compensations else None
    return radii_all, torch.stack(m2_all), torch.stack(d_all), torch.stack(con_all), comps


# --------------------------------------------------------------------- binning
def _tile_bbox(means2d: Tensor, radii: Tensor, tile_size: int, tile_w: int, tile_h: int):
    """Inclusive-min / exclusive-max tile rectangle of each Gaussian. (A.2)"""
    ts = float(tile_size)
    m = means2d.detach().to(torch.float32)
    r = radii.to(torch.float32)
    tr = r / ts
    txc = m[..., 0] / ts
    tyc = m[..., 1] / ts
    # (uint32_t)floor(negative) saturates to 0 on the GPU; max(0, .) then min(., tile_{w,h})
    xmin = torch.clamp(torch.floor(txc - tr), 0, tile_w).to(torch.int64)
    ymin = torch.clamp(torch.floor(tyc - tr), 0, tile_h).to(torch.int64)
    xmax = torch.clamp(torch.ceil(txc + tr), 0, tile_w).to(torch.int64)
    ymax = torch.clamp(torch.ceil(tyc + tr), 0, tile_h).to(torch.int64)
    return xmin, ymin, xmax, ymax


def isect_tiles(
    means2d: Tensor,  # [C,N,2]
    radii: Tensor,  # [C,N]
    depths: Tensor,  # [C,N]
    tile_size: int,
    tile_width: int,
    tile_height: int,
    sort: bool = True,
):
    """gsplat.isect_tiles (IDX:14360): tiles_per_gauss[C,N] i32, isect_ids[I] i64,
    flatten_ids[I] i32.  Key = cam | tile_id | float32 bits of depth. (A.2)"""
    C, N = radii.shape
    dev = radii.device
    xmin, ymin, xmax, ymax = _tile_bbox(means2d, radii, tile_size, tile_width, tile_height)
    cnt = (xmax - xmin) * (ymax - ymin)
    cnt = torch.where(radii > 0, cnt, torch.zeros_like(cnt))
    tiles_per_gauss = cnt.to(torch.int32)
    n_tiles = tile_width * tile_height
    tile_n_bits = int(math.floor(math.log2(n_tiles))) + 1
    flat_cnt = cnt.reshape(-1)
    total = int(flat_cnt.sum())
    # emit order: by flattened (camera, gaussian) index, row-major over the tile rectangle
    owner = torch.repeat_interleave(torch.arange(C * N, device=dev), flat_cnt)
    start = torch.cumsum(flat_cnt, 0) - flat_cnt
    k = torch.arange(total, device=dev) - start[owner]
    wdt = (xmax - xmin).reshape(-1)[owner]
    ty = ymin.reshape(-1)[owner] + k // torch.clamp(wdt, min=1)
    tx = xmin.reshape(-1)[owner] + k % torch.clamp(wdt, min=1)
    tile_id = ty * tile_width + tx
    cam = owner // N
    dbits = depths.detach().to(torch.float32).reshape(-1)[owner].contiguous().view(torch.int32).to(torch.int64)
    isect_ids = (cam << (32 + tile_n_bits)) | (tile_id << 32) | (dbits & 0xFFFFFFFF)
    flatten_ids = owner.to(torch.int32)
    if sort:
        order = torch.sort(isect_ids, stable=True).indices
        isect_ids = isect_ids[order]
        flatten_ids = flatten_ids[order]
    return tiles_per_gauss, isect_ids, flatten_ids


def isect_offset_encode(isect_ids: Tensor, n_cameras: int, tile_width: int, tile_height: int) -> Tensor:
    """gsplat.isect_offset_encode (IDX:14369): start index of every (camera, tile). (A.2)"""
    n_tiles = tile_width * tile_height
    tile_n_bits = int(math.floor(math.log2(n_tiles))) + 1
    cam = isect_ids >> (32 + tile_n_bits)
    tile = (isect_ids >> 32) & ((1 << tile_n_bits) - 1)
    flat = cam * n_tiles + tile
    q = torch.arange(n_cameras * n_tiles, device=isect_ids.device)
    offsets = torch.searchsorted(flat.contiguous(), q, right=False)
    return offsets.to(torch.int32).reshape(n_cameras, tile_height, tile_width)


# ----------------------------------------------------------------- compositing
def rasterize_to_pixels(
    means2d: Tensor,  # [C,N,2]
    conics: Tensor,  # [C,N,3]
    colors: Tensor,  # [C,N,D]
    opacities: Tensor,  # [C,N]
    image_width: int,
    image_height: int,
    tile_size: int,
    isect_offsets: Tensor,  # [C,th,tw]
    flatten_ids: Tensor,  # [I]
    backgrounds: Optional[Tensor] = None,  # [C,D]
) -> Tuple[Tensor, Tensor]:
    """gsplat.rasterize_to_pixels (IDX:14378), front-to-back alpha compositing. (A.3)

    Vectorised per tile: [pixels, list] alpha matrix + cumulative product.  The
    cumulative product visits splats in list order, so the transmittance sequence
    is the sequential one of the kernel (``oracle/sequential.py`` checks this).
    """
    C, N, D = colors.shape
    th, tw = isect_offsets.shape[1:]
    dt, dev = means2d.dtype, means2d.device
    n_isects = flatten_ids.shape[0]
    offs = isect_offsets.reshape(-1).tolist() + [n_isects]
    m2 = means2d.reshape(C * N, 2)
    con = conics.reshape(C * N, 3)
    col = colors.reshape(C * N, D)
    opa = opacities.reshape(C * N)
    fid = flatten_ids.to(torch.int64)
    out_c = torch.zeros(C, image_height, image_width, D, dtype=dt, device=dev)
    out_a = torch.zeros(C, image_height, image_width, 1, dtype=dt, device=dev)
    rows_c, rows_a = [], []
    for ci in range(C):
        tiles_c = [[None] * tw for _ in range(th)]
        tiles_a = [[None] * tw for _ in range(th)]
        for tyi in range(th):
            y0, y1 = tyi * tile_size, min((tyi + 1) * tile_size, image_height)
            for txi in range(tw):
                x0, x1 = txi * tile_size, min((txi + 1) * tile_size, image_width)
                t = (ci * th + tyi) * tw + txi
                s, e = offs[t], offs[t + 1]
                ph, pw = y1 - y0, x1 - x0
                if e <= s:
                    tiles_c[tyi][txi] = torch.zeros(ph, pw, D, dtype=dt, device=dev)
                    tiles_a[tyi][txi] = torch.zeros(ph, pw, 1, dtype=dt, device=dev)
                    continue
                g = fid[s:e]
                py, px = torch.meshgrid(
                    torch.arange(y0, y1, device=dev, dtype=dt) + 0.5,
                    torch.arange(x0, x1, device=dev, dtype=dt) + 0.5,
                    indexing="ij",
                )
                px, py = px.reshape(-1, 1), py.reshape(-1, 1)
                dx = m2[g, 0][None] - px
                dy = m2[g, 1][None] - py
                cg = con[g]
                sigma = 0.5 * (cg[:, 0] * dx * dx + cg[:, 2] * dy * dy) + cg[:, 1] * dx * dy
                alpha = torch.clamp(opa[g][None] * torch.exp(-sigma), max=ALPHA_MAX)
                valid = (sigma >= 0) & (alpha >= ALPHA_MIN)
                a_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
                Tincl = torch.cumprod(1 - a_eff, dim=1)
                incl = valid & (Tincl.detach() > T_STOP)
                Texcl = torch.cat([torch.ones_like(Tincl[:, :1]), Tincl[:, :-1]], dim=1)
                w = torch.where(incl, a_eff * Texcl, torch.zeros_like(a_eff))
                pix = w @ col[g]
                Tfin = torch.prod(torch.where(incl, 1 - a_eff, torch.ones_like(a_eff)), dim=1)
                tiles_c[tyi][txi] = pix.reshape(ph, pw, D)
                tiles_a[tyi][txi] = (1 - Tfin).reshape(ph, pw, 1)
        rows_c.append(torch.cat([torch.cat(r, dim=1) for r in tiles_c], dim=0))
        rows_a.append(torch.cat([torch.cat(r, dim=1) for r in tiles_a], dim=0))
    out_c = torch.stack(rows_c)
    out_a = torch.stack(rows_a)
    if backgrounds is not None:
        out_c = out_c + (1 - out_a) * backgrounds[:, None, None, :]
    return out_c, out_a


# ------------------------------------------------------------------------- SH
def _sh_bases(deg: int, dirs: Tensor) -> Tensor:
    """Real SH basis up to degree 4, gsplat's 'fast' evaluation order.  [...,(deg+1)^2]"""
    d = dirs / dirs.norm(dim=-1, keepdim=True)
    x, y, z = d.unbind(-1)
    out = [torch.full_like(x, SH_C0)]
    if deg >= 1:
        out += [-SH_C1 * y, SH_C1 * z, -SH_C1 * x]
    if deg >= 2:
        z2 = z * z
        fTmpB = -1.092548430592079 * z
        fC1 = x * x - y * y
        fS1 = 2 * x * y
        out += [
            0.5462742152960395 * fS1,
            fTmpB * y,
            0.9461746957575601 * z2 - 0.3153915652525201,
            fTmpB * x,
            0.5462742152960395 * fC1,
        ]
    if deg >= 3:
        fTmpC = -2.285228997322329 * z2 + 0.4570457994644658
        fTmpBb = 1.445305721320277 * z
        fC2 = x * fC1 - y * fS1
        fS2 = x * fS1 + y * fC1
        out += [
            -0.5900435899266435 * fS2,
            fTmpBb * fS1,
            fTmpC * y,
            z * (1.865881662950577 * z2 - 1.119528997770346),
            fTmpC * x,
            fTmpBb * fC1,
            -0.5900435899266435 * fC2,
        ]
    if deg >= 4:
        raise NotImplementedError("oracle covers SH degree <= 3")
    return torch.stack(out, dim=-1)


def spherical_harmonics(degree: int, dirs: Tensor, coeffs: Tensor, masks: Optional[Tensor] = None) -> Tensor:
    """gsplat.spherical_harmonics (IDX:14306): colours [...,3] from coeffs [...,K,3]."""
    K = (degree + 1) ** 2
    B = _sh_bases(degree, dirs)  # [...,K]
    col = (B[..., None] * coeffs[..., :K, :]).sum(-2)
    if masks is not None:
        col = torch.where(masks[..., None], col, torch.zeros_like(col))
    return col


# --------------------------------------------------------------- rasterization
def rasterization(
    means: Tensor,
    quats: Tensor,
    scales: Tensor,
    opacities: Tensor,
    colors: Tensor,
    viewmats: Tensor,
    Ks: Tensor,
    width: int,
    height: int,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    radius_clip: float = 0.0,
    eps2d: float = 0.3,
    sh_degree: Optional[int] = None,
    packed: bool = False,
    tile_size: int = 16,
    backgrounds: Optional[Tensor] = None,
    render_mode: str = "RGB",
    sparse_grad: bool = False,
    absgrad: bool = False,
    rasterize_mode: str = "classic",
    channel_chunk: int = 32,
) -> Tuple[Tensor, Tensor, Dict]:
    """gsplat.rasterization (IDX:14954), dense (packed=False) semantics.  """
    assert render_mode in ("RGB", "D", "ED", "RGB+D", "RGB+ED")
    C = viewmats.shape[0]
    radii, means2d, depths, conics, comps = fully_fused_projection(
        means, quats, scales, viewmats, Ks, width, height, eps2d, near_plane, far_plane,
        radius_clip, calc_compensations=(rasterize_mode == "antialiased"),
    )
    opac = opacities[None].expand(C, -1)
    if comps is not None:
        opac = opac * comps
    if sh_degree is None:
        cols = colors[None].expand(C, -1, -1) if colors.dim() == 2 else colors
    else:
        c2w = torch.linalg.inv(viewmats)
        dirs = means[None] - c2w[:, None, :3, 3]
        shs = colors[None].expand(C, -1, -1, -1) if colors.dim() == 3 else colors
        cols = spherical_harmonics(sh_degree, dirs, shs, masks=radii > 0)
        cols = torch.clamp_min(cols + 0.5, 0.0)
    if render_mode in ("RGB+D", "RGB+ED"):
        cols = torch.cat([cols, depths[..., None]], dim=-1)
        if backgrounds is not None:
            backgrounds = torch.cat([backgrounds, torch.zeros(C, 1, dtype=cols.dtype)], dim=-1)
    elif render_mode in ("D", "ED"):
        cols = depths[..., None]
        if backgrounds is not None:
            backgrounds = torch.zeros(C, 1, dtype=cols.dtype)
    tw = math.ceil(width / float(tile_size))
    th = math.ceil(height / float(tile_size))
    tiles_per_gauss, isect_ids, flatten_ids = isect_tiles(means2d, radii, depths, tile_size, tw, th)
    isect_offsets = isect_offset_encode(isect_ids, C, tw, th)
    rc, ra = rasterize_to_pixels(
        means2d, conics, cols, opac, width, height, tile_size, isect_offsets, flatten_ids, backgrounds
    )
    if render_mode in ("ED", "RGB+ED"):
        rc = torch.cat([rc[..., :-1], rc[..., -1:] / ra.clamp(min=ED_ALPHA_CLAMP)], dim=-1)
    meta = dict(
        radii=radii, means2d=means2d, depths=depths, conics=conics, opacities=opac,
        tile_width=tw, tile_height=th, tiles_per_gauss=tiles_per_gauss, isect_ids=isect_ids,
        flatten_ids=flatten_ids, isect_offsets=isect_offsets, width=width, height=height,
        tile_size=tile_size, n_cameras=C, colors=cols,
    )
    return rc, ra, meta
