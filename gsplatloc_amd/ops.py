"""Stage operators with the gsplat 1.3.0 signatures, backed by libgsloc_hip.

Mirrors ``gsplat/cuda/_wrapper.py`` of the gsplat fork GsplatLoc installs
(un-vendored; signatures from /root/reference/.vscode/PythonImportHelper-v2-Completion.json):
``fully_fused_projection`` (IDX:14351), ``isect_tiles`` (IDX:14360),
``isect_offset_encode`` (IDX:14369), ``rasterize_to_pixels`` (IDX:14378),
``spherical_harmonics`` (IDX:14306).  Same argument names, shapes, return
tuples and assertion-style error behaviour; tensors are fp32 on the HIP device.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _lib
from ._lib import check, current_stream, load_library, ptr

SUPPORTED_CHANNELS = (1, 2, 3, 4, 5, 8, 16, 32)


def _dev_f32(t: Tensor, name: str) -> Tensor:
    assert t.is_cuda, f"{name} must live on the GPU (got {t.device}); there is no CPU path"
    assert t.dtype == torch.float32, f"{name} must be float32 (got {t.dtype})"
    return t.contiguous()


# --------------------------------------------------------------------------- #
# projection
# --------------------------------------------------------------------------- #
class _FullyFusedProjection(torch.autograd.Function):
    """Projects Gaussians to 2D (autograd twin of gsplat's, IDX:14270)."""

    @staticmethod
    def forward(ctx, means, quats, scales, viewmats, Ks, width, height, eps2d, near_plane, far_plane,
                radius_clip, calc_compensations):
        lib = load_library()
        C, N = viewmats.shape[0], means.shape[0]
        dev = means.device
        radii = torch.empty(C, N, dtype=torch.int32, device=dev)
        means2d = torch.empty(C, N, 2, dtype=torch.float32, device=dev)
        depths = torch.empty(C, N, dtype=torch.float32, device=dev)
        conics = torch.empty(C, N, 3, dtype=torch.float32, device=dev)
        comps = torch.empty(C, N, dtype=torch.float32, device=dev) if calc_compensations else None
        st = current_stream()
        for c in range(C):
            check(lib.gsl_project_fwd(ptr(means), ptr(quats), ptr(scales), ptr(viewmats[c]), ptr(Ks[c]), N, width,
                                      height, eps2d, near_plane, far_plane, radius_clip, ptr(radii[c]),
                                      ptr(means2d[c]), ptr(depths[c]), ptr(conics[c]),
                                      ptr(comps[c]) if comps is not None else None, st), "gsl_project_fwd")
        ctx.save_for_backward(means, quats, scales, viewmats, Ks, radii, conics,
                              comps if comps is not None else torch.empty(0, device=dev))
        ctx.dims = (width, height, eps2d, calc_compensations)
        ctx.mark_non_differentiable(radii)
        if comps is None:
            return radii, means2d, depths, conics, None
        return radii, means2d, depths, conics, comps

    @staticmethod
    def backward(ctx, v_radii, v_means2d, v_depths, v_conics, v_comps):
        lib = load_library()
        means, quats, scales, viewmats, Ks, radii, conics, comps = ctx.saved_tensors
        width, height, eps2d, calc_comp = ctx.dims
        C, N = viewmats.shape[0], means.shape[0]
        dev = means.device
        need_full = any(ctx.needs_input_grad[:3])
        need_view = ctx.needs_input_grad[3]
        v_means2d = v_means2d.contiguous()
        v_depths = v_depths.contiguous()
        v_conics = v_conics.contiguous()
        if calc_comp and v_comps is not None:
            v_comps = v_comps.contiguous()
        else:
            v_comps = None
        v_means = v_quats = v_scales = None
        v_viewmats = torch.zeros(C, 4, 4, dtype=torch.float32, device=dev) if need_view else None
        ws_bytes = lib.gsl_project_bwd_ws_bytes(N)
        ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=dev)
        st = current_stream()
        for c in range(C):
            if need_full:
                vm = torch.empty(N, 3, dtype=torch.float32, device=dev)
                vq = torch.empty(N, 4, dtype=torch.float32, device=dev)
                vs = torch.empty(N, 3, dtype=torch.float32, device=dev)
            else:
                vm = vq = vs = None
            check(lib.gsl_project_bwd(
                ptr(means), ptr(quats), ptr(scales), ptr(viewmats[c]), ptr(Ks[c]), N, width, height, eps2d,
                ptr(radii[c]), ptr(conics[c]), ptr(comps[c]) if calc_comp else None, ptr(v_means2d[c]),
                ptr(v_depths[c]), ptr(v_conics[c]), ptr(v_comps[c]) if v_comps is not None else None, ptr(vm),
                ptr(vq), ptr(vs), ptr(v_viewmats[c]) if need_view else None, ptr(ws), ws_bytes, st),
                "gsl_project_bwd")
            if need_full:
                v_means = vm if v_means is None else v_means + vm
                v_quats = vq if v_quats is None else v_quats + vq
                v_scales = vs if v_scales is None else v_scales + vs
        ni = ctx.needs_input_grad
        return (v_means if ni[0] else None, v_quats if ni[1] else None, v_scales if ni[2] else None,
                v_viewmats, None, None, None, None, None, None, None, None)


def fully_fused_projection(
    means: Tensor,  # [N, 3]
    covars: Optional[Tensor],  # [N, 6] or None
    quats: Optional[Tensor],  # [N, 4] or None
    scales: Optional[Tensor],  # [N, 3] or None
    viewmats: Tensor,  # [C, 4, 4]
    Ks: Tensor,  # [C, 3, 3]
    width: int,
    height: int,
    eps2d: float = 0.3,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    radius_clip: float = 0.0,
    packed: bool = False,
    sparse_grad: bool = False,
    calc_compensations: bool = False,
) -> Tuple[Tensor, Tensor, Tensor, Tensor, Optional[Tensor]]:
    """Project Gaussians to the image plane: radii [C,N] int32, means2d [C,N,2],
    depths [C,N], conics [C,N,3], compensations [C,N] | None."""
    C = viewmats.size(0)
    N = means.size(0)
    assert means.size() == (N, 3), means.size()
    assert viewmats.size() == (C, 4, 4), viewmats.size()
    assert Ks.size() == (C, 3, 3), Ks.size()
    if covars is not None:
        raise NotImplementedError("precomputed covars are not supported; pass quats and scales (as GsplatLoc does)")
    assert quats is not None, "covars or quats is required"
    assert scales is not None, "covars or scales is required"
    assert quats.size() == (N, 4), quats.size()
    assert scales.size() == (N, 3), scales.size()
    if packed:
        raise NotImplementedError("packed=True is not supported (GsplatLoc runs packed=False, model.py:123)")
    if sparse_grad:
        raise NotImplementedError("sparse_grad requires packed=True")
    means, quats, scales = _dev_f32(means, "means"), _dev_f32(quats, "quats"), _dev_f32(scales, "scales")
    viewmats, Ks = _dev_f32(viewmats, "viewmats"), _dev_f32(Ks, "Ks")
    return _FullyFusedProjection.apply(means, quats, scales, viewmats, Ks, int(width), int(height), float(eps2d),
                                       float(near_plane), float(far_plane), float(radius_clip),
                                       bool(calc_compensations))


# --------------------------------------------------------------------------- #
# spherical harmonics
# --------------------------------------------------------------------------- #
class _SphericalHarmonics(torch.autograd.Function):
    @staticmethod
    def forward(ctx, degree, dirs, coeffs, masks):
        lib = load_library()
        M = dirs.numel() // 3
        K = coeffs.shape[-2]
        colors = torch.empty(dirs.shape, dtype=torch.float32, device=dirs.device)
        check(lib.gsl_sh_fwd(degree, ptr(dirs), ptr(coeffs), ptr(masks), M, K, ptr(colors), current_stream()),
              "gsl_sh_fwd")
        ctx.save_for_backward(dirs, coeffs, masks if masks is not None else torch.empty(0, device=dirs.device))
        ctx.meta = (degree, M, K, masks is not None)
        return colors

    @staticmethod
    def backward(ctx, v_colors):
        lib = load_library()
        dirs, coeffs, masks = ctx.saved_tensors
        degree, M, K, has_mask = ctx.meta
        v_colors = v_colors.contiguous()
        v_coeffs = torch.empty_like(coeffs)
        v_dirs = torch.empty_like(dirs) if ctx.needs_input_grad[1] else None
        check(lib.gsl_sh_bwd(degree, ptr(dirs), ptr(coeffs), ptr(masks) if has_mask else None, M, K, ptr(v_colors),
                             ptr(v_coeffs), ptr(v_dirs), current_stream()), "gsl_sh_bwd")
        return None, v_dirs, v_coeffs if ctx.needs_input_grad[2] else None, None


def spherical_harmonics(degrees_to_use: int, dirs: Tensor, coeffs: Tensor, masks: Optional[Tensor] = None) -> Tensor:
    """Colours [...,3] from SH coefficients [...,K,3] along (unnormalised) dirs [...,3]."""
    assert 0 <= degrees_to_use <= 3, "SH degree 0..3 is supported"
    assert (degrees_to_use + 1) ** 2 <= coeffs.shape[-2], coeffs.shape
    assert dirs.shape[:-1] == coeffs.shape[:-2], (dirs.shape, coeffs.shape)
    assert dirs.shape[-1] == 3, dirs.shape
    assert coeffs.shape[-1] == 3, coeffs.shape
    if masks is not None:
        assert masks.shape == dirs.shape[:-1], masks.shape
        masks = masks.to(torch.bool).contiguous().view(torch.uint8)
    return _SphericalHarmonics.apply(int(degrees_to_use), _dev_f32(dirs, "dirs"), _dev_f32(coeffs, "coeffs"), masks)


# --------------------------------------------------------------------------- #
# tile binning
# --------------------------------------------------------------------------- #
def tile_n_bits(n_tiles: int) -> int:
    return int(math.floor(math.log2(n_tiles))) + 1


@torch.no_grad()
def isect_tiles(
    means2d: Tensor,  # [C, N, 2]
    radii: Tensor,  # [C, N]
    depths: Tensor,  # [C, N]
    tile_size: int,
    tile_width: int,
    tile_height: int,
    sort: bool = True,
    packed: bool = False,
    n_cameras: Optional[int] = None,
    camera_ids: Optional[Tensor] = None,
    gaussian_ids: Optional[Tensor] = None,
) -> Tuple[Tensor, Tensor, Tensor]:
    """Map projected Gaussians to intersecting tiles.

    Returns tiles_per_gauss [C,N] int32, isect_ids [I] int64 (camera | tile | depth bits),
    flatten_ids [I] int32 (index into the flattened [C*N] arrays); sorted by isect_ids when
    ``sort`` (ties in Gaussian-index order, as a stable sort of the emit order gives)."""
    if packed:
        raise NotImplementedError("packed=True is not supported")
    C, N, _ = means2d.shape
    assert means2d.shape == (C, N, 2), means2d.size()
    assert radii.shape == (C, N), radii.size()
    assert depths.shape == (C, N), depths.size()
    lib = load_library()
    dev = means2d.device
    means2d, depths = _dev_f32(means2d, "means2d"), _dev_f32(depths, "depths")
    radii = radii.to(torch.int32).contiguous()
    n_tiles = tile_width * tile_height
    nbits = tile_n_bits(n_tiles)
    st = current_stream()
    tiles_per_gauss = torch.empty(C, N, dtype=torch.int32, device=dev)
    ws_bytes = lib.gsl_isect_ws_bytes(n_tiles)
    wss = [torch.empty(ws_bytes, dtype=torch.uint8, device=dev) for _ in range(C)]
    offs = torch.empty(C, n_tiles + 1, dtype=torch.int32, device=dev)
    counts = torch.empty(C, dtype=torch.int32, device=dev)
    for c in range(C):
        check(lib.gsl_isect_count(ptr(means2d[c]), ptr(radii[c]), N, tile_size, tile_width, tile_height, 0,
                                  tile_height, ptr(tiles_per_gauss[c]), ptr(offs[c]), ptr(counts[c:c + 1]),
                                  ptr(wss[c]), ws_bytes, st), "gsl_isect_count")
    per_cam = counts.tolist()  # host sync: output sizes depend on it (as in gsplat)
    total = int(sum(per_cam))
    isect_ids = torch.empty(total, dtype=torch.int64, device=dev)
    flatten_ids = torch.empty(total, dtype=torch.int32, device=dev)
    if total == 0:
        return tiles_per_gauss, isect_ids, flatten_ids
    base = 0
    if sort:
        keys = torch.empty(max(per_cam), dtype=torch.int64, device=dev)
        for c in range(C):
            n = per_cam[c]
            if n:
                check(lib.gsl_isect_fill(ptr(means2d[c]), ptr(radii[c]), ptr(depths[c]), N, tile_size, tile_width,
                                         tile_height, 0, tile_height, c, nbits, ptr(offs[c]), n, ptr(keys),
                                         ptr(flatten_ids[base:]), ptr(isect_ids[base:]), ptr(wss[c]), ws_bytes, st),
                      "gsl_isect_fill")
                if c:
                    flatten_ids[base:base + n] += c * N
            base += n
    else:
        cum = torch.cumsum(tiles_per_gauss.to(torch.int64), dim=1).contiguous()
        for c in range(C):
            n = per_cam[c]
            if n:
                check(lib.gsl_isect_emit(ptr(means2d[c]), ptr(radii[c]), ptr(depths[c]), ptr(cum[c]), N, tile_size,
                                         tile_width, tile_height, c, nbits, c * N, ptr(isect_ids[base:]),
                                         ptr(flatten_ids[base:]), st), "gsl_isect_emit")
            base += n
    return tiles_per_gauss, isect_ids, flatten_ids


@torch.no_grad()
def isect_offset_encode(isect_ids: Tensor, n_cameras: int, tile_width: int, tile_height: int) -> Tensor:
    """Start offset of every (camera, tile) in the sorted intersection list: [C, tile_height, tile_width] int32."""
    lib = load_library()
    n_tiles = tile_width * tile_height
    offsets = torch.empty(n_cameras, tile_height, tile_width, dtype=torch.int32, device=isect_ids.device)
    isect_ids = isect_ids.contiguous()
    check(lib.gsl_isect_offsets(ptr(isect_ids) if isect_ids.numel() else None, isect_ids.numel(), n_cameras, n_tiles,
                                tile_n_bits(n_tiles), ptr(offsets), current_stream()), "gsl_isect_offsets")
    return offsets


# --------------------------------------------------------------------------- #
# compositing
# --------------------------------------------------------------------------- #
def _pad_channels(D: int) -> int:
    for s in SUPPORTED_CHANNELS:
        if D <= s:
            return s
    raise AssertionError(f"Unsupported number of color channels: {D} (max {SUPPORTED_CHANNELS[-1]})")


class _RasterizeToPixels(torch.autograd.Function):
    """Rasterize gaussians (autograd twin of gsplat's, IDX:14279)."""

    @staticmethod
    def forward(ctx, means2d, conics, colors, opacities, backgrounds, width, height, tile_size, isect_offsets,
                flatten_ids):
        lib = load_library()
        C, N, D = colors.shape
        th, tw = isect_offsets.shape[1:]
        dev = means2d.device
        n_tiles = th * tw
        n_isects = flatten_ids.numel()
        offs_ext = torch.cat([isect_offsets.reshape(-1).to(torch.int32),
                              torch.tensor([n_isects], dtype=torch.int32, device=dev)]).contiguous()
        render_colors = torch.empty(C, height, width, D, dtype=torch.float32, device=dev)
        render_alphas = torch.empty(C, height, width, 1, dtype=torch.float32, device=dev)
        last_ids = torch.empty(C, height, width, dtype=torch.int32, device=dev)
        st = current_stream()
        for c in range(C):
            check(lib.gsl_rasterize_fwd(
                ptr(means2d), ptr(conics), ptr(colors), ptr(opacities),
                ptr(backgrounds[c]) if backgrounds is not None else None, D, width, height, tile_size, tw, th, 0, th,
                ptr(offs_ext[c * n_tiles:]), ptr(flatten_ids) if n_isects else None, n_isects, ptr(render_colors[c]),
                ptr(render_alphas[c]), ptr(last_ids[c]), st), "gsl_rasterize_fwd")
        ctx.save_for_backward(means2d, conics, colors, opacities,
                              backgrounds if backgrounds is not None else torch.empty(0, device=dev), offs_ext,
                              flatten_ids, render_alphas, last_ids)
        ctx.dims = (width, height, tile_size, tw, th, backgrounds is not None)
        return render_colors, render_alphas

    @staticmethod
    def backward(ctx, v_render_colors, v_render_alphas):
        lib = load_library()
        (means2d, conics, colors, opacities, backgrounds, offs_ext, flatten_ids, render_alphas,
         last_ids) = ctx.saved_tensors
        width, height, tile_size, tw, th, has_bg = ctx.dims
        C, N, D = colors.shape
        n_tiles = th * tw
        n_isects = flatten_ids.numel()
        v_render_colors = v_render_colors.contiguous()
        v_render_alphas = v_render_alphas.contiguous()
        v_means2d = torch.empty_like(means2d)
        v_conics = torch.empty_like(conics)
        v_colors = torch.empty_like(colors)
        v_opacities = torch.empty_like(opacities)
        st = current_stream()
        vacc = torch.zeros(lib.gsl_vacc_bytes(C * N, D) // 4, dtype=torch.float32, device=means2d.device)
        if n_isects:
            for c in range(C):
                check(lib.gsl_rasterize_bwd(
                    ptr(means2d), ptr(conics), ptr(colors), ptr(opacities), ptr(backgrounds[c]) if has_bg else None,
                    D, width, height, tile_size, tw, th, 0, th, ptr(offs_ext[c * n_tiles:]), ptr(flatten_ids),
                    n_isects, ptr(render_alphas[c]), ptr(last_ids[c]), ptr(v_render_colors[c]),
                    ptr(v_render_alphas[c]), ptr(vacc), st), "gsl_rasterize_bwd")
        check(lib.gsl_vacc_unpack(ptr(vacc), C * N, D, ptr(v_means2d), ptr(v_conics), ptr(v_colors),
                                  ptr(v_opacities), st), "gsl_vacc_unpack")
        v_backgrounds = None
        if has_bg and ctx.needs_input_grad[4]:
            v_backgrounds = (v_render_colors * (1.0 - render_alphas)).sum(dim=(1, 2))
        return v_means2d, v_conics, v_colors, v_opacities, v_backgrounds, None, None, None, None, None


def rasterize_to_pixels(
    means2d: Tensor,  # [C, N, 2]
    conics: Tensor,  # [C, N, 3]
    colors: Tensor,  # [C, N, channels]
    opacities: Tensor,  # [C, N]
    image_width: int,
    image_height: int,
    tile_size: int,
    isect_offsets: Tensor,  # [C, tile_height, tile_width]
    flatten_ids: Tensor,  # [n_isects]
    backgrounds: Optional[Tensor] = None,  # [C, channels]
    masks: Optional[Tensor] = None,
    packed: bool = False,
    absgrad: bool = False,
) -> Tuple[Tensor, Tensor]:
    """Rasterize to pixels: render_colors [C,H,W,channels], render_alphas [C,H,W,1]."""
    if packed:
        raise NotImplementedError("packed=True is not supported")
    if absgrad:
        raise NotImplementedError("absgrad is a densification aid GsplatLoc does not use (model.py:124)")
    if masks is not None:
        raise NotImplementedError("tile masks are not supported")
    C, N = means2d.shape[:2]
    assert means2d.shape == (C, N, 2), means2d.shape
    assert conics.shape == (C, N, 3), conics.shape
    assert colors.shape[:2] == (C, N), colors.shape
    assert opacities.shape == (C, N), opacities.shape
    if backgrounds is not None:
        assert backgrounds.shape == (C, colors.shape[-1]), backgrounds.shape
        backgrounds = _dev_f32(backgrounds, "backgrounds")
    assert tile_size == 16, "the gfx950 kernels are built for 16x16 tiles (gsplat's default)"
    tile_height, tile_width = isect_offsets.shape[1:3]
    assert tile_height * tile_size >= image_height, f"Assert Failed: {tile_height} * {tile_size} >= {image_height}"
    assert tile_width * tile_size >= image_width, f"Assert Failed: {tile_width} * {tile_size} >= {image_width}"
    channels = colors.shape[-1]
    padded = _pad_channels(channels)
    if padded != channels:
        colors = torch.cat([colors, torch.zeros(*colors.shape[:-1], padded - channels, device=colors.device,
                                                dtype=colors.dtype)], dim=-1)
        if backgrounds is not None:
            backgrounds = torch.cat([backgrounds, torch.zeros(C, padded - channels, device=colors.device,
                                                              dtype=colors.dtype)], dim=-1)
    render_colors, render_alphas = _RasterizeToPixels.apply(
        _dev_f32(means2d, "means2d"), _dev_f32(conics, "conics"), _dev_f32(colors, "colors"),
        _dev_f32(opacities, "opacities"), backgrounds, int(image_width), int(image_height), int(tile_size),
        isect_offsets.contiguous(), flatten_ids.to(torch.int32).contiguous())
    if padded != channels:
        render_colors = render_colors[..., :channels]
    return render_colors, render_alphas
