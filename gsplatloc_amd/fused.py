"""Fused single-camera render (host side of csrc/fused.hip).

``fused_rasterization`` runs ``gsplat.rasterization``'s whole forward in five
launches and its backward in three, for the configuration GsplatLoc uses
(/root/reference/src/my_gsplat/model.py:195-213: one camera, packed=False, 16x16
tiles, no background, SH or RGB colours, any of the five render modes).
``rendering.rasterization`` dispatches here when the call fits and to the stage
operators otherwise.  ``tile_rows=(ty0, ty1)`` restricts binning and compositing to
a strip of tile rows (screen-tile parallelism, ``gsplatloc_amd.parallel``).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from ._lib import check, current_stream, load_library, ptr

MAX_STRIP_TILES = 8192
_MODES = {"RGB": (3, False), "D": (1, False), "ED": (1, True), "RGB+D": (4, False), "RGB+ED": (4, True)}


def tile_n_bits(n_tiles: int) -> int:
    return int(math.floor(math.log2(n_tiles))) + 1


def alloc_records(lib, N: int, rgb: bool, dev, zero: bool = False):
    """The per-Gaussian record arrays Q0, Q1, Q2 ([N,4] each; Q2 None without colours)."""
    make = torch.zeros if zero else torch.empty
    Q0 = make(N, 4, dtype=torch.float32, device=dev)
    Q1 = make(N, 4, dtype=torch.float32, device=dev)
    Q2 = make(N, 4, dtype=torch.float32, device=dev) if rgb else None
    return Q0, Q1, Q2


class _FusedRasterization(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means, quats, scales, opacities, colors, viewmat, K, cfg, meta):
        lib = load_library()
        (W, H, sh_degree, mode, eps2d, near, far, radius_clip, antialiased, ty0, ty1, want_isect_ids) = cfg
        D, ed = _MODES[mode]
        rgb = D >= 3
        N = means.shape[0]
        dev = means.device
        tw, th = (W + 15) // 16, (H + 15) // 16
        n_tiles = tw * th
        f32, i32 = torch.float32, torch.int32
        radii = torch.empty(N, dtype=i32, device=dev)
        Q0, Q1, Q2 = alloc_records(lib, N, rgb, dev)
        comps = torch.empty(N, dtype=f32, device=dev) if antialiased else None
        tpg = torch.empty(N, dtype=i32, device=dev)
        offs = torch.empty(n_tiles + 1, dtype=i32, device=dev)
        n_is = torch.empty(1, dtype=i32, device=dev)
        ws_bytes = lib.gsl_fused_ws_bytes(N, n_tiles)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        st = current_stream()
        K_sh = colors.shape[1] if (rgb and sh_degree >= 0) else 0
        check(lib.gsl_fused_project(
            ptr(means), ptr(quats), ptr(scales), ptr(opacities), ptr(colors) if rgb else None, sh_degree, K_sh,
            ptr(viewmat), ptr(K), N, W, H, eps2d, near, far, radius_clip, int(antialiased), tw, th, ty0, ty1,
            ptr(radii), ptr(Q0), ptr(Q1), ptr(Q2), ptr(comps), ptr(tpg), ptr(offs), ptr(n_is), ptr(ws), ws_bytes,
            None, None, 0, None, None, st), "gsl_fused_project")  # two-pass binning: tile sizes are not known in advance
        n_isects = int(n_is.item())  # output sizes depend on it (gsplat syncs at the same point)
        cap = max(n_isects, 1)
        keys = torch.empty(cap, dtype=torch.int64, device=dev)
        flatten_ids = torch.empty(n_isects, dtype=i32, device=dev)
        isect_ids = torch.empty(n_isects, dtype=torch.int64, device=dev) if want_isect_ids else None
        check(lib.gsl_fused_bin(ptr(Q0), ptr(radii), N, tw, th, ty0, ty1, tile_n_bits(n_tiles), ptr(offs), n_isects,
                                ptr(keys), ptr(flatten_ids) if n_isects else None,
                                ptr(isect_ids) if (want_isect_ids and n_isects) else None, ptr(ws), ws_bytes, 0, None, 0,
                                None, None, 0, None, None, st), "gsl_fused_bin")
        render = torch.zeros(H, W, D, dtype=f32, device=dev) if (ty0, ty1) != (0, th) else \
            torch.empty(H, W, D, dtype=f32, device=dev)
        alphas = torch.zeros(H, W, 1, dtype=f32, device=dev) if (ty0, ty1) != (0, th) else \
            torch.empty(H, W, 1, dtype=f32, device=dev)
        last_ids = torch.zeros(H, W, dtype=i32, device=dev)
        # per tile and quadrant: the entries it composited (a hit word holds the list index in 28 bits: none beyond that,
        # the backward then tests the blocks geometrically)
        hits = torch.empty(4 * cap, dtype=torch.int32, device=dev) if cap < (1 << 28) else None
        hit_counts = torch.empty(4 * n_tiles + 1, dtype=i32, device=dev) if hits is not None else None
        check(lib.gsl_fused_raster_fwd(ptr(Q0), ptr(Q1), ptr(Q2), D, int(ed), W, H, tw, th, ty0, ty1, ptr(offs),
                                       ptr(flatten_ids) if n_isects else None, n_isects, ptr(render), ptr(alphas),
                                       ptr(last_ids), 0, H, None, None, ptr(hits), ptr(hit_counts), 0, None, 0, None, None,
                                       None, st),
              "gsl_fused_raster_fwd")
        ctx.save_for_backward(means, quats, scales, opacities, colors if rgb else torch.empty(0, device=dev),
                              viewmat, K, radii, Q0, Q1, Q2 if rgb else torch.empty(0, device=dev),
                              comps if antialiased else torch.empty(0, device=dev), offs, flatten_ids, render,
                              alphas, last_ids, ws, hits, hit_counts)
        ctx.cfg = cfg
        ctx.n_isects = n_isects
        ctx.K_sh = K_sh
        if meta is not None:
            meta.update(radii=radii, Q0=Q0, Q1=Q1, Q2=Q2, compensations=comps, tiles_per_gauss=tpg,
                        tile_offsets=offs, flatten_ids=flatten_ids, isect_ids=isect_ids, n_isects=n_isects,
                        last_ids=last_ids)
        ctx.mark_non_differentiable(last_ids)
        return render, alphas, last_ids

    @staticmethod
    def backward(ctx, v_render, v_alphas, _v_last):
        lib = load_library()
        (means, quats, scales, opacities, colors, viewmat, K, radii, Q0, Q1, Q2, comps, offs, flatten_ids, render,
         alphas, last_ids, ws, hits, hit_counts) = ctx.saved_tensors
        (W, H, sh_degree, mode, eps2d, near, far, radius_clip, antialiased, ty0, ty1, _) = ctx.cfg
        D, ed = _MODES[mode]
        rgb = D >= 3
        N = means.shape[0]
        dev = means.device
        tw, th = (W + 15) // 16, (H + 15) // 16
        n_tiles = tw * th
        st = current_stream()
        f32 = torch.float32
        v_render = v_render.contiguous()
        v_alphas = v_alphas.contiguous()
        vacc = torch.zeros(N, 16, dtype=f32, device=dev)
        n_isects = ctx.n_isects
        check(lib.gsl_fused_raster_bwd(ptr(Q0), ptr(Q1), ptr(Q2) if rgb else None, D, int(ed), W, H, tw, th, ty0,
                                       ty1, ptr(offs), ptr(flatten_ids) if n_isects else None, n_isects,
                                       ptr(render), ptr(alphas), ptr(last_ids), ptr(v_render), ptr(v_alphas),
                                       ptr(vacc), 0, H, None, None, ptr(hits), ptr(hit_counts), 0, None, st),
              "gsl_fused_raster_bwd")
        ni = ctx.needs_input_grad
        full = any(ni[:5])
        v_means = v_quats = v_scales = v_opac = v_colors = None
        if full:
            v_means = torch.empty(N, 3, dtype=f32, device=dev)
            v_quats = torch.empty(N, 4, dtype=f32, device=dev)
            v_scales = torch.empty(N, 3, dtype=f32, device=dev)
            v_opac = torch.empty(N, dtype=f32, device=dev)
            if rgb:
                v_colors = torch.empty_like(colors)
        v_viewmat = torch.empty(4, 4, dtype=f32, device=dev) if ni[5] else None
        ws_bytes = ws.numel()
        check(lib.gsl_fused_project_bwd(
            ptr(means), ptr(quats), ptr(scales), ptr(opacities), ptr(colors) if rgb else None, sh_degree, ctx.K_sh,
            ptr(viewmat), ptr(K), N, W, H, eps2d, int(antialiased), D, ptr(radii), ptr(Q1),
            ptr(comps) if antialiased else None, ptr(vacc), ptr(v_means), ptr(v_quats), ptr(v_scales), ptr(v_opac),
            ptr(v_colors), ptr(v_viewmat), ptr(ws), ws_bytes, n_tiles, None, None, None, None, 0, 0, 0, 0, 0, None, None, 1, None,
            st),
            "gsl_fused_project_bwd")
        return (v_means if ni[0] else None, v_quats if ni[1] else None, v_scales if ni[2] else None,
                v_opac if ni[3] else None, v_colors if (ni[4] and rgb) else None, v_viewmat, None, None, None)


# ---------------------------------------------------------------------------------------------------------------------
# Drop-in call with a cached RenderContext.  `from gsplat import rasterization` under the reference's loop
# (/root/reference/src/my_gsplat/gs_trainer_total.py:79-267) renders the same N Gaussians at the same size a few hundred
# times per frame: allocating ~15 tensors per call, binning in two passes and reading the intersection count back to
# size them (what gsplat itself does, and what _FusedRasterization above does) made that call allocator- and
# launch-bound (0.43 ms wall for 0.09 ms of kernels at S).  Here the context -- records, bins,
# lists, capacity -- is kept per call signature; a call is 3 launches into freshly allocated OUTPUT tensors (the caller
# owns what it gets; nothing returned aliases the context), one 32-byte status read (overflow flags and count: the
# lists are complete or the call is repeated after growing the buffers), and the backward is 3 launches.
# GSLOC_DROPIN_CACHE=0 selects the allocate-per-call path.
_CTX_CACHE: Dict[tuple, "object"] = {}
_CTX_CACHE_MAX = 4


def _cached_context(key, make):
    rc = _CTX_CACHE.pop(key, None)
    if rc is None:
        rc = make()
        while len(_CTX_CACHE) >= _CTX_CACHE_MAX:
            _CTX_CACHE.pop(next(iter(_CTX_CACHE)))
    _CTX_CACHE[key] = rc  # most recently used last
    return rc


def clear_context_cache() -> None:
    _CTX_CACHE.clear()


class _CachedRasterization(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means, quats, scales, opacities, colors, viewmat, K, rc, meta):
        dev = means.device
        f32, i32 = torch.float32, torch.int32
        inputs = (means, quats, scales, opacities, colors if rc.rgb else None, viewmat, K)
        if rc.keys is None:
            rc.calibrate(*inputs)
        for attempt in range(3):
            # outputs are fresh tensors the caller owns; the kernels write every pixel of the frame
            rc.render = torch.empty(rc.H, rc.W, rc.D, dtype=f32, device=dev)
            rc.alphas = torch.empty(rc.H, rc.W, 1, dtype=f32, device=dev)
            rc.last_ids = torch.empty(rc.H, rc.W, dtype=i32, device=dev)
            why = rc.forward_checked(*inputs)
            if why is None:
                break
            rc.calibrate(*inputs, headroom=1.5 * (attempt + 1))  # the scene moved past the head-room: re-measure, repeat
        else:
            raise RuntimeError(f"rasterization: buffers kept overflowing ({why})")
        ctx.rc, ctx.gen = rc, rc.generation
        ctx.save_for_backward(means, quats, scales, opacities, colors if rc.rgb else torch.empty(0, device=dev), viewmat, K,
                              rc.render, rc.alphas, rc.last_ids)
        if meta is not None:
            meta.update(radii=rc.radii, Q0=rc.Q0, Q1=rc.Q1, offs=rc.offs, flatten_ids=rc.flatten_ids[:rc.last_n_isects],
                        n_isects=rc.last_n_isects, last_ids=rc.last_ids, tiles_per_gauss=rc.tiles_per_gauss)
        ctx.mark_non_differentiable(rc.last_ids)
        return rc.render, rc.alphas, rc.last_ids

    @staticmethod
    def backward(ctx, v_render, v_alphas, _v_last):
        rc = ctx.rc
        means, quats, scales, opacities, colors, viewmat, K, render, alphas, last_ids = ctx.saved_tensors
        dev = means.device
        f32 = torch.float32
        inputs = (means, quats, scales, opacities, colors if rc.rgb else None, viewmat, K)
        rc.render, rc.alphas, rc.last_ids = render, alphas, last_ids
        if rc.generation != ctx.gen:
            # another forward has used this context since: its records and lists are not this node's any more --
            # run this node's forward again (same inputs: same outputs, rewritten into the saved tensors).  The buffers
            # may have been re-sized for the later call's scene: check, and refuse to build a gradient from truncated lists
            why = rc.forward_checked(*inputs)
            if why is not None:
                raise RuntimeError(f"backward of a stale rasterization node: {why} (a later call with the same signature "
                                   "re-sized the cached context; set GSLOC_DROPIN_CACHE=0 to give every call its own buffers)")
        rc._inputs = inputs
        ni = ctx.needs_input_grad
        full = any(ni[:5])
        N = rc.N
        if full:  # gradient tensors are fresh too
            rc.v_means = torch.empty(N, 3, dtype=f32, device=dev)
            rc.v_quats = torch.empty(N, 4, dtype=f32, device=dev)
            rc.v_scales = torch.empty(N, 3, dtype=f32, device=dev)
            rc.v_opacities = torch.empty(N, dtype=f32, device=dev)
            rc.v_colors = torch.empty_like(colors) if rc.rgb else None
        rc.v_viewmat = torch.empty(4, 4, dtype=f32, device=dev)
        g = rc.backward(v_render.contiguous(), v_alphas.contiguous(), full=full)
        return (g["means"] if ni[0] else None, g["quats"] if ni[1] else None, g["scales"] if ni[2] else None,
                g["opacities"] if ni[3] else None, g["colors"] if (ni[4] and rc.rgb) else None,
                g["viewmat"] if ni[5] else None, None, None, None)


def cached_rasterization(means, quats, scales, opacities, colors, viewmat, K, width, height, sh_degree=None,
                         render_mode="RGB", eps2d=0.3, near_plane=0.01, far_plane=1e10, radius_clip=0.0,
                         antialiased=False) -> Tuple[Tensor, Tensor, Dict]:
    """fused_rasterization through a cached RenderContext (whole frame, one camera).  Same returns; the meta tensors
    (radii, means2d, depths, conics, opacities, tile lists) are VIEWS of the context's buffers: valid until the next
    call with the same signature (GsplatLoc never reads them); isect_ids is not produced."""
    from .context import RenderContext

    def prep(t, name):
        assert t.is_cuda, f"{name} must live on the GPU (got {t.device}); there is no CPU path"
        assert t.dtype == torch.float32, f"{name} must be float32 (got {t.dtype})"
        return t.contiguous()

    N = means.shape[0]
    D, _ = _MODES[render_mode]
    rgb = D >= 3
    deg = -1 if sh_degree is None else int(sh_degree)
    K_sh = colors.shape[1] if (rgb and deg >= 0) else 0
    key = (N, int(width), int(height), render_mode, deg, K_sh, float(eps2d), float(near_plane), float(far_plane),
           float(radius_clip), bool(antialiased), means.device.index)
    rc = _cached_context(key, lambda: RenderContext(
        N, width, height, render_mode, sh_degree=sh_degree, K_sh=K_sh, device=means.device, eps2d=eps2d,
        near_plane=near_plane, far_plane=far_plane, radius_clip=radius_clip, antialiased=antialiased, full_grads=False,
        reorder=False))  # (the drop-in call returns per-Gaussian meta tensors and lists in the caller's order)
    tensors = (means, quats, scales, opacities, colors, viewmat)
    # hit lists only when somebody can back-propagate through this call (geometry.py:117-132 renders under no_grad)
    rc.record_hits = torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in tensors)
    rc.allow_tiny = False  # a splat that outgrew the tiny backward is only known after the backward: not in this API
    rc.full_grads = True  # gradient buffers are allocated per call (owned by the caller), not by the context
    if rc.tiles_per_gauss is None:
        rc.tiles_per_gauss = torch.zeros(N, dtype=torch.int32, device=means.device)
    raw: Dict = {}
    render, alphas, _ = _CachedRasterization.apply(
        prep(means, "means"), prep(quats, "quats"), prep(scales, "scales"), prep(opacities, "opacities"),
        prep(colors, "colors") if rgb else colors, prep(viewmat, "viewmats"), prep(K, "Ks"), rc, raw)
    tw, th = rc.tw, rc.th
    Q0, Q1 = raw["Q0"], raw["Q1"]
    meta = {
        "camera_ids": None, "gaussian_ids": None,
        "radii": raw["radii"][None], "means2d": Q0[None, :, 0:2], "depths": Q0[None, :, 2],
        "conics": Q1[None, :, 0:3], "opacities": Q0[None, :, 3],
        "tile_width": tw, "tile_height": th, "tiles_per_gauss": raw["tiles_per_gauss"][None],
        "isect_ids": None, "flatten_ids": raw["flatten_ids"],
        "isect_offsets": raw["offs"][:-1].reshape(1, th, tw), "width": width, "height": height,
        "tile_size": 16, "n_cameras": 1,
    }
    return render, alphas, meta


def fused_supported(N: int, C: int, colors: Tensor, sh_degree: Optional[int], width: int, height: int,
                    tile_size: int, backgrounds, render_mode: str, tile_rows=None) -> bool:
    if C != 1 or tile_size != 16 or backgrounds is not None:
        return False
    tw, th = (width + 15) // 16, (height + 15) // 16
    ty0, ty1 = tile_rows if tile_rows is not None else (0, th)
    if (ty1 - ty0) * tw > MAX_STRIP_TILES:
        return False
    if _MODES[render_mode][0] >= 3:
        if sh_degree is None:
            return colors.dim() == 2 and colors.shape == (N, 3)
        return colors.dim() == 3 and colors.shape[0] == N and colors.shape[2] == 3 and 0 <= sh_degree <= 3
    return True


def fused_rasterization(
    means: Tensor, quats: Tensor, scales: Tensor, opacities: Tensor, colors: Tensor, viewmat: Tensor, K: Tensor,
    width: int, height: int, sh_degree: Optional[int] = None, render_mode: str = "RGB", eps2d: float = 0.3,
    near_plane: float = 0.01, far_plane: float = 1e10, radius_clip: float = 0.0, antialiased: bool = False,
    tile_rows: Optional[Tuple[int, int]] = None, want_isect_ids: bool = True,
) -> Tuple[Tensor, Tensor, Dict]:
    """One camera.  Returns render [H,W,X], alphas [H,W,1], meta (gsplat keys, [1,N,...] views)."""
    def prep(t, name):
        assert t.is_cuda, f"{name} must live on the GPU (got {t.device}); there is no CPU path"
        assert t.dtype == torch.float32, f"{name} must be float32 (got {t.dtype})"
        return t.contiguous()

    tw, th = (width + 15) // 16, (height + 15) // 16
    ty0, ty1 = tile_rows if tile_rows is not None else (0, th)
    assert 0 <= ty0 <= ty1 <= th, (ty0, ty1, th)
    cfg = (int(width), int(height), -1 if sh_degree is None else int(sh_degree), render_mode, float(eps2d),
           float(near_plane), float(far_plane), float(radius_clip), bool(antialiased), int(ty0), int(ty1),
           bool(want_isect_ids))
    raw: Dict = {}
    render, alphas, _ = _FusedRasterization.apply(
        prep(means, "means"), prep(quats, "quats"), prep(scales, "scales"), prep(opacities, "opacities"),
        prep(colors, "colors"), prep(viewmat, "viewmats"), prep(K, "Ks"), cfg, raw)
    Q0, Q1 = raw["Q0"], raw["Q1"]
    meta = {
        "camera_ids": None, "gaussian_ids": None,
        "radii": raw["radii"][None], "means2d": Q0[None, :, 0:2], "depths": Q0[None, :, 2],
        "conics": Q1[None, :, 0:3], "opacities": Q0[None, :, 3],
        "tile_width": tw, "tile_height": th, "tiles_per_gauss": raw["tiles_per_gauss"][None],
        "isect_ids": raw["isect_ids"], "flatten_ids": raw["flatten_ids"],
        "isect_offsets": raw["tile_offsets"][:-1].reshape(1, th, tw), "width": width, "height": height,
        "tile_size": 16, "n_cameras": 1,
    }
    return render, alphas, meta
