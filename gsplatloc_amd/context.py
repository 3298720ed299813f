"""RenderContext -- the fused pipeline with every buffer preallocated.

``rasterization`` (the gsplat-compatible entry) allocates its outputs per call and reads the
intersection count back to size them, exactly like gsplat.  A pose tracker renders the same
Gaussians a few hundred times per frame with a pose that moves by a fraction of a pixel, so
this context fixes all shapes up front: no allocation, no host synchronisation, a constant
launch sequence -- which also makes one iteration capturable in a HIP graph.  The intersection
buffers have a fixed ``capacity`` (measured once with head-room); the device-side count is
checked with ``check_capacity()`` whenever the caller synchronises anyway.

Mirrors what /root/reference/src/my_gsplat/gs_trainer_total.py:97-152 does per iteration
(gs_splats(...) then total_loss.backward()), minus the allocator traffic.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from ._lib import check, current_stream, load_library, ptr
from .fused import _MODES, MAX_STRIP_TILES, _raster_fn, tile_n_bits


class RenderContext:
    def __init__(self, N: int, width: int, height: int, render_mode: str = "RGB+ED", sh_degree: Optional[int] = 1,
                 K_sh: int = 4, device="cuda", eps2d: float = 0.3, near_plane: float = 0.01, far_plane: float = 1e10,
                 radius_clip: float = 0.0, antialiased: bool = False, tile_rows: Optional[Tuple[int, int]] = None,
                 capacity: Optional[int] = None, full_grads: bool = True):
        self.lib = load_library()
        self.N, self.W, self.H = int(N), int(width), int(height)
        self.mode = render_mode
        self.D, self.ed = _MODES[render_mode]
        self.rgb = self.D >= 3
        self.sh_degree = -1 if sh_degree is None else int(sh_degree)
        self.K_sh = int(K_sh) if (self.rgb and self.sh_degree >= 0) else 0
        self.eps2d, self.near, self.far, self.radius_clip = float(eps2d), float(near_plane), float(far_plane), float(radius_clip)
        self.antialiased = bool(antialiased)
        self.tw, self.th = (self.W + 15) // 16, (self.H + 15) // 16
        self.n_tiles = self.tw * self.th
        self.ty0, self.ty1 = tile_rows if tile_rows is not None else (0, self.th)
        assert 0 <= self.ty0 <= self.ty1 <= self.th
        assert (self.ty1 - self.ty0) * self.tw <= MAX_STRIP_TILES, "strip too large for the LDS tile histogram"
        self.full_grads = bool(full_grads)
        dev = torch.device(device)
        self.device = dev
        f32, i32 = torch.float32, torch.int32
        N = self.N
        self.radii = torch.zeros(N, dtype=i32, device=dev)
        self.Q0 = torch.zeros(N, 4, dtype=f32, device=dev)
        self.Q1 = torch.zeros(N, 4, dtype=f32, device=dev)
        self.Q2 = torch.zeros(N, 4, dtype=f32, device=dev) if self.rgb else None
        self.comps = torch.zeros(N, dtype=f32, device=dev) if self.antialiased else None
        self.offs = torch.zeros(self.n_tiles + 1, dtype=i32, device=dev)
        self.n_is = torch.zeros(1, dtype=i32, device=dev)
        self.ws_bytes = self.lib.gsl_fused_ws_bytes(N, self.n_tiles)
        self.ws = torch.zeros(self.ws_bytes, dtype=torch.uint8, device=dev)
        self.render = torch.zeros(self.H, self.W, self.D, dtype=f32, device=dev)
        self.alphas = torch.zeros(self.H, self.W, 1, dtype=f32, device=dev)
        self.last_ids = torch.zeros(self.H, self.W, dtype=i32, device=dev)
        self.vacc = torch.zeros(N, 16, dtype=f32, device=dev)  # cleared by the projection backward
        self.v_viewmat = torch.zeros(4, 4, dtype=f32, device=dev)
        if self.full_grads:
            self.v_means = torch.zeros(N, 3, dtype=f32, device=dev)
            self.v_quats = torch.zeros(N, 4, dtype=f32, device=dev)
            self.v_scales = torch.zeros(N, 3, dtype=f32, device=dev)
            self.v_opacities = torch.zeros(N, dtype=f32, device=dev)
            if self.rgb:
                shape = (N, self.K_sh, 3) if self.sh_degree >= 0 else (N, 3)
                self.v_colors = torch.zeros(*shape, dtype=f32, device=dev)
            else:
                self.v_colors = None
        else:
            self.v_means = self.v_quats = self.v_scales = self.v_opacities = self.v_colors = None
        self.capacity = 0
        self.keys = self.flatten_ids = None
        if capacity is not None:
            self._alloc_isects(int(capacity))
        self._inputs = None

    # ------------------------------------------------------------------ buffers
    def _alloc_isects(self, capacity: int) -> None:
        self.capacity = max(int(capacity), 1)
        self.keys = torch.zeros(self.capacity, dtype=torch.int64, device=self.device)
        self.flatten_ids = torch.zeros(self.capacity, dtype=torch.int32, device=self.device)

    def calibrate(self, means, quats, scales, opacities, colors, viewmat, K, headroom: float = 1.3) -> int:
        """One synchronising projection pass to size the intersection buffers."""
        self._project(means, quats, scales, opacities, colors, viewmat, K)
        n = int(self.n_is.item())
        self._alloc_isects(int(n * headroom) + 1024)
        return n

    def check_capacity(self) -> int:
        """Host sync: intersections of the last forward; raises if they did not fit."""
        n = int(self.n_is.item())
        if n > self.capacity:
            raise RuntimeError(f"intersection capacity exceeded ({n} > {self.capacity}); call calibrate() again")
        return n

    # ------------------------------------------------------------------ forward
    def _project(self, means, quats, scales, opacities, colors, viewmat, K) -> None:
        check(self.lib.gsl_fused_project(
            ptr(means), ptr(quats), ptr(scales), ptr(opacities), ptr(colors) if self.rgb else None, self.sh_degree,
            self.K_sh, ptr(viewmat), ptr(K), self.N, self.W, self.H, self.eps2d, self.near, self.far,
            self.radius_clip, int(self.antialiased), self.tw, self.th, self.ty0, self.ty1, ptr(self.radii),
            ptr(self.Q0), ptr(self.Q1), ptr(self.Q2), ptr(self.comps), None, ptr(self.offs), ptr(self.n_is),
            ptr(self.ws), self.ws_bytes, current_stream()), "gsl_fused_project")

    def forward(self, means: Tensor, quats: Tensor, scales: Tensor, opacities: Tensor, colors: Optional[Tensor],
                viewmat: Tensor, K: Tensor) -> Tuple[Tensor, Tensor]:
        """Render into the context's buffers (valid until the next forward).  Inputs: contiguous fp32
        device tensors; viewmat [4,4], K [3,3].  No allocation, no host sync."""
        assert self.keys is not None, "call calibrate() (or pass capacity=) before forward()"
        st = current_stream()
        self._project(means, quats, scales, opacities, colors, viewmat, K)
        check(self.lib.gsl_fused_bin(ptr(self.Q0), ptr(self.radii), self.N, self.tw, self.th, self.ty0, self.ty1,
                                     tile_n_bits(self.n_tiles), ptr(self.offs), self.capacity, ptr(self.keys),
                                     ptr(self.flatten_ids), None, ptr(self.ws), self.ws_bytes, st), "gsl_fused_bin")
        check(_raster_fn(self.lib, 'fwd')(ptr(self.Q0), ptr(self.Q1), ptr(self.Q2), self.D, int(self.ed), self.W,
                                            self.H, self.tw, self.th, self.ty0, self.ty1, ptr(self.offs),
                                            ptr(self.flatten_ids), self.capacity, ptr(self.render), ptr(self.alphas),
                                            ptr(self.last_ids), st), "gsl_fused_raster_fwd")
        self._inputs = (means, quats, scales, opacities, colors, viewmat, K)
        return self.render, self.alphas

    # ----------------------------------------------------------------- backward
    def backward(self, v_render: Tensor, v_alphas: Tensor, full: Optional[bool] = None) -> Dict[str, Tensor]:
        """vjp of the last forward.  Returns the context's gradient buffers: always ``viewmat``
        ([4,4], row 3 zero); with full gradients also means/quats/scales/opacities/colors."""
        assert self._inputs is not None, "forward() first"
        full = self.full_grads if full is None else full
        assert not full or self.full_grads, "context was built with full_grads=False"
        means, quats, scales, opacities, colors, viewmat, K = self._inputs
        st = current_stream()
        check(_raster_fn(self.lib, 'bwd')(ptr(self.Q0), ptr(self.Q1), ptr(self.Q2), self.D, int(self.ed), self.W,
                                            self.H, self.tw, self.th, self.ty0, self.ty1, ptr(self.offs),
                                            ptr(self.flatten_ids), self.capacity, ptr(self.render), ptr(self.alphas),
                                            ptr(self.last_ids), ptr(v_render), ptr(v_alphas), ptr(self.vacc), st),
              "gsl_fused_raster_bwd")
        check(self.lib.gsl_fused_project_bwd(
            ptr(means), ptr(quats), ptr(scales), ptr(opacities), ptr(colors) if self.rgb else None, self.sh_degree,
            self.K_sh, ptr(viewmat), ptr(K), self.N, self.W, self.H, self.eps2d, int(self.antialiased), self.D,
            ptr(self.radii), ptr(self.Q1), ptr(self.comps), ptr(self.vacc),
            ptr(self.v_means) if full else None, ptr(self.v_quats) if full else None,
            ptr(self.v_scales) if full else None, ptr(self.v_opacities) if full else None,
            ptr(self.v_colors) if (full and self.rgb) else None, ptr(self.v_viewmat), ptr(self.ws), self.ws_bytes,
            self.n_tiles, st), "gsl_fused_project_bwd")
        out = {"viewmat": self.v_viewmat}
        if full:
            out.update(means=self.v_means, quats=self.v_quats, scales=self.v_scales, opacities=self.v_opacities,
                       colors=self.v_colors)
        return out

    # ------------------------------------------------------------ autograd glue
    def render_autograd(self, means, quats, scales, opacities, colors, viewmat, K) -> Tuple[Tensor, Tensor]:
        """forward() as a differentiable op (gradient to viewmat, and to the Gaussians when the
        context carries full gradients and they require grad)."""
        return _CtxRender.apply(self, means, quats, scales, opacities, colors, viewmat, K)


class _CtxRender(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rc: RenderContext, means, quats, scales, opacities, colors, viewmat, K):
        render, alphas = rc.forward(means, quats, scales, opacities, colors, viewmat, K)
        ctx.rc = rc
        ctx.mark_non_differentiable()
        return render, alphas

    @staticmethod
    def backward(ctx, v_render, v_alphas):
        rc = ctx.rc
        ni = ctx.needs_input_grad
        full = rc.full_grads and any(ni[1:6])
        g = rc.backward(v_render.contiguous(), v_alphas.contiguous(), full=full)
        return (None, g["means"] if (full and ni[1]) else None, g["quats"] if (full and ni[2]) else None,
                g["scales"] if (full and ni[3]) else None, g["opacities"] if (full and ni[4]) else None,
                g["colors"] if (full and ni[5] and rc.rgb) else None, g["viewmat"] if ni[6] else None, None)


def time_stages(rc: RenderContext, inputs, v_render: Tensor, v_alphas: Tensor, full: bool, steps: int = 20) -> Dict:
    """Average duration (ms) of each of the five stage calls of one iteration, measured with HIP events
    recorded on the launch stream around every C-ABI call (dev/bench tool; same launches as
    forward()/backward())."""
    means, quats, scales, opacities, colors, viewmat, K = inputs
    lib = rc.lib
    names = ["project_fwd", "bin", "raster_fwd", "raster_bwd", "project_bwd"]
    acc = {n: 0.0 for n in names}

    def ev():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    for it in range(steps + 2):
        st = current_stream()
        marks = [ev()]
        rc._project(means, quats, scales, opacities, colors, viewmat, K)
        marks.append(ev())
        check(lib.gsl_fused_bin(ptr(rc.Q0), ptr(rc.radii), rc.N, rc.tw, rc.th, rc.ty0, rc.ty1,
                                tile_n_bits(rc.n_tiles), ptr(rc.offs), rc.capacity, ptr(rc.keys),
                                ptr(rc.flatten_ids), None, ptr(rc.ws), rc.ws_bytes, st), "gsl_fused_bin")
        marks.append(ev())
        check(_raster_fn(lib, 'fwd')(ptr(rc.Q0), ptr(rc.Q1), ptr(rc.Q2), rc.D, int(rc.ed), rc.W, rc.H, rc.tw,
                                       rc.th, rc.ty0, rc.ty1, ptr(rc.offs), ptr(rc.flatten_ids), rc.capacity,
                                       ptr(rc.render), ptr(rc.alphas), ptr(rc.last_ids), st), "gsl_fused_raster_fwd")
        marks.append(ev())
        check(_raster_fn(lib, 'bwd')(ptr(rc.Q0), ptr(rc.Q1), ptr(rc.Q2), rc.D, int(rc.ed), rc.W, rc.H, rc.tw,
                                       rc.th, rc.ty0, rc.ty1, ptr(rc.offs), ptr(rc.flatten_ids), rc.capacity,
                                       ptr(rc.render), ptr(rc.alphas), ptr(rc.last_ids), ptr(v_render),
                                       ptr(v_alphas), ptr(rc.vacc), st), "gsl_fused_raster_bwd")
        marks.append(ev())
        check(lib.gsl_fused_project_bwd(
            ptr(means), ptr(quats), ptr(scales), ptr(opacities), ptr(colors) if rc.rgb else None, rc.sh_degree,
            rc.K_sh, ptr(viewmat), ptr(K), rc.N, rc.W, rc.H, rc.eps2d, int(rc.antialiased), rc.D, ptr(rc.radii),
            ptr(rc.Q1), ptr(rc.comps), ptr(rc.vacc), ptr(rc.v_means) if full else None,
            ptr(rc.v_quats) if full else None, ptr(rc.v_scales) if full else None,
            ptr(rc.v_opacities) if full else None, ptr(rc.v_colors) if (full and rc.rgb) else None,
            ptr(rc.v_viewmat), ptr(rc.ws), rc.ws_bytes, rc.n_tiles, st), "gsl_fused_project_bwd")
        marks.append(ev())
        torch.cuda.synchronize()
        if it >= 2:
            for i, n in enumerate(names):
                acc[n] += marks[i].elapsed_time(marks[i + 1])
    return {n: acc[n] / steps for n in names}
