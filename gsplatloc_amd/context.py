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

import math
import os
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from ._lib import check, current_stream, load_library, ptr
from .fused import _MODES, MAX_STRIP_TILES, alloc_records, tile_n_bits


LONG_MIN = 2048   # a tile list longer than this is split over workgroups (segments of gsl_long_segment() entries)
BIN_BYTES_MAX = 2 << 30  # per-tile key bins larger than this in total: stay with two-pass binning
TINY_RCULL_MAX = 1.999  # r_cull below this: the alpha >= 1/255 disc spans at most 4 pixel centres per axis
# Above this many Gaussians the general backward is the faster one even on a depth frame (end of round 4, whole tracker
# iterations per second, tiny / general: 102 k 12 930 / 12 360, 307 k 7 020 / 7 090, 816 k 3 390 / 3 510; the step of
# workload D 0.283 / 0.270 ms): the slabs are 256 bytes of traffic per Gaussian, the launch they save a few microseconds.
# Render modes with colour only: with one composited channel ("ED", what gsplatloc_amd.eval renders) the slabs still win
# at 816 k (3 990 against 3 710 it/s).
TINY_MAX_N = 400_000


class RenderContext:
    def __init__(self, N: int, width: int, height: int, render_mode: str = "RGB+ED", sh_degree: Optional[int] = 1,
                 K_sh: int = 4, device="cuda", eps2d: float = 0.3, near_plane: float = 0.01, far_plane: float = 1e10,
                 radius_clip: float = 0.0, antialiased: bool = False, tile_rows: Optional[Tuple[int, int]] = None,
                 capacity: Optional[int] = None, full_grads: bool = True,
                 pixel_rows: Optional[Tuple[int, int]] = None, staging: str = "fp32", deterministic: bool = False,
                 sort_in_forward: bool = False, reorder: Optional[bool] = None):
        self.lib = load_library()
        # reorder (tile-order placement, see calibrate()): True / False, or None = decide at calibration -- on when the
        # Gaussians are numerous and NOT already in a screen-coherent order (a back-projected depth frame is)
        self.reorder = reorder
        self.order_ids = self.storage_of = None  # int32[N]: original index of a storage slot / slot of an original index
        self._placed = None       # context-owned copies of the per-Gaussian inputs in storage order
        self._placed_key = None   # (data_ptr, version) of the caller's tensors they were made from
        # sort_in_forward (callers whose every forward is followed by a backward: the tracker): the compositing forward
        # sorts its own tile's bin -- no gsl_fused_bin launch -- whenever the frame allows it (sorts_in_forward())
        self.sort_in_forward = bool(sort_in_forward)
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
        # pixel rows actually composited inside the tile rows (a strip plus its one-pixel Sobel halo); default: all
        lo, hi = self.ty0 * 16, min(self.ty1 * 16, self.H)
        self.row0, self.row1 = pixel_rows if pixel_rows is not None else (lo, hi)
        assert lo <= self.row0 <= self.row1 <= hi or self.ty0 == self.ty1, (pixel_rows, tile_rows)
        self.full_grads = bool(full_grads)
        dev = torch.device(device)
        self.device = dev
        f32, i32 = torch.float32, torch.int32
        N = self.N
        self.radii = torch.zeros(N, dtype=i32, device=dev)
        self.Q0, self.Q1, self.Q2 = alloc_records(self.lib, N, self.rgb, dev, zero=True)
        # "fp16": the compositing kernels gather one 32-byte half-precision record per splat (BASELINE.json configs[4]);
        # transmittance and all accumulators stay float32
        assert staging in ("fp32", "fp16"), staging
        self.staging = staging
        self.Qh = torch.zeros(N, 8, dtype=i32, device=dev) if staging == "fp16" else None
        # deterministic=True: no float atomics anywhere in the backward (one gradient row per intersection, summed in a
        # fixed order): bit-identical gradients from run to run, at the price of a slower backward
        self.deterministic = bool(deterministic)
        self.vrow = None
        self.comps = torch.zeros(N, dtype=f32, device=dev) if self.antialiased else None
        self.offs = torch.zeros(self.n_tiles + 1, dtype=i32, device=dev)
        # [flags 0..3 | n_isects]: one buffer, so that a caller that must look at them reads them with ONE copy.
        # flags (device-set, sticky): [0] a splat outgrew the tiny backward, [1] a tile outgrew its bin, [2] that tile's
        # size, [3] a binned projection started on uncleared tile counters
        self.status = torch.zeros(8, dtype=i32, device=dev)
        self.n_is = self.status[4:5]
        self.hit_counts = torch.zeros(4 * self.n_tiles + 1, dtype=i32, device=dev)  # lengths of the quadrants' hit lists
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
                self.vc_state = torch.ones(1, dtype=i32, device=dev)  # "v_colors holds zeros only" (gsl_fused_project_bwd)
            else:
                self.v_colors = None
        else:
            self.v_means = self.v_quats = self.v_scales = self.v_opacities = self.v_colors = None
        if not hasattr(self, "vc_state"):
            self.vc_state = None
        self.capacity = 0
        self.tiny = False
        self.flags = self.status[0:4]
        self.last_n_isects = 0
        self.tiles_per_gauss = None  # optional [N] int32 output of the projection (gsplat's meta key)
        self.record_hits = True  # False: a forward nobody will back-propagate skips the hit lists (and its backward, if
        # one is asked for after all, tests the blocks geometrically: same result)
        self.generation = 0  # counts forwards: an autograd node checks that "its" forward is still the last one
        self.bins, self.bin_cap = None, 0
        self._counters_dirty = False
        self._hits_valid = False
        self.trec = self.vcT = None
        # long tile lists (a pile of splats in one tile): split over workgroups when calibrate() finds one
        self.long_min, self.max_seg, self.long_ws, self.long_ws_bytes, self.long_passes = 0, 0, None, 0, 0
        self._mean_list = 0.0
        self.keys = self.flatten_ids = self.hits = None
        if capacity is not None:
            self._alloc_isects(int(capacity))
        self._inputs = None

    # ------------------------------------------------------------------ buffers
    def _alloc_isects(self, capacity: int) -> None:
        self.capacity = max(int(capacity), 1)
        self.keys = torch.zeros(self.capacity, dtype=torch.int64, device=self.device)
        self.flatten_ids = torch.zeros(self.capacity, dtype=torch.int32, device=self.device)
        # per tile and quadrant: the list entries its pixels composited (written by the forward, read by the backward).
        # A hit word keeps the list index in 28 bits: beyond 2^28 intersections the forward records nothing and the backward
        # tests the blocks geometrically (same result; the C side refuses hit lists at that size)
        self.hits = (torch.zeros(4 * self.capacity, dtype=torch.int32, device=self.device)
                     if self.capacity < (1 << 28) else None)
        if self.deterministic:
            self.vrow = torch.zeros(self.capacity, 16, dtype=torch.float32, device=self.device)

    def calibrate(self, means, quats, scales, opacities, colors, viewmat, K, headroom: float = 1.3) -> int:
        """One synchronising projection pass to size the intersection buffers and the per-tile bins -- and, for Gaussians
        in no screen-coherent order, to choose their tile-order placement (_choose_placement)."""
        self.bins, self.bin_cap = None, 0  # two-pass binning for this measuring pass
        self.order_ids = self.storage_of = self._placed = self._placed_key = None
        self._project(means, quats, scales, opacities, colors, viewmat, K)
        if self._choose_placement():
            # records in storage order (what _choose_backward looks at); same lists, same sizes
            self._project(*self._place(means, quats, scales, opacities, colors), viewmat, K)
        n = int(self.n_is.item())
        self._alloc_isects(int(n * headroom) + 1024)
        sizes = (self.offs[1:] - self.offs[:-1]) if self.n_tiles else self.offs[:0]
        longest = int(sizes.max()) if self.n_tiles else 0
        self._alloc_bins(int(longest * max(headroom, 1.5)) + 64)
        self._choose_backward()
        self._alloc_long(sizes, headroom)
        return n

    # ------------------------------------------------------------------ tile-order placement
    REORDER_MIN_N = 65536  # below this the whole record set sits in one XCD's L2 anyway
    # above this the sorts' relabelling gather (4 bytes per list entry from a table of 4 N bytes, in random order) leaves
    # the 4 MiB L2: measured at 5 M Gaussians / 1920x1080 the sort goes 163 -> 363 us, more than projection and compositing
    # gain (step 1.53 -> 1.64 ms); at 2 M it is a draw (0.768 -> 0.759 ms); at 1 M a win (0.579 -> 0.543 ms).  Carrying the
    # slot through the sort as a payload instead of gathering it afterwards is the way past this -- not built.
    REORDER_MAX_N = 1_500_000

    def _choose_placement(self) -> bool:
        """Tile-order placement of the Gaussians, once per frame (the inputs are static while the pose is optimised:
        /root/reference/src/my_gsplat/model.py:137-175).  With Gaussians in random order every list entry's record gather
        pulls its own 64-128-byte line (3.5-7 x the algorithmic bytes of the compositing forward) and
        the projection scatters its bin writes over the whole image.  Here the context keeps its OWN copies of the inputs
        sorted (stably) by the tile of the projected centre at the calibration pose; `order_ids[slot]` is the original
        index and stays the low word of the sort key, so depth ties break exactly as in the caller's order -- the lists
        are the unpermuted run's lists with every id relabelled (`storage_of`), images and last_ids bit-identical
        (tests/test_gpu_reorder.py).  Everything per Gaussian the context owns (records, radii, tiles_per_gauss, gradient
        buffers) is then in STORAGE order: use grads_in_input_order() / order_ids."""
        want = self.reorder
        if want is None:
            want = (os.environ.get("GSLOC_REORDER", "1") != "0" and self.REORDER_MIN_N <= self.N <= self.REORDER_MAX_N
                    and not self.deterministic and not self._screen_coherent_order())
        if not want or self.N < 2:
            return False
        assert not self.deterministic, "deterministic mode finds a Gaussian's rows by key: no tile-order placement"
        vis = self.Q1[:, 3] > 0
        tx = torch.clamp(torch.floor(self.Q0[:, 0] / 16.0), 0, self.tw - 1).to(torch.int64)
        ty = torch.clamp(torch.floor(self.Q0[:, 1] / 16.0), 0, self.th - 1).to(torch.int64)
        tile = torch.where(vis, ty * self.tw + tx, torch.full_like(tx, self.n_tiles))  # culled Gaussians last
        perm = torch.sort(tile, stable=True).indices
        self.order_ids = perm.to(torch.int32).contiguous()
        self.storage_of = torch.empty_like(self.order_ids)
        self.storage_of[perm] = torch.arange(self.N, dtype=torch.int32, device=self.device)
        self._perm64 = perm.contiguous()
        return True

    def _place(self, means, quats, scales, opacities, colors):
        """The caller's per-Gaussian inputs -> the context's copies in storage order (re-gathered only when the caller
        passes other tensors, or has written to them in place, since the last call)."""
        if self.order_ids is None:
            return means, quats, scales, opacities, colors
        srcs = (means, quats, scales, opacities) + ((colors,) if (self.rgb and colors is not None) else ())
        key = tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in srcs)
        if self._placed_key != key:
            if self._placed is None or any(p.shape != t.shape for p, t in zip(self._placed, srcs)):
                self._placed = tuple(torch.empty_like(t) for t in srcs)
            for p, t in zip(self._placed, srcs):
                torch.index_select(t.detach(), 0, self._perm64, out=p)
            self._placed_key = key
        pl = self._placed
        return pl[0], pl[1], pl[2], pl[3], (pl[4] if len(pl) > 4 else colors)

    def gaussians_reaching(self, rows: Tuple[int, int], guard_tiles: int = 1) -> Tensor:
        """Indices -- in the CALLER's order, whatever the placement -- of the Gaussians whose splat (centre +- radius of the
        last projection) touches tile rows [rows[0] - guard, rows[1] + guard): what a strip keeps (parallel.py).  Use this,
        not gaussians_for_strip(rc.Q0, rc.radii, ...) on a context that may have placed its Gaussians: Q0 and radii are in
        STORAGE order then, and indexing the caller's arrays with storage slots silently selects other Gaussians (round 4:
        a strip-scaling estimate that rendered half of every strip's intersections and looked super-linear)."""
        from .parallel import gaussians_for_strip
        idx = gaussians_for_strip(self.Q0[:, 0:2], self.radii, rows, guard_tiles=guard_tiles)
        return idx if self.order_ids is None else self.order_ids.long()[idx]

    def grads_in_input_order(self, grads: Dict[str, Tensor]) -> Dict[str, Tensor]:
        """backward()'s per-Gaussian gradients in the CALLER's order (new tensors; `viewmat` passed through)."""
        if self.storage_of is None:
            return grads
        idx = self.storage_of.to(torch.int64)
        return {k: (v if (k == "viewmat" or v is None) else torch.index_select(v, 0, idx)) for k, v in grads.items()}

    def _alloc_long(self, sizes: Tensor, headroom: float) -> None:
        """Long-list mode: on when some tile list comes near LONG_MIN entries (deterministic mode keeps its own
        backward).  The workspace holds headroom x the segments of every tile that is at least half that long."""
        self.long_min, self.max_seg, self.long_ws, self.long_ws_bytes, self.long_passes = 0, 0, None, 0, 0
        if self.deterministic or os.environ.get("GSLOC_LONG_LISTS", "1") == "0" or not sizes.numel():
            return
        # "long" is relative to the frame: four times its mean (non-empty) list, at least LONG_MIN entries -- workload X
        # has ordinary lists of ~1 500 entries, and launching the long-list kernels over empty segment grids would cost it
        # 4 % for nothing
        nonempty = sizes[sizes > 0]
        self._mean_list = float(nonempty.double().mean()) if nonempty.numel() else 0.0
        long_min = max(LONG_MIN, int(4.0 * self._mean_list))
        near = sizes[sizes > long_min // 2]
        if not near.numel() or int(near.max()) <= int(long_min * 0.75):
            return
        # every near-long tile's segments with head-room, plus three more copies of the longest list: a pile that sits
        # on a tile corner appears in four tile lists at once, and it moves by tens of pixels per iteration
        hr = max(headroom, 1.5)
        seg = float(self.lib.gsl_long_segment())
        segs = int(torch.ceil(near.double() * hr / seg).sum()) + 3 * int(math.ceil(float(near.max()) * hr / seg)) + 8
        self.long_min, self.max_seg = long_min, segs
        # merge passes of the long-list sort: runs of one sort segment doubled until they cover 1.5 x the longest list
        self.long_passes = max(1, math.ceil(math.log2(max(2.0, 1.5 * float(near.max()) / self.lib.gsl_long_sort_segment()))))
        self.long_ws_bytes = self.lib.gsl_long_ws_bytes(segs)
        self.long_ws = torch.zeros(self.long_ws_bytes, dtype=torch.uint8, device=self.device)

    def grow_long(self, needed: int) -> None:
        """Recovery after long_overflowed(): a workspace for 1.5 x the segments the frame needed."""
        self.long_min, self.max_seg = max(self.long_min, LONG_MIN), max(self.max_seg, int(needed * 1.5) + 8)
        runs = needed * self.lib.gsl_long_segment() / self.lib.gsl_long_sort_segment()
        self.long_passes = max(self.long_passes, math.ceil(math.log2(max(2.0, 1.5 * runs))))
        self.long_ws_bytes = self.lib.gsl_long_ws_bytes(self.max_seg)
        self.long_ws = torch.zeros(self.long_ws_bytes, dtype=torch.uint8, device=self.device)

    def long_overflowed(self) -> int:
        """Host sync: 0, or the number of (tile, segment) pairs a frame needed beyond the long-list workspace."""
        if self.long_ws is None:
            return 0
        st = self.long_ws[:16].view(torch.int32).tolist()
        return int(st[1]) if st[1] else (int(st[2]) // self.lib.gsl_long_segment() + 1 if st[2] else 0)

    def _alloc_bins(self, bin_cap: int) -> None:
        """Fixed-capacity per-tile key bins: the projection kernel then bins directly (no scatter pass, no counter
        clearing launch).  Skipped -- two-pass binning stays -- when one tile is so long that bins for every tile
        would not be worth their memory (BIN_BYTES_MAX)."""
        self.flags[1:3] = 0
        if self.n_tiles * bin_cap * 8 > BIN_BYTES_MAX or os.environ.get("GSLOC_BINNING", "direct") == "two-pass":
            self.bins, self.bin_cap = None, 0
            return
        self.bin_cap = int(bin_cap)
        self.bins = torch.zeros(self.n_tiles * self.bin_cap, dtype=torch.int64, device=self.device)
        self.ws.zero_()  # the binned projection relies on cleared tile counters (it leaves them cleared)
        self.flags[3] = 0
        self._counters_dirty = False

    def counters_were_dirty(self) -> bool:
        """Host sync: did a binned projection start on tile counters a skipped / failed forward had left uncleared?
        (flags[3], raised by the projection kernel; the lists of that iteration are then wrong.)"""
        return bool(self.bins is not None and int(self.flags[3].item()))

    def bins_overflowed(self) -> int:
        """Host sync: 0, or the length of the longest tile list that did not fit its bin since the last calibration."""
        if self.bins is None:
            return 0
        f = self.flags.tolist()
        return int(f[2]) if f[1] else 0

    def grow_bins(self, longest: int) -> None:
        """Recovery after bins_overflowed(): bins for 1.5 x the longest list seen -- and, if that list is long by the
        frame's standards (a pile of invalid-depth points that came into view DURING the optimisation: it sits at the
        previous camera's origin and passes the near plane only once the camera has moved back far enough), the
        long-list split, which calibrate() could not have switched on."""
        self._alloc_bins(int(longest * 1.5) + 64)
        if self.long_min == 0 and not self.deterministic and os.environ.get("GSLOC_LONG_LISTS", "1") != "0":
            long_min = max(LONG_MIN, int(4.0 * self._mean_list))
            if longest > int(long_min * 0.75):
                seg = float(self.lib.gsl_long_segment())
                segs = 4 * int(math.ceil(longest * 1.5 / seg)) + 8
                self.long_min, self.max_seg = long_min, segs
                self.long_passes = max(1, math.ceil(math.log2(max(2.0, 1.5 * longest / self.lib.gsl_long_sort_segment()))))
                self.long_ws_bytes = self.lib.gsl_long_ws_bytes(segs)
                self.long_ws = torch.zeros(self.long_ws_bytes, dtype=torch.uint8, device=self.device)

    def _choose_backward(self) -> None:
        """Tiny-splat backward (per-splat 4x4 record slabs, no reduction, no atomics) when no splat reaches
        more than 4x4 pixel centres (r_cull < 2 px) -- GsplatLoc's as-coded scales; otherwise the general
        compositing backward.  Only for at most TINY_MAX_N Gaussians (any number in a one-channel mode) in a screen-coherent order
        (_screen_coherent_order)."""
        r_max = float(self.Q1[:, 3].max()) if self.N else 0.0
        mode = os.environ.get("GSLOC_BWD", "auto")  # dev switch: "general" / "tiny" (whatever the order) / "auto"
        want = (mode != "general" and r_max < TINY_RCULL_MAX and self.Qh is None and getattr(self, "allow_tiny", True)
                and (mode == "tiny" or ((self.N <= TINY_MAX_N or self.D == 1) and self._screen_coherent_order())))
        if want and self.trec is None:
            self.trec = torch.zeros(self.N, 32, dtype=torch.float32, device=self.device)
            self.vcT = torch.zeros(self.H, self.W, self.D, dtype=torch.float32, device=self.device)
        self.tiny = want
        self.flags[0] = 0

    def sorts_in_forward(self) -> bool:
        """The compositing forward does gsl_fused_bin's work for its own tile (gsl_fused_raster_fwd(..., sort_bins)):
        asked for (sort_in_forward), binned projection, whole frame, every bin at most 1024 keys, no long lists, not the
        deterministic mode.  The tile counters are then cleared by the compositing backward."""
        mode = os.environ.get("GSLOC_SORT_IN_FORWARD", "1")  # "0": never; "force": also for bins of 1025..2048 keys
        # (bins above 1024 keys need 32 KB of merge buffers per workgroup: the forward then loses more occupancy than the
        # launch saves -- 1 M random splats, lists of ~740: 0.569 -> 0.597 ms per step -- so those keep the sort launch)
        return bool(self.sort_in_forward and mode != "0" and self.bins is not None
                    and 0 < self.bin_cap <= (2048 if mode == "force" else 1024) and self.long_min == 0
                    and not self.deterministic and self.ty0 == 0 and self.ty1 == self.th)

    def _screen_coherent_order(self) -> bool:
        """Do consecutive Gaussians land in the same or a neighbouring tile (a back-projected depth frame in pixel
        order: the reference's only input, /root/reference/src/my_gsplat/geometry.py:138-161)?  The tiny-splat backward
        keeps one 128-byte slab per Gaussian, written by the compositing kernel in tile order and folded by the
        projection backward in Gaussian order: with a random order both ends are scattered and the general backward is
        the faster one (1 M random sub-pixel splats: 0.41 against 0.50 ms per step; in pixel order the slabs win by 3 %
        at the tracker's sizes)."""
        vis = self.Q1[:, 3] > 0
        if int(vis.sum()) < 2:
            return True
        t = torch.floor(self.Q0[:, 0:2][vis] / 16.0)
        step = (t[1:] - t[:-1]).abs().sum(dim=1)
        return float((step <= 1).float().mean()) > 0.5

    def viewmat_rows(self) -> Tuple[int, int]:
        """(device pointer, row count) of the pose-gradient partial rows a backward(reduce_viewmat=False) leaves."""
        return self.lib.gsl_fused_viewmat_rows(ptr(self.ws), self.n_tiles), (self.N + 255) // 256

    def tiny_overflowed(self) -> bool:
        """Host sync: did a splat outgrow the tiny backward since the last calibration?  (The kernel raises the
        sticky device flag instead of dropping the gradient silently.)"""
        return bool(self.tiny and int(self.flags[0].item()))

    def use_general_backward(self) -> None:
        """Recovery after tiny_overflowed(): switch this context to the general compositing backward."""
        self.tiny = False
        self.flags[0] = 0

    def check_capacity(self) -> int:
        """Host sync: intersections of the last forward; raises if they did not fit (or if a splat outgrew the
        tiny-splat backward chosen at calibration)."""
        n = int(self.n_is.item())
        if n > self.capacity:
            raise RuntimeError(f"intersection capacity exceeded ({n} > {self.capacity}); call calibrate() again")
        if self.tiny_overflowed():
            raise RuntimeError("a splat outgrew the tiny-splat backward (r_cull >= 2 px); call calibrate() again")
        if self.bins_overflowed():
            raise RuntimeError(f"a tile list outgrew its bin ({self.bins_overflowed()} > {self.bin_cap}); call calibrate() again")
        if self.long_overflowed():
            raise RuntimeError(f"long tile lists need {self.long_overflowed()} segments (> {self.max_seg}); call calibrate() again")
        if self.counters_were_dirty():
            raise RuntimeError("a binned projection found the tile counters uncleared (a forward was skipped between two "
                               "projections); call calibrate() again")
        return n

    # ------------------------------------------------------------------ stages (one C-ABI call each)
    def _project(self, means, quats, scales, opacities, colors, viewmat, K) -> None:
        if self.bins is not None:
            if self._counters_dirty:  # the previous projection was never followed by a compositing forward
                self.ws.zero_()
            self._counters_dirty = True
        check(self.lib.gsl_fused_project(
            ptr(means), ptr(quats), ptr(scales), ptr(opacities), ptr(colors) if self.rgb else None, self.sh_degree,
            self.K_sh, ptr(viewmat), ptr(K), self.N, self.W, self.H, self.eps2d, self.near, self.far,
            self.radius_clip, int(self.antialiased), self.tw, self.th, self.ty0, self.ty1, ptr(self.radii),
            ptr(self.Q0), ptr(self.Q1), ptr(self.Q2), ptr(self.comps), ptr(self.tiles_per_gauss), ptr(self.offs),
            ptr(self.n_is),
            ptr(self.ws), self.ws_bytes, ptr(self.Qh), ptr(self.bins), self.bin_cap, ptr(self.flags),
            ptr(self.order_ids), current_stream()), "gsl_fused_project")

    def _bin(self) -> None:
        if self.sorts_in_forward():
            return  # (the compositing forward sorts its own tile's bin)
        check(self.lib.gsl_fused_bin(ptr(self.Q0), ptr(self.radii), self.N, self.tw, self.th, self.ty0, self.ty1,
                                     tile_n_bits(self.n_tiles), ptr(self.offs), self.capacity, ptr(self.keys),
                                     ptr(self.flatten_ids), None, ptr(self.ws), self.ws_bytes, int(self.deterministic),
                                     ptr(self.bins), self.bin_cap, ptr(self.n_is), ptr(self.flags),
                                     self.long_min if self.bins is not None else 0, ptr(self.order_ids),
                                     ptr(self.storage_of), current_stream()),
              "gsl_fused_bin")
        if self.long_min and self.bins is not None:  # the long lists: sorted by several workgroups
            check(self.lib.gsl_long_sort(ptr(self.offs), self.tw, self.th, self.ty0, self.ty1, self.capacity,
                                         ptr(self.bins), self.bin_cap, ptr(self.keys), ptr(self.flatten_ids),
                                         self.long_min, ptr(self.long_ws), self.long_ws_bytes, self.max_seg,
                                         self.long_passes, ptr(self.storage_of), current_stream()), "gsl_long_sort")

    def _raster_fwd(self) -> None:
        sif = self.sorts_in_forward()
        check(self.lib.gsl_fused_raster_fwd(ptr(self.Q0), ptr(self.Q1), ptr(self.Q2), self.D, int(self.ed), self.W,
                                            self.H, self.tw, self.th, self.ty0, self.ty1, ptr(self.offs),
                                            ptr(self.flatten_ids), self.capacity, ptr(self.render), ptr(self.alphas),
                                            ptr(self.last_ids), self.row0, self.row1, ptr(self.Qh),
                                            ptr(self.ws) if self.bins is not None else None,
                                            ptr(self.hits) if (self.record_hits and self.hits is not None) else None,
                                            ptr(self.hit_counts) if (self.record_hits and self.hits is not None) else None,
                                            self.long_min, ptr(self.bins) if sif else None, self.bin_cap if sif else 0,
                                            ptr(self.n_is) if sif else None, ptr(self.flags) if sif else None,
                                            ptr(self.storage_of) if sif else None, current_stream()),
              "gsl_fused_raster_fwd")
        if not sif:  # (sorting forward: the counters stay set until the backward clears them)
            self._counters_dirty = False
        self._hits_valid = self.record_hits and self.hits is not None
        if self.long_min:
            check(self.lib.gsl_long_raster_fwd(ptr(self.Q0), ptr(self.Q1), ptr(self.Q2), self.D, int(self.ed), self.W,
                                               self.H, self.tw, self.th, self.ty0, self.ty1, ptr(self.offs),
                                               ptr(self.flatten_ids), self.capacity, ptr(self.render), ptr(self.alphas),
                                               ptr(self.last_ids), self.row0, self.row1, ptr(self.Qh),
                                               ptr(self.hits) if (self.record_hits and self.hits is not None) else None,
                                               self.long_min, ptr(self.long_ws), self.long_ws_bytes, self.max_seg,
                                               int(self.bins is not None), current_stream()), "gsl_long_raster_fwd")

    def _raster_bwd(self, v_render: Tensor, v_alphas: Tensor, tracking_loss=None) -> None:
        common = (ptr(self.Q0), ptr(self.Q1), ptr(self.Q2), self.D, int(self.ed), self.W, self.H, self.tw, self.th,
                  self.ty0, self.ty1, ptr(self.offs), ptr(self.flatten_ids), self.capacity, ptr(self.render),
                  ptr(self.alphas), ptr(self.last_ids), ptr(v_render), ptr(v_alphas))
        clear = self.sorts_in_forward() and self._counters_dirty  # the sorting forward left the tile counters set
        if self.tiny:
            loss = tracking_loss if tracking_loss is not None else (None, 0.0, 0.0, None)
            check(self.lib.gsl_tiny_raster_bwd(*common, ptr(self.trec), ptr(self.vcT), self.row0, self.row1,
                                               ptr(self.flags), self.long_min, ptr(loss[0]), float(loss[1]),
                                               float(loss[2]), ptr(loss[3]), ptr(self.ws) if clear else None,
                                               current_stream()), "gsl_tiny_raster_bwd")
            # (pass 2, the fold of the slabs into gradient rows, runs inside the projection backward)
        else:
            check(self.lib.gsl_fused_raster_bwd(*common, ptr(self.vacc), self.row0, self.row1, ptr(self.Qh),
                                                ptr(self.vrow), ptr(self.hits) if self._hits_valid else None,
                                                ptr(self.hit_counts) if self._hits_valid else None, self.long_min,
                                                ptr(self.ws) if clear else None, current_stream()),
                  "gsl_fused_raster_bwd")
        if clear:
            self._counters_dirty = False
        if self.long_min:  # the segments of the long tiles: rows added to vacc
            check(self.lib.gsl_long_raster_bwd(*common, ptr(self.vacc), self.row0, self.row1, ptr(self.Qh),
                                               ptr(self.hits) if self._hits_valid else None,
                                               self.long_min, ptr(self.long_ws), self.max_seg, current_stream()),
                  "gsl_long_raster_bwd")

    def _project_bwd(self, full: bool, reduce: bool = True) -> None:
        means, quats, scales, opacities, colors, viewmat, K = self._inputs
        if self.vrow is not None and not self.tiny:  # deterministic general backward: rows per intersection
            det = (ptr(self.vrow), ptr(self.keys), ptr(self.offs), ptr(self.Q0), self.tw, self.th, self.ty0, self.ty1,
                   self.capacity)
        else:  # (the tiny-splat backward has no atomics to begin with)
            det = (None, None, None, None, 0, 0, 0, 0, 0)
        tiny = (ptr(self.trec), ptr(self.vcT)) if self.tiny else (None, None)
        if tiny[0] is not None:  # the projection backward folds the slabs itself: it needs the records
            det = (None, None, None, ptr(self.Q0), 0, 0, 0, 0, 0)
        check(self.lib.gsl_fused_project_bwd(
            ptr(means), ptr(quats), ptr(scales), ptr(opacities), ptr(colors) if self.rgb else None, self.sh_degree,
            self.K_sh, ptr(viewmat), ptr(K), self.N, self.W, self.H, self.eps2d, int(self.antialiased), self.D,
            ptr(self.radii), ptr(self.Q1), ptr(self.comps),
            ptr(self.vacc) if (not self.tiny or self.long_min) else None,
            ptr(self.v_means) if full else None, ptr(self.v_quats) if full else None,
            ptr(self.v_scales) if full else None, ptr(self.v_opacities) if full else None,
            ptr(self.v_colors) if (full and self.rgb) else None, ptr(self.v_viewmat), ptr(self.ws), self.ws_bytes,
            self.n_tiles, *det, *tiny, int(reduce), ptr(self.vc_state) if (full and self.rgb) else None,
            current_stream()), "gsl_fused_project_bwd")

    # ------------------------------------------------------------------ forward / backward
    def forward(self, means: Tensor, quats: Tensor, scales: Tensor, opacities: Tensor, colors: Optional[Tensor],
                viewmat: Tensor, K: Tensor) -> Tuple[Tensor, Tensor]:
        """Render into the context's buffers (valid until the next forward).  Inputs: contiguous fp32
        device tensors; viewmat [4,4], K [3,3].  No allocation, no host sync."""
        assert self.keys is not None, "call calibrate() (or pass capacity=) before forward()"
        means, quats, scales, opacities, colors = self._place(means, quats, scales, opacities, colors)
        self._project(means, quats, scales, opacities, colors, viewmat, K)
        self._bin()
        self._raster_fwd()
        self._inputs = (means, quats, scales, opacities, colors, viewmat, K)
        self.generation += 1
        return self.render, self.alphas

    def forward_checked(self, means, quats, scales, opacities, colors, viewmat, K) -> Optional[str]:
        """forward() with the overflow check placed where gsplat has its own synchronisation: after projection and
        binning (cheap kernels), before compositing.  Returns None and the rendered buffers are valid, or the reason
        the lists are incomplete (compositing skipped: grow the buffers / calibrate() and call again)."""
        assert self.keys is not None, "call calibrate() (or pass capacity=) before forward()"
        means, quats, scales, opacities, colors = self._place(means, quats, scales, opacities, colors)
        self._project(means, quats, scales, opacities, colors, viewmat, K)
        self._bin()
        why = self.overflow_status()
        if why is not None:
            return why  # (the tile counters stay uncleared: _project / calibrate() zero them before the next use)
        self._raster_fwd()
        self._inputs = (means, quats, scales, opacities, colors, viewmat, K)
        self.generation += 1
        return None

    def overflow_status(self) -> Optional[str]:
        """Host sync (one 32-byte copy): None, or why the last forward's lists are incomplete."""
        st = self.status.tolist()
        self.last_n_isects = min(int(st[4]), self.capacity)
        if st[4] > self.capacity:
            return f"intersection capacity exceeded ({st[4]} > {self.capacity})"
        if self.bins is not None and st[1]:
            return f"a tile list outgrew its bin ({st[2]} > {self.bin_cap})"
        if self.bins is not None and st[3]:
            return "tile counters were not cleared"
        if self.tiny and st[0]:
            return "a splat outgrew the tiny-splat backward"
        if self.long_ws is not None and self.long_overflowed():
            return "long tile lists outgrew their workspace"
        return None

    def can_fuse_tracking_loss(self) -> bool:
        """backward(tracking_loss=...) is available: tiny-splat backward over the whole frame, a depth channel."""
        return bool(self.tiny and self.ty0 == 0 and self.ty1 == self.th and self.row0 == 0 and self.row1 >= self.H
                    and self.D in (1, 4))

    def backward(self, v_render: Tensor, v_alphas: Tensor, full: Optional[bool] = None,
                 reduce_viewmat: bool = True, tracking_loss=None) -> Dict[str, Tensor]:
        """vjp of the last forward.  Returns the context's gradient buffers: always ``viewmat``
        ([4,4], row 3 zero); with full gradients also means/quats/scales/opacities/colors -- in STORAGE order when the
        context placed the Gaussians in tile order (``order_ids`` is not None: row p belongs to the caller's Gaussian
        order_ids[p]; grads_in_input_order() un-permutes).
        ``reduce_viewmat=False`` skips the last launch: the pose gradient stays as ``viewmat_rows()`` for
        gsl_pose_step / gsl_pack_pose_reduce, which sum them in the same fixed order (``viewmat`` is then stale).
        ``tracking_loss=(depth_gt [H,W], depth_lambda, edge_lambda, partials [tiles,2])`` (can_fuse_tracking_loss()):
        the compositing backward computes gsl_tracking_loss's loss and gradient itself -- it WRITES v_render's depth
        channel and the partials -- instead of reading an upstream gradient a separate launch left there."""
        assert self._inputs is not None, "forward() first"
        full = self.full_grads if full is None else full
        assert not full or self.full_grads, "context was built with full_grads=False"
        assert tracking_loss is None or self.can_fuse_tracking_loss()
        self._raster_bwd(v_render, v_alphas, tracking_loss)
        self._project_bwd(full, reduce_viewmat)
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
        if full:
            g = rc.grads_in_input_order(g)
        return (None, g["means"] if (full and ni[1]) else None, g["quats"] if (full and ni[2]) else None,
                g["scales"] if (full and ni[3]) else None, g["opacities"] if (full and ni[4]) else None,
                g["colors"] if (full and ni[5] and rc.rgb) else None, g["viewmat"] if ni[6] else None, None)


def pack_pose_reduce(v_viewmat: Tensor, out16: Tensor, loss_partials: Optional[Tensor] = None,
                     normal_sum: Optional[Tensor] = None) -> None:
    """One rank's contribution to the per-iteration all-reduce: out16[0:12] = v_viewmat rows 0..2,
    out16[12:14] = the sums of loss_partials[n,2] (or 0), out16[14] = normal_sum (or 0), written by a kernel of the
    library -- no torch op, so a captured iteration holds this library's launches only."""
    n = 0 if loss_partials is None else loss_partials.numel() // 2
    check(load_library().gsl_pack_pose_reduce(ptr(v_viewmat), None, 0, None, None, ptr(loss_partials), n,
                                              ptr(normal_sum), ptr(out16), current_stream()), "gsl_pack_pose_reduce")


def time_stages(rc: RenderContext, inputs, v_render: Tensor, v_alphas: Tensor, full: bool, steps: int = 20) -> Dict:
    """Average duration (ms) of each of the five stage calls of one iteration, measured with HIP events
    recorded on the launch stream around every C-ABI call (same launches as forward()/backward())."""
    names = ["project_fwd", "bin", "raster_fwd", "raster_bwd", "project_bwd"]
    acc = {n: 0.0 for n in names}

    def ev():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    inputs = tuple(rc._place(*inputs[:5])) + tuple(inputs[5:])
    for it in range(steps + 2):
        marks = [ev()]
        rc._project(*inputs)
        marks.append(ev())
        rc._bin()
        marks.append(ev())
        rc._raster_fwd()
        rc._inputs = tuple(inputs)
        marks.append(ev())
        rc._raster_bwd(v_render, v_alphas)
        marks.append(ev())
        rc._project_bwd(full)
        marks.append(ev())
        torch.cuda.synchronize()
        if it >= 2:
            for i, n in enumerate(names):
                acc[n] += marks[i].elapsed_time(marks[i + 1])
    return {n: acc[n] / steps for n in names}
