"""GraphTracker -- GsplatLoc's per-frame pose optimisation as one HIP graph per iteration.

Same arithmetic as ``Runner.train``'s inner loop (/root/reference/src/my_gsplat/gs_trainer_total.py:79-267)
and as ``my_gsplat.PoseTracker``; the difference is where the glue runs: the loss (loss.py:10-59), the pose
chain (model.py:79-82, transform.py:50-66, geometry.py:12-20), both Adam optimisers, the exponential LR
decay and the early-stop bookkeeping (data/base.py:34-43) live in two device kernels (loss, pose step:
csrc/tracker.hip), so one iteration is a fixed sequence of 5 launches (projection; compositing forward, which sorts its own
tile's bin; compositing backward, which computes its tile's loss; projection backward; pose step -- up to 8 with several
ranks, a normal term or long tile lists) with no allocation and no host synchronisation, replayed as a HIP graph.  The host looks at the device-side "stopped" flag every
``poll`` iterations only.

Several GPUs (``rows=``, ``group=``): every rank tracks the same pose on its tile-row strip (plus ONE pixel row
of halo on each interior side), the 12 pose-gradient entries and the two loss sums are packed by a kernel of the
library into one 16-float buffer, summed with ONE all-reduce per iteration, and every rank applies the identical
update.  The iteration is then two graph-replayable halves around the collective: [render, loss, backward, pack]
and [pose step].

Strips keep only the Gaussians that can reach them: per frame every rank projects all N Gaussians once at
the initial pose and keeps those whose splat touches its rendered tile rows widened by a guard band of ``guard_tiles``
tile rows (parallel.gaussians_for_strip); the iteration then projects, bins and back-propagates the kept ones only.
The guard band is CHECKED, not assumed: at every poll a projection of all N at the current pose tells whether a Gaussian
outside the kept set now reaches the strip; the answer is MAX-reduced with the overflow flags, and on a violation every
rank doubles its guard band, re-buckets and re-runs the frame from its initial pose (a pile of invalid-depth points
1.5 cm in front of the camera moves 20 px per iteration: NOTES.md).

Overflow handling: the kernels never drop work silently.  A splat that outgrows the tiny-splat backward raises a
sticky device flag, and the intersection count is compared with the buffer capacity; both are read at the
existing poll of the "stopped" flag, and on either the frame is re-run from its initial pose with the general
backward / a larger capacity.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

import os

import torch
from torch import Tensor

from ._lib import check, current_stream, load_library, ptr
from .context import RenderContext
from .my_gsplat.trainer import TrackerConfig, TrackResult
from .my_gsplat.utils import rgb_to_sh
from .parallel import gaussians_for_strip, halo_pixel_rows, halo_rows


class GraphTracker:
    def __init__(self, N: int, width: int, height: int, config: TrackerConfig = TrackerConfig(), device="cuda",
                 render_mode: str = "RGB+ED", rows: Optional[Tuple[int, int]] = None, group=None,
                 use_graph: bool = True, poll: int = 25, prune: Optional[bool] = None, guard_tiles: int = 1):
        assert render_mode in ("RGB+ED", "ED"), "the tracker's loss reads expected depth"
        self.lib = load_library()
        self.cfg = config
        self.N, self.W, self.H = int(N), int(width), int(height)
        self.dev = torch.device(device)
        self.mode = render_mode
        self.group = group
        self.use_graph = use_graph
        self.poll = int(poll)
        th = (self.H + 15) // 16
        self.rows = rows if rows is not None else (0, th)
        self.render_rows = halo_rows(self.rows, th) if rows is not None else (0, th)
        self.pixel_rows = halo_pixel_rows(self.rows, self.H) if rows is not None else (0, self.H)
        self.row0, self.row1 = self.rows[0] * 16, min(self.rows[1] * 16, self.H)
        sh_deg = config.gs.sh_degree
        self.K_sh = (sh_deg + 1) ** 2
        # a strip keeps only the Gaussians that can reach it (+ guard band, checked at every poll): _bucket()
        self.prune = (rows is not None) if prune is None else bool(prune)
        self.guard0 = self.guard = max(int(guard_tiles), 0)
        self.kept = self.kept_mask = self._active = self._check = None
        self.rebuckets = 0  # guard-band violations recovered from (all frames)
        self.rc = self._make_rc(self.N)
        f32 = torch.float32
        d = self.dev
        self.means = torch.zeros(self.N, 3, dtype=f32, device=d)
        self.quats = torch.tensor([1.0, 0.0, 0.0, 0.0], device=d).repeat(self.N, 1).contiguous()
        self.scales = torch.zeros(self.N, 3, dtype=f32, device=d)
        self.opac = torch.ones(self.N, dtype=f32, device=d)
        self.sh = torch.zeros(self.N, self.K_sh, 3, dtype=f32, device=d)
        self.K = torch.zeros(3, 3, dtype=f32, device=d)
        self.gt_depth = torch.zeros(self.H, self.W, dtype=f32, device=d)
        self.init_c2w = torch.eye(4, dtype=f32, device=d)
        self.gt_c2w = torch.eye(4, dtype=f32, device=d)
        self.pose_f = torch.zeros(32, dtype=f32, device=d)
        self.pose_i = torch.zeros(4, dtype=torch.int32, device=d)
        self.c2w = torch.eye(4, dtype=f32, device=d)
        self.viewmat = torch.eye(4, dtype=f32, device=d)
        self.v_render = torch.zeros(self.H, self.W, self.rc.D, dtype=f32, device=d)
        self.v_alphas = torch.zeros(self.H, self.W, 1, dtype=f32, device=d)
        self.loss_ws_bytes = self.lib.gsl_loss_ws_bytes(self.W, self.H)
        self.loss_ws = torch.zeros(self.loss_ws_bytes, dtype=torch.uint8, device=d)
        self.n_partials = self.lib.gsl_loss_n_partials(self.W, self.H, self.row0, self.row1)
        self.partials = torch.zeros(max(self.n_partials, 1) * 2, dtype=f32, device=d)
        self.loss_hist = torch.zeros(max(config.max_steps, 1), dtype=f32, device=d)
        self.reduce_buf = torch.zeros(16, dtype=f32, device=d)  # 12 pose-gradient entries + 2 loss sums + row cosines
        self.normal_sum = self.normal_ws = None
        if config.normal_lambda != 0.0:  # the term the reference keeps switched off (gs_trainer_total.py:138-143)
            self.normal_ws_bytes = self.lib.gsl_normal_ws_bytes(self.W, self.H)
            self.normal_ws = torch.zeros(self.normal_ws_bytes, dtype=torch.uint8, device=d)
            self.normal_sum = torch.zeros(1, dtype=f32, device=d)
        self._host16 = None
        self._intrinsics = (1.0, 1.0, 0.0, 0.0)
        self.graph = self.graph_tail = None
        self.collective_captured = False  # True: the all-reduce is a node of self.graph (one replay per iteration)
        self._side = torch.cuda.Stream(device=d)
        self.headroom = 1.5

    def _make_rc(self, n: int) -> RenderContext:
        c = self.cfg
        return RenderContext(n, self.W, self.H, self.mode, sh_degree=c.gs.sh_degree, K_sh=self.K_sh, device=self.dev,
                             near_plane=c.gs.near_plane, far_plane=c.gs.far_plane, tile_rows=self.render_rows,
                             pixel_rows=self.pixel_rows, full_grads=False,
                             sort_in_forward=self.group is None)  # (every forward of an iteration has its backward)

    # ------------------------------------------------------------------ strip buckets and their guard band
    def _gauss(self):
        """The per-Gaussian inputs of the iteration: all N, or the strip's kept subset."""
        return self._active if self._active is not None else (self.means, self.quats, self.scales, self.opac, self.sh)

    def _reach(self, guard: int) -> Tensor:
        """[N] bool: whose splat touches this rank's rendered tile rows widened by `guard` tile rows, at the CURRENT pose
        (one projection of all N Gaussians through the library; radii and centres only)."""
        if self._check is None:
            c = self.cfg
            self._check = RenderContext(self.N, self.W, self.H, "ED", sh_degree=None, device=self.dev,
                                        near_plane=c.gs.near_plane, far_plane=c.gs.far_plane, tile_rows=self.render_rows,
                                        full_grads=False, reorder=False)
        chk = self._check
        chk._project(self.means, self.quats, self.scales, self.opac, None, self.viewmat, self.K)
        idx = gaussians_for_strip(chk.Q0[:, 0:2], chk.radii, self.render_rows, guard_tiles=guard)
        mask = torch.zeros(self.N, dtype=torch.bool, device=self.means.device)
        mask[idx] = True
        return mask

    def _bucket(self) -> None:
        """Per frame (and again after a guard-band violation): the kept set at the initial pose, a render context of its
        size, calibrated.  Nothing visible at all (or launches refused: the CPU tests) keeps everything."""
        mask = self._reach(self.guard)
        idx = mask.nonzero(as_tuple=True)[0]
        if idx.numel() == 0:
            mask[:] = True
            idx = mask.nonzero(as_tuple=True)[0]
        self.kept, self.kept_mask = idx, mask
        self._active = tuple(t[idx].contiguous() for t in (self.means, self.quats, self.scales, self.opac, self.sh))
        n = int(idx.numel())
        if self.rc.N != n:
            self.rc = self._make_rc(n)
        got = self.rc.calibrate(*self._active, self.viewmat, self.K, headroom=self.headroom)
        # the kept set must reproduce the full projection's lists of the rendered rows (the check context just binned all N
        # Gaussians over exactly those rows at this pose)
        want = int(self._check.n_is.item())
        assert got == want or want == 0, f"strip {self.render_rows}: {got} intersections from the kept set, {want} from all"

    def _band_violations(self) -> int:
        """Gaussians outside the kept set whose splat reaches the rendered rows at the current pose (0 = the band held)."""
        if not self.prune or self.kept_mask is None or bool(self.kept_mask.all()):
            return 0
        return int((self._reach(0) & ~self.kept_mask).sum())

    # ------------------------------------------------------------------ frame setup
    def load_frame(self, tar_points: Tensor, colors: Tensor, scales: Tensor, src_depth: Tensor, tar_c2w: Tensor,
                   src_c2w: Tensor, K: Tensor) -> None:
        """Copy one frame pair into the tracker's persistent buffers (shapes fixed at construction):
        Gaussians = tar_points [N,3] with isotropic scales [N,3] and colours [N,3]; target depth
        src_depth [...,H,W,...]; initial pose tar_c2w; reference pose src_c2w (error read-out only)."""
        assert tar_points.shape == (self.N, 3), tar_points.shape
        self.means.copy_(tar_points)
        self.scales.copy_(scales)
        self.sh.zero_()
        self.sh[:, 0, :] = rgb_to_sh(colors.to(self.dev))
        self.opac.copy_(torch.sigmoid(torch.logit(torch.full((self.N,), self.cfg.gs.init_opa, device=self.dev))))
        self.K.copy_(K)
        Kh = K.detach().cpu()
        self._intrinsics = (float(Kh[0, 0]), float(Kh[1, 1]), float(Kh[0, 2]), float(Kh[1, 2]))
        self.gt_depth.copy_(src_depth.reshape(self.H, self.W))
        self.init_c2w.copy_(tar_c2w)
        self.gt_c2w.copy_(src_c2w)
        self.loss_hist.zero_()
        self.v_render.zero_()
        cam = self.cfg.camera
        check(self.lib.gsl_pose_init(ptr(self.pose_f), ptr(self.pose_i), ptr(self.init_c2w), cam.quat_lr, cam.trans_lr,
                                     ptr(self.c2w), ptr(self.viewmat), current_stream()), "gsl_pose_init")
        self.guard = self.guard0
        if self.prune:
            self._bucket()
        else:
            self.rc.calibrate(self.means, self.quats, self.scales, self.opac, self.sh, self.viewmat, self.K,
                              headroom=self.headroom)
        self.graph = self.graph_tail = None  # capacity buffers may have been reallocated

    # ------------------------------------------------------------------ one iteration
    def _render_and_loss(self) -> None:
        """Forward, loss, backward; with several ranks also the pack of this rank's 16 floats.  Library launches
        and memsets only: safe to capture."""
        cfg, lib = self.cfg, self.lib
        st = current_stream()
        self.rc.forward(*self._gauss(), self.viewmat, self.K)
        edge_w = 1.0 - cfg.depth_lambda - cfg.normal_lambda
        # one rank, no normal term, tiny-splat backward: the compositing backward computes the loss of its own tile and
        # its gradient (gsl_tiny_raster_bwd(..., loss_depth_gt, ...)): one launch fewer per iteration
        fuse = (self.group is None and self.normal_ws is None and self.rc.can_fuse_tracking_loss()
                and os.environ.get("GSLOC_FUSE_LOSS", "1") != "0")
        if fuse:
            self.rc.backward(self.v_render, self.v_alphas, full=False, reduce_viewmat=False,
                             tracking_loss=(self.gt_depth, cfg.depth_lambda, edge_w, self.partials))
            return
        check(lib.gsl_tracking_loss(ptr(self.rc.render), self.rc.D, ptr(self.gt_depth), self.W, self.H, self.row0,
                                    self.row1, cfg.depth_lambda, edge_w, ptr(self.v_render), ptr(self.partials), None,
                                    ptr(self.loss_ws), self.loss_ws_bytes, st), "gsl_tracking_loss")
        if self.normal_ws is not None:
            check(lib.gsl_normal_loss(ptr(self.rc.render), self.rc.D, ptr(self.gt_depth), self.W, self.H, self.row0,
                                      self.row1, *self._intrinsics, cfg.normal_lambda, ptr(self.v_render),
                                      ptr(self.normal_sum), ptr(self.normal_ws), self.normal_ws_bytes, st),
                  "gsl_normal_loss")
        # the pose gradient stays as partial rows: the pack kernel / the pose step sum them (one launch fewer)
        self.rc.backward(self.v_render, self.v_alphas, full=False, reduce_viewmat=False)
        if self.group is not None:
            rows, n_rows = self.rc.viewmat_rows()
            check(lib.gsl_pack_pose_reduce(None, rows, n_rows, ptr(self.viewmat), ptr(self.K), ptr(self.partials),
                                           self.n_partials, ptr(self.normal_sum), ptr(self.reduce_buf), st),
                  "gsl_pack_pose_reduce")

    def _pose_step(self) -> None:
        cfg, lib = self.cfg, self.lib
        if self.group is not None:  # summed over the ranks: gradient in [0,12), loss sums in [12,14)
            v_viewmat, loss_sums = ptr(self.reduce_buf), self.reduce_buf.data_ptr() + 12 * 4
            rows, n_rows = None, 0
        else:
            v_viewmat, loss_sums = None, None
            rows, n_rows = self.rc.viewmat_rows()
        cam = cfg.camera
        edge_w = 1.0 - cfg.depth_lambda - cfg.normal_lambda
        gamma = 0.2 ** (1.0 / cfg.max_steps)
        check(lib.gsl_pose_step(ptr(self.pose_f), ptr(self.pose_i), v_viewmat, rows, n_rows, ptr(self.K),
                                ptr(self.partials), self.n_partials, loss_sums, ptr(self.normal_sum), ptr(self.gt_c2w), self.W, self.H,
                                cfg.depth_lambda, edge_w, cfg.normal_lambda, 0.9, 0.999, 1e-8, cam.quat_opt_reg,
                                cam.trans_opt_reg, gamma, cfg.min_step,
                                cfg.patience, int(cfg.early_stop), cfg.max_steps, ptr(self.c2w), ptr(self.viewmat),
                                ptr(self.loss_hist), current_stream()), "gsl_pose_step")

    def _collective(self) -> None:
        """THE collective of the path: 16 floats summed over the ranks (RCCL on device buffers; the one-GPU
        rehearsal's gloo group goes through a pinned host buffer)."""
        import torch.distributed as dist
        if not self.reduce_buf.is_cuda:  # host tensors (CPU tests of the glue)
            dist.all_reduce(self.reduce_buf, group=self.group)
        elif dist.get_backend(self.group) == "gloo":
            if self._host16 is None:
                self._host16 = torch.zeros(16, dtype=torch.float32).pin_memory()
            self._host16.copy_(self.reduce_buf, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            dist.all_reduce(self._host16, group=self.group)
            self.reduce_buf.copy_(self._host16, non_blocking=True)
        else:
            dist.all_reduce(self.reduce_buf, group=self.group)

    def _iteration(self) -> None:
        if self.group is None or (self.graph is not None and self.collective_captured):
            if self.graph is not None:
                self.graph.replay()
            else:
                self._render_and_loss()
                self._pose_step()
            return
        if self.graph is not None:
            self.graph.replay()
        else:
            self._render_and_loss()
        self._collective()
        if self.graph_tail is not None:
            self.graph_tail.replay()
        else:
            self._pose_step()

    def _state(self):
        return (self.pose_f, self.pose_i, self.c2w, self.viewmat, self.loss_hist)

    def _capture(self) -> None:
        """One HIP graph per iteration.  Several ranks over RCCL: the 16-float all-reduce is captured INSIDE that graph
        (torch records the RCCL kernel as a node), so an iteration stays one replay; over gloo (the CPU / one-GPU
        rehearsal) or if the capture is refused, two graphs around an eager collective.
        GSLOC_CAPTURE_COLLECTIVE=0 forces the two-graph form."""
        import os
        state = [t.clone() for t in self._state()]
        want_one = False
        if self.group is not None and self.reduce_buf.is_cuda and os.environ.get("GSLOC_CAPTURE_COLLECTIVE", "1") != "0":
            import torch.distributed as dist
            want_one = dist.get_backend(self.group) == "nccl"
        self.collective_captured = False
        with torch.cuda.stream(self._side):
            self._render_and_loss()  # warm-up outside capture (also creates the communicator on first use)
            if self.group is not None:
                self._collective()
            self._pose_step()
            torch.cuda.synchronize()
            if want_one:
                try:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=self._side, capture_error_mode="thread_local"):
                        self._render_and_loss()
                        self._collective()
                        self._pose_step()
                    self.graph, self.graph_tail, self.collective_captured = g, None, True
                except Exception as exc:  # noqa: BLE001 - any refusal: fall back to the two-graph form
                    import warnings
                    warnings.warn(f"capturing the all-reduce in the iteration graph failed ({type(exc).__name__}: {exc}); "
                                  "using two graphs around an eager collective")
                    torch.cuda.synchronize()
            if not self.collective_captured:
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph, stream=self._side):
                    self._render_and_loss()
                    if self.group is None:
                        self._pose_step()
                if self.group is not None:
                    self.graph_tail = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self.graph_tail, stream=self._side):
                        self._pose_step()
        torch.cuda.synchronize()
        for dst, src in zip(self._state(), state):  # restore what the warm-up and the captures consumed
            dst.copy_(src)

    # ------------------------------------------------------------------ frame loop
    def _poll(self) -> Tuple[int, int, int, int, int, int]:
        """The only host sync of the loop: (stopped, intersections beyond the capacity or 0, a splat outgrew the tiny
        backward, longest tile list that outgrew its bin or 0, segments of long tile lists beyond their workspace or 0,
        Gaussians that left their strip's guard band or 0).  With several ranks the numbers are MAX-reduced over the
        group, so every rank takes the same decision at the same iteration: a rank that re-ran the frame on its own while
        the others returned would pair its all-reduces with those of a different iteration or frame."""
        n_is = int(self.rc.n_is.item())
        local = [int(self.pose_i[2].item()), n_is if n_is > self.rc.capacity else 0, int(self.rc.tiny_overflowed()),
                 int(self.rc.bins_overflowed()), int(self.rc.long_overflowed()), self._band_violations()]
        if self.group is None:
            return tuple(local)
        import torch.distributed as dist
        on_device = self.reduce_buf.is_cuda and dist.get_backend(self.group) != "gloo"
        t = torch.tensor(local, dtype=torch.int64, device=self.dev if on_device else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return tuple(int(x) for x in t.tolist())

    def run(self) -> TrackResult:
        """Optimise the loaded frame until early stop or max_steps.  Returns the reference's read-outs."""
        start = [t.clone() for t in self._state()]
        for attempt in range(8):
            if self.use_graph and self.graph is None:
                self._capture()
            done, redo = 0, False
            while done < self.cfg.max_steps:
                n = min(self.poll, self.cfg.max_steps - done)
                for _ in range(n):
                    self._iteration()
                done += n
                stopped, n_over, tiny_over, bin_over, long_over, band_over = self._poll()
                if n_over or tiny_over or bin_over or long_over or band_over:
                    redo = True
                    break
                if stopped:
                    break
            if not redo:
                break
            # recover: iterations since the last poll ran on truncated lists or a dropped gradient.  Every rank of a
            # group arrives here together (the decision was reduced); the backward variant is switched on all of
            # them, the rank-local buffers grow where they were too small.
            if tiny_over:
                self.rc.use_general_backward()
            own_bins = self.rc.bins_overflowed()
            if own_bins:
                self.rc.grow_bins(own_bins)
            own_long = self.rc.long_overflowed()
            if own_long:
                self.rc.grow_long(own_long)
            n_is = int(self.rc.n_is.item())
            if n_is > self.rc.capacity:
                self.headroom *= 1.5
                self.rc._alloc_isects(int(n_is * self.headroom) + 1024)
            self.graph = self.graph_tail = None
            for dst, src in zip(self._state(), start):
                dst.copy_(src)
            if band_over:
                # a Gaussian outside the kept set reached the strip: the gradient of the iterations since the last poll
                # missed it.  Every rank (the decision was reduced) widens its guard band and buckets again at the
                # frame's initial pose, which the state copy above has just restored.
                self.guard = max(2 * self.guard, 1)
                self.rebuckets += 1
                self._bucket()
        else:
            raise RuntimeError("the frame kept overflowing its buffers (or leaving its guard band) after seven recoveries")
        pi = self.pose_i.tolist()
        pf = self.pose_f.tolist()
        res = TrackResult()
        res.steps = pi[0]
        res.losses = self.loss_hist[:pi[0]].tolist()
        res.best_loss, res.best_depth_loss, res.best_silhouette_loss = pf[23], pf[24], pf[25]
        res.best_eT, res.best_eR = pf[26], pf[27]
        res.final_c2w = self.c2w.clone()
        return res
