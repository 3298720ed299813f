"""GraphTracker -- GsplatLoc's per-frame pose optimisation as one HIP graph per iteration.

Same arithmetic as ``Runner.train``'s inner loop (/root/reference/src/my_gsplat/gs_trainer_total.py:79-267)
and as ``my_gsplat.PoseTracker``; the difference is where the glue runs: the loss (loss.py:10-59), the pose
chain (model.py:79-82, transform.py:50-66, geometry.py:12-20), both Adam optimisers, the exponential LR
decay and the early-stop bookkeeping (data/base.py:34-43) live in three device kernels
(csrc/tracker.hip), so one iteration is a fixed sequence of ~13 launches with no allocation and no host
synchronisation, replayed as a HIP graph.  The host looks at the device-side "stopped" flag every
``poll`` iterations only.

Several GPUs (``rows=``, ``group=``): every rank tracks the same pose on its tile-row strip (plus one halo
tile row), the 12 pose-gradient entries and the two loss sums are summed with ONE all-reduce of 16 floats
per iteration, and every rank applies the identical update.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
from torch import Tensor

from ._lib import check, current_stream, load_library, ptr
from .context import RenderContext
from .my_gsplat.trainer import TrackerConfig, TrackResult
from .my_gsplat.utils import rgb_to_sh
from .parallel import halo_rows


class GraphTracker:
    def __init__(self, N: int, width: int, height: int, config: TrackerConfig = TrackerConfig(), device="cuda",
                 render_mode: str = "RGB+ED", rows: Optional[Tuple[int, int]] = None, group=None,
                 use_graph: bool = True, poll: int = 25):
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
        self.row0, self.row1 = self.rows[0] * 16, min(self.rows[1] * 16, self.H)
        sh_deg = config.gs.sh_degree
        self.K_sh = (sh_deg + 1) ** 2
        self.rc = RenderContext(self.N, self.W, self.H, render_mode, sh_degree=sh_deg, K_sh=self.K_sh, device=self.dev,
                                near_plane=config.gs.near_plane, far_plane=config.gs.far_plane,
                                tile_rows=self.render_rows, full_grads=False)
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
        self.n_partials = ((self.row1 - self.row0) * self.W + 255) // 256
        self.partials = torch.zeros(max(self.n_partials, 1) * 2, dtype=f32, device=d)
        self.loss_hist = torch.zeros(max(config.max_steps, 1), dtype=f32, device=d)
        self.reduce_buf = torch.zeros(16, dtype=f32, device=d)  # 12 pose-gradient entries + 2 loss sums
        self.graph = None
        self._side = torch.cuda.Stream(device=d)

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
        self.gt_depth.copy_(src_depth.reshape(self.H, self.W))
        self.init_c2w.copy_(tar_c2w)
        self.gt_c2w.copy_(src_c2w)
        self.loss_hist.zero_()
        self.v_render.zero_()
        cam = self.cfg.camera
        check(self.lib.gsl_pose_init(ptr(self.pose_f), ptr(self.pose_i), ptr(self.init_c2w), cam.quat_lr, cam.trans_lr,
                                     ptr(self.c2w), ptr(self.viewmat), current_stream()), "gsl_pose_init")
        self.rc.calibrate(self.means, self.quats, self.scales, self.opac, self.sh, self.viewmat, self.K, headroom=1.5)
        self.graph = None  # capacity buffers may have been reallocated

    # ------------------------------------------------------------------ one iteration
    def _iteration(self) -> None:
        cfg, lib = self.cfg, self.lib
        st = current_stream()
        self.rc.forward(self.means, self.quats, self.scales, self.opac, self.sh, self.viewmat, self.K)
        edge_w = 1.0 - cfg.depth_lambda - cfg.normal_lambda
        check(lib.gsl_tracking_loss(ptr(self.rc.render), self.rc.D, ptr(self.gt_depth), self.W, self.H, self.row0,
                                    self.row1, cfg.depth_lambda, edge_w, ptr(self.v_render), ptr(self.partials), None,
                                    ptr(self.loss_ws), self.loss_ws_bytes, st), "gsl_tracking_loss")
        g = self.rc.backward(self.v_render, self.v_alphas, full=False)
        loss_sums = None
        v_viewmat = g["viewmat"]
        if self.group is not None:
            import torch.distributed as dist
            self.reduce_buf[:12].copy_(v_viewmat.reshape(16)[:12])
            self.reduce_buf[12:14].copy_(self.partials.view(-1, 2).sum(0))
            dist.all_reduce(self.reduce_buf, group=self.group)  # THE collective: 16 floats
            v_viewmat = self.reduce_buf
            loss_sums = self.reduce_buf[12:14]
        cam = cfg.camera
        gamma = 0.2 ** (1.0 / cfg.max_steps)
        check(lib.gsl_pose_step(ptr(self.pose_f), ptr(self.pose_i), ptr(v_viewmat), ptr(self.partials),
                                self.n_partials, ptr(loss_sums), ptr(self.gt_c2w), self.W, self.H, cfg.depth_lambda,
                                edge_w, 0.9, 0.999, 1e-8, cam.quat_opt_reg, cam.trans_opt_reg, gamma, cfg.min_step,
                                cfg.patience, int(cfg.early_stop), cfg.max_steps, ptr(self.c2w), ptr(self.viewmat),
                                ptr(self.loss_hist), st), "gsl_pose_step")

    def _capture(self) -> None:
        state = (self.pose_f.clone(), self.pose_i.clone(), self.c2w.clone(), self.viewmat.clone(), self.loss_hist.clone())
        with torch.cuda.stream(self._side):
            self._iteration()  # warm-up outside capture
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=self._side):
                self._iteration()
        torch.cuda.synchronize()
        # restore the optimiser state consumed by the warm-up and by capture
        for dst, src in zip((self.pose_f, self.pose_i, self.c2w, self.viewmat, self.loss_hist), state):
            dst.copy_(src)

    # ------------------------------------------------------------------ frame loop
    def run(self) -> TrackResult:
        """Optimise the loaded frame until early stop or max_steps.  Returns the reference's read-outs."""
        use_graph = self.use_graph and self.group is None
        if use_graph and self.graph is None:
            self._capture()
        done = 0
        while done < self.cfg.max_steps:
            n = min(self.poll, self.cfg.max_steps - done)
            for _ in range(n):
                if use_graph:
                    self.graph.replay()
                else:
                    self._iteration()
            done += n
            if int(self.pose_i[2].item()):  # stopped (early stop or max_steps) -- the only host sync
                break
        self.rc.check_capacity()
        pi = self.pose_i.tolist()
        pf = self.pose_f.tolist()
        res = TrackResult()
        res.steps = pi[0]
        res.losses = self.loss_hist[:pi[0]].tolist()
        res.best_loss, res.best_depth_loss, res.best_silhouette_loss = pf[23], pf[24], pf[25]
        res.best_eT, res.best_eR = pf[26], pf[27]
        res.final_c2w = self.c2w.clone()
        return res
