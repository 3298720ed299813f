"""Sequence evaluation driver -- the role of /root/reference/src/GsplatLoc_eval.py + Runner.train +
eval/logger.py:258-304 without W&B: track every consecutive frame pair of a room from the ground-truth pose
of frame i, record the error of the minimum-loss iterate against frame i+1, and report
ATE := RMSE of the translation errors, AAE := RMSE of the rotation errors (eval/utils.py:113-119), in the
res.json layout of /root/reference/docs/res.json.

    python -m gsplatloc_amd.eval --dataset Replica --rooms room0 --root /data/Replica --num-iters 200
"""
from __future__ import annotations

import argparse
import json
import math
import time
from typing import Dict, List, Optional

import torch

from .data import Parser
from .graph_tracker import GraphTracker
from .my_gsplat import TrackerConfig, init_gs_scales


def rmse(values: List[float]) -> float:
    """eval/utils.py:113-119."""
    return math.sqrt(sum(v * v for v in values) / max(len(values), 1))


def evaluate_room(parser: Parser, num_iters: int = 2000, max_frames: Optional[int] = 1998, verbose: bool = False,
                  profile: bool = False) -> Dict:
    """Runner.train (gs_trainer_total.py:45-282): frames 0..min(len, 1998).  profile=True: also the wall time per phase
    of a frame (synchronising at the phase boundaries): reading + back-projection + PCA normalisation, the query-depth
    render, the kNN scale initialisation, load_frame (copies + calibration pass), and the optimisation itself (graph
    capture + replays + polls) -- `phase_seconds` in the result."""
    cfg = TrackerConfig(max_steps=num_iters)
    phases = {} if profile else None
    parser.phase_seconds = phases

    def tick():
        if profile:
            torch.cuda.synchronize()
        return time.perf_counter()

    tracker = None
    eTs, eRs, steps = [], [], []
    long_frames = 0  # frames with a tile list long enough to be split over workgroups (a pile of invalid-depth points)
    t0 = time.perf_counter()
    n = len(parser) if max_frames is None else min(len(parser), max_frames)
    for i in range(n):
        d = parser[i]
        H, W = d.src_depth.shape[1:3]
        if tracker is None or tracker.N != d.tar_points.shape[0]:
            # expected depth only: the reference's loop asks for "RGB+ED" (model.py:195-213) and reads renders[..., 3:4]
            # alone (gs_trainer_total.py:104-123) -- the SH colours, their records and three of four composited channels
            # are work whose result nobody looks at.  Same depth, same loss, same pose; 2 % more
            # iterations per second at 102 k Gaussians, 20 % at 816 k.  render_mode="RGB+ED" is the literal call.
            tracker = GraphTracker(d.tar_points.shape[0], W, H, cfg, device=d.tar_points.device, render_mode="ED")
        ta = tick()
        scales = init_gs_scales(d.tar_points)
        tb = tick()
        tracker.load_frame(d.tar_points, d.colors, scales, d.src_depth, d.tar_c2w, d.src_c2w, parser.K)
        tc = tick()
        res = tracker.run()
        if profile:
            td = tick()
            for k, v in (("knn_scales", tb - ta), ("load_frame_calibrate", tc - tb), ("optimise", td - tc)):
                phases[k] = phases.get(k, 0.0) + v
        long_frames += int(tracker.rc.long_min > 0)
        # early stop never fired before min_step: fall back to the last iterate's errors, as the reference would log inf
        eTs.append(res.best_eT)
        eRs.append(res.best_eR)
        steps.append(res.steps)
        if verbose:
            print(f"frame {i}: steps {res.steps} loss {res.best_loss:.3e} eT {res.best_eT:.3e} eR {res.best_eR:.3e}")
    dt = time.perf_counter() - t0
    parser.phase_seconds = None
    finite = [(a, b) for a, b in zip(eTs, eRs) if math.isfinite(a) and math.isfinite(b)]
    if not finite:  # e.g. num_iters <= 101: the minimum-loss read-out starts after step 100 and never fired
        return {"ATE": float("nan"), "AAE": float("nan"), "frames": n, "frames_with_result": 0,
                "mean_steps": sum(steps) / max(len(steps), 1), "seconds": dt, "frames_per_s": n / dt if dt > 0 else None,
                "error": "no frame produced a minimum-loss read-out (num_iters must exceed 101)"}
    return {"ATE": rmse([a for a, _ in finite]), "AAE": rmse([b for _, b in finite]), "frames": n,
            "frames_with_result": len(finite), "mean_steps": sum(steps) / max(len(steps), 1),
            "frames_with_long_lists": long_frames, "seconds": dt, "frames_per_s": n / dt if dt > 0 else None,
            **({"phase_seconds": phases} if profile else {})}


def main(argv=None):
    ap = argparse.ArgumentParser(description="GsplatLoc evaluation on the MI355X rasterizer")
    ap.add_argument("--dataset", choices=["Replica", "TUM"], default="Replica")
    ap.add_argument("--rooms", nargs="+", default=["room0"])
    ap.add_argument("--root", required=True, help="dataset root (contains <room>/ for Replica, rgbd_dataset_* for TUM)")
    ap.add_argument("--num-iters", type=int, default=2000)
    ap.add_argument("--max-frames", type=int, default=1998)
    ap.add_argument("--no-normalize", action="store_true")
    ap.add_argument("--out", default="res.json")
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args(argv)
    out = {}
    for room in a.rooms:
        parser = Parser(a.dataset, room, normalize=not a.no_normalize, input_folder=a.root)
        r = evaluate_room(parser, a.num_iters, a.max_frames, a.verbose)
        out[room] = {"gsplatloc_amd": r}
        print(room, json.dumps(r))
    with open(a.out, "w") as f:
        json.dump(out, f, indent=2)


if __name__ == "__main__":
    main()
