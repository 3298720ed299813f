#!/usr/bin/env python3
"""Headline benchmark: Gaussians/s, forward + backward, of the GsplatLoc render hot path.

One "step" = one full pass of the path BASELINE.json names on a batch of synthetic input:
projection + SH colour -> tile binning + per-tile depth sort -> alpha compositing ("RGB+ED",
the reference's call, /root/reference/src/my_gsplat/model.py:195-213) -> backward from a given
depth-channel gradient to every Gaussian input and to the 4x4 view matrix.  Inputs are resident
in HBM before the timed region.  Workload "R" of SURVEY.md 8(d): N = 1,000,000 Gaussians,
1200x680, sigma_px = 1.0 (I/N ~ 2.4 tile intersections per Gaussian).

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 (launched by torch.distributed.run, one rank per GPU over RCCL): screen-tile rows are
split across ranks, every rank renders and back-propagates its strip for all N Gaussians, and
the 12+4 pose-gradient floats are summed with ONE all-reduce per step (strong scaling: the
frame is fixed, value = Gaussians of the frame / step time).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n", "--gaussians", dest="n", type=int, default=1_000_000,
                    help="(--n is an ambiguous prefix for torch.distributed.run's own parser: use --gaussians there)")
    ap.add_argument("--width", type=int, default=1200)
    ap.add_argument("--height", type=int, default=680)
    ap.add_argument("--sigma-px", type=float, default=1.0)
    ap.add_argument("--order", choices=["random", "raster"], default="random")
    ap.add_argument("--pose-only", action="store_true", help="skip per-Gaussian gradient outputs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-div", type=int, default=3,
                    help="fallback cpu_baseline (torch oracle) sample: N/div^2 Gaussians on a (W/div)x(H/div) image")
    ap.add_argument("--no-tracker", action="store_true", help="skip the pose-opt iterations/s side measurement")
    ap.add_argument("--no-variants", action="store_true", help="skip the side measurements of the other workload variants")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of one HIP graph per step")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="dev: run the N-rank path with every rank on cuda:0 and a gloo (host) all-reduce")
    return ap.parse_args()


def trace(msg):
    """GSLOC_BENCH_TRACE=1: synchronise and print a phase marker (to localise an asynchronous GPU fault)."""
    if os.environ.get("GSLOC_BENCH_TRACE"):
        torch.cuda.synchronize()
        print(f"[trace rank {os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)


def algorithmic_bytes(N, I, P, D, n_tiles, full):
    """BASELINE.md section 3 byte model, per stage (bytes per step)."""
    import math
    p = math.ceil((32 + math.ceil(math.log2(max(n_tiles, 2)))) / 8)
    g = 1 if full else 0
    return {
        "project_fwd": N * 68,
        "bin": N * 16 + I * 12 + I * 24 * p + I * 8 + n_tiles * 4,
        "raster_fwd": I * (28 + 4 * D) + P * (4 * D + 8),
        "raster_bwd": I * (28 + 4 * D) + P * (4 * D + 12) + 2 * I * (24 + 4 * D),
        "project_bwd": N * (64 + 4 * D) + N * 40 * g,
    }


STAGE_KERNELS = {  # stage timer -> kernels it brackets (names as rocprofv3 prints them, template arguments dropped)
    "project_fwd": ("k_fproject<",), "bin": ("k_ftile_scan", "k_fscatter", "k_tile_sort"),
    "raster_fwd": ("k_praster_fwd", "k_fraster_fwd"), "raster_bwd": ("k_mraster_bwd", "k_fraster_bwd", "k_praster_bwd", "k_sraster_bwd", "k_tiny_bwd", "k_tiny_gather"),
    "project_bwd": ("k_fproject_bwd", "k_freduce_viewmat"),
}


def pmc_traffic(stage):
    """HBM bytes per launch of the stage's kernels from the committed PMC passes of this same command
    (profiles/r01_pmc_traffic.json, made by scripts/pmc_summary.py from separate --pmc FETCH_SIZE /
    WRITE_SIZE runs; counters cannot be read from inside the process).  None when the file is absent."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if not os.path.exists(path):
        return None
    ks = json.load(open(path))["kernels"]
    tot = [v["hbm_bytes"] for k, v in ks.items() if any(k.startswith(p) for p in STAGE_KERNELS.get(stage, ()))]
    return sum(tot) if tot else None


def cpu_baseline(args):
    """The CPU oracle (the 'port') timed on the host cores: oracle/csrc/gsplat_oracle.c built for float32 (the
    arithmetic type of the HIP path) with OpenMP on min(host cores, 16) threads -- a 1-GPU box exposes a 16-core
    share -- on the FULL workload (same N, image, sigma_px, order, gradient outputs), one warm-up step and the best
    of two timed steps (~10 s of CPU work).  If the C library cannot be built, the PyTorch oracle on a
    same-density subsample (N/div^2 on W/div x H/div) is timed instead and says so."""
    from gsplatloc_amd.synthetic import perturbed_pose, random_scene

    cores = min(os.cpu_count() or 1, 16)
    V = torch.linalg.inv(perturbed_pose())
    try:
        from oracle import c_oracle
        c_oracle.load("f32")
    except Exception as exc:  # no compiler on the box and no prebuilt library: fall back to the torch oracle
        return cpu_baseline_torch(args, cores, V, reason=f"{type(exc).__name__}: {exc}")
    n, w, h = args.n, args.width, args.height
    sc = random_scene(n, w, h, sigma_px=args.sigma_px, order=args.order)
    g = torch.Generator().manual_seed(1)
    v = torch.zeros(h, w, 4)
    v[..., 3] = torch.randn(h, w, generator=g)
    arrays = [sc[k].numpy() for k in ("means", "quats", "scales", "opacities", "sh")] + [V.numpy(), sc["K"].numpy()]
    v = v.numpy()

    def step():
        return c_oracle.rasterization(*arrays, w, h, sh_degree=1, render_mode="RGB+ED", v_render=v, precision="f32",
                                      threads=cores)

    step()
    ts = []
    for _ in range(2):
        t = time.perf_counter()
        out = step()
        ts.append(time.perf_counter() - t)
    dt = min(ts)
    return {"value": n / dt, "unit": "Gaussians/s", "cores": cores, "kind": "port",
            "sample": f"oracle/csrc/gsplat_oracle.c float32 + OpenMP, the full workload (N={n}, {w}x{h}, "
                      f"{out['n_isects']} intersections), fwd+bwd with all gradients, best of 2 after one warm-up "
                      f"step, {dt:.2f} s/step"}


def cpu_baseline_torch(args, cores, V, reason):
    from oracle import gsplat_oracle as G
    from gsplatloc_amd.synthetic import random_scene

    torch.set_num_threads(cores)
    div = max(1, args.cpu_sample_div)
    n, w, h = args.n // (div * div), args.width // div, args.height // div
    sc = random_scene(n, w, h, sigma_px=args.sigma_px)
    g = torch.Generator().manual_seed(1)
    vd = torch.randn(1, h, w, generator=g)

    def step():
        ins = [sc[k].clone().requires_grad_() for k in ("means", "quats", "scales", "opacities", "sh")]
        Vg = V[None].clone().requires_grad_()
        rc, ra, _ = G.rasterization(*ins, Vg, sc["K"][None], w, h, sh_degree=1, render_mode="RGB+ED")
        (rc[..., 3] * vd).sum().backward()

    step()  # warm-up at the same size (the first pass pays for allocator growth: ~3x slower)
    ts = []
    for _ in range(2):
        t = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t)
    dt = min(ts)
    return {"value": n / dt, "unit": "Gaussians/s", "cores": cores, "kind": "port",
            "sample": f"oracle/gsplat_oracle.py fp32 (C oracle unavailable: {reason}), N={n} on {w}x{h} "
                      f"(N/{div * div}, W/{div} x H/{div} of the workload, same splat density), fwd+bwd, best of 2 "
                      f"after one warm-up step, {dt:.2f} s/step"}


def variant_rate(dev, N, W, H, sigma_px, order, steps=15, warmup=3):
    """Side measurement: the same step on another variant of the workload (graph replay, all gradients)."""
    from gsplatloc_amd.context import RenderContext
    from gsplatloc_amd.synthetic import perturbed_pose, random_scene

    sc = random_scene(N, W, H, sigma_px=sigma_px, device=dev, order=order)
    viewmat = torch.linalg.inv(perturbed_pose()).to(dev).contiguous()
    K = sc["K"].contiguous()
    ctx = RenderContext(N, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=dev, full_grads=True)
    inp = (sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["sh"], viewmat, K)
    n_is = ctx.calibrate(*inp)
    g = torch.Generator().manual_seed(1)
    v = torch.zeros(H, W, 4)
    v[..., 3] = torch.randn(H, W, generator=g)
    v = v.to(dev)
    va = torch.zeros(H, W, 1, device=dev)

    def step():
        ctx.forward(*inp)
        ctx.backward(v, va, full=True)

    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        step()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            step()
    torch.cuda.synchronize()
    for _ in range(warmup):
        graph.replay()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        graph.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / steps
    ctx.check_capacity()
    return {"sigma_px": sigma_px, "order": order, "intersections_per_gaussian": n_is / N, "ms_per_step": dt * 1e3,
            "gaussians_per_s": N / dt, "backward": backward_name(ctx)}


def backward_name(ctx):
    if getattr(ctx, "slab", 0):
        return f"per-pixel walk + {ctx.slab}x{ctx.slab} LDS slabs"
    import os
    kind = os.environ.get("GSLOC_RASTER_BWD", "mfma")
    return "tiny-splat slabs (global)" if getattr(ctx, "tiny", False) else {"mfma": "quadrant walk + MFMA pixel sums", "quad": "quadrant walk + wave reduce-scatter", "px": "per-pixel two-phase"}[kind]


def pose_opt_rate(dev):
    """BASELINE metric 2 (side measurement, rank 0, N=1): pose-optimisation iterations/s of the whole
    tracker iteration (render fwd+bwd, depth+edge loss, pose chain, 2x Adam, LR decay, early-stop
    bookkeeping) as one HIP graph, config S of BASELINE.json: ~100k Gaussians, 640x480, 200 iterations."""
    import gsplatloc_amd.my_gsplat as M
    from gsplatloc_amd.graph_tracker import GraphTracker
    from gsplatloc_amd.my_gsplat.geometry import depth_to_points
    from gsplatloc_amd.synthetic import frame_pair

    W, H, iters = 640, 480, 200
    fp = frame_pair(W, H, rot_deg=0.3, trans=0.01)
    K = fp["K"].to(dev)
    pts0 = depth_to_points(fp["depth0"].to(dev), K)[::3].contiguous()
    rgb = fp["rgb"].to(dev)[::3].contiguous()
    pts1 = depth_to_points(fp["depth1"].to(dev), K)
    scales = M.init_gs_scales(pts0)
    src = M.compute_depth_gt(pts1, fp["rgb"].to(dev), K[None], torch.eye(4, device=dev)[None], H, W)
    cfg = M.TrackerConfig(max_steps=iters, min_step=100, patience=10 ** 9)
    gt = GraphTracker(pts0.shape[0], W, H, cfg, device=dev, poll=50)
    frame = (pts0, rgb, scales, src, fp["c2w0"].to(dev), fp["c2w1"].to(dev), K)
    gt.load_frame(*frame)
    gt.run()
    gt.load_frame(*frame)
    torch.cuda.synchronize()
    t = time.perf_counter()
    res = gt.run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    return {"config": "S: depth-map-like frame pair, %d Gaussians, 640x480, 200 iterations, HIP graph" % pts0.shape[0],
            "iters_per_s": res.steps / dt, "ms_per_iter": dt / res.steps * 1e3, "loss_first": res.losses[0],
            "loss_last": res.losses[-1], "eT_init_m": 0.01, "best_eT_m": res.best_eT}


def guarded(fn, *a, **k):
    """Side measurements must not cost the headline line: report their failure instead of raising."""
    try:
        return fn(*a, **k)
    except Exception as exc:  # noqa: BLE001 - anything: the JSON line still has to be printed
        import traceback
        traceback.print_exc(file=sys.stderr)
        return {"error": f"{type(exc).__name__}: {exc}"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"

    from gsplatloc_amd.context import RenderContext
    from gsplatloc_amd.parallel import strip_rows
    from gsplatloc_amd.synthetic import perturbed_pose, random_scene

    N, W, H = args.n, args.width, args.height
    sc = random_scene(N, W, H, sigma_px=args.sigma_px, device=dev, order=args.order)
    viewmat = torch.linalg.inv(perturbed_pose()).to(dev).contiguous()
    K = sc["K"].contiguous()
    th = (H + 15) // 16
    full = not args.pose_only

    # ---- tile-row strip of this rank (balanced on a calibration pass) -------------------------
    rows = (0, th)
    n_local = N
    if world > 1:
        # once per frame (untimed): balance strips on a full binning pass, keep the Gaussians that can reach
        # this rank's strip (1-tile guard band), the rest never touch its pixels
        from gsplatloc_amd.parallel import gaussians_for_strip
        cal = RenderContext(N, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=dev, full_grads=False)
        cal.calibrate(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["sh"], viewmat, K)
        rows = strip_rows(cal.offs, cal.tw, cal.th, world)[rank]
        idx = gaussians_for_strip(cal.Q0[:, 0:2], cal.radii, rows)
        for k in ("means", "quats", "scales", "opacities", "sh"):
            sc[k] = sc[k][idx].contiguous()
        n_local = int(idx.numel())
        del cal
        trace(f"strip rows {rows}, {n_local} local Gaussians")
    ctx = RenderContext(n_local, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=dev, tile_rows=rows, full_grads=full)
    n_isects = ctx.calibrate(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["sh"], viewmat, K)
    trace(f"calibrated, {n_isects} intersections")
    g = torch.Generator().manual_seed(1)
    v_render = torch.zeros(H, W, 4)
    v_render[..., 3] = torch.randn(H, W, generator=g)
    v_render = v_render.to(dev)
    v_alphas = torch.zeros(H, W, 1, device=dev)
    pose_grad = torch.zeros(16, device=dev)
    args_in = (sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["sh"], viewmat, K)

    def step():
        ctx.forward(*args_in)
        grads = ctx.backward(v_render, v_alphas, full=full)
        if dist is not None:
            if args.rehearse_on_one_gpu:
                host = grads["viewmat"].reshape(16).cpu()
                dist.all_reduce(host)
                pose_grad.copy_(host)
            else:
                pose_grad.copy_(grads["viewmat"].reshape(16))
                dist.all_reduce(pose_grad)  # THE collective of the path: 12 pose-gradient entries (+4 spare)

    graph = None
    side = torch.cuda.Stream()
    if not args.no_graph and dist is None:
        # one iteration = a fixed sequence of 10 launches: replay it as a single HIP graph.  (N > 1 launches
        # eagerly: a per-rank graph followed by the collective faulted in the one-GPU rehearsal)
        with torch.cuda.stream(side):
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            trace("eager steps done")
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                step()
        torch.cuda.synchronize()
        trace("graph captured")
    run = graph.replay if graph is not None else step

    for _ in range(args.warmup):
        run()
        trace("warm-up step")
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device="cpu" if args.rehearse_on_one_gpu else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    n_total = ctx.check_capacity()
    ms = dt / args.steps * 1e3
    trace("timed steps done")

    # ---- per-kernel durations: HIP events on the launch stream, same steps, second pass -------
    stage_ms = {}
    if rank == 0:
        from gsplatloc_amd import context as C
        stage_ms = guarded(C.time_stages, ctx, args_in, v_render, v_alphas, full, steps=min(args.steps, 20))
        if "error" in stage_ms:
            stage_ms = {}

    if rank == 0:
        P = W * H
        bytes_stage = algorithmic_bytes(n_local, n_total, P if world == 1 else (rows[1] - rows[0]) * 16 * W, 4,
                                        ctx.n_tiles, full)
        dom = max(stage_ms, key=stage_ms.get) if stage_ms else "raster_bwd"
        dom_ms = stage_ms.get(dom)  # None if the stage timing pass failed
        achieved = bytes_stage[dom] / (dom_ms * 1e-3) / 1e9 if dom_ms else None
        out = {
            "metric": "Gaussians/s fwd+bwd @1M splats 1200x680",
            "value": N / (dt / args.steps),
            "unit": "Gaussians/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms,
            "higher_is_better": True,
            "scaling": "strong",  # the frame (N Gaussians, one image) is fixed; more GPUs split its tile rows
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"R: {N} random Gaussians ({args.order} order), {W}x{H}, sigma_px={args.sigma_px}, "
                            f"render_mode=RGB+ED sh_degree=1, backward from a depth-channel gradient, "
                            f"{'full per-Gaussian gradients + pose gradient' if full else 'pose gradient only'}",
                "intersections_per_gaussian": (n_total / N) if world == 1 else None,
                "strip_intersections_rank0": n_total, "tile_rows_rank0": list(rows), "gaussians_rank0": n_local,
                "parallelism": "single GPU" if world == 1 else f"{world} screen-tile strips + 1 all-reduce(16 f32)",
                "launch": "hipGraph replay" if graph is not None else "eager",
                "backward": backward_name(ctx),
            },
            "roofline": {
                "bound": "hbm", "kernel": dom, "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                "frac": (achieved / 8000.0) if achieved else None,
                "traffic": pmc_traffic(dom) if (world == 1 and args.sigma_px == 1.0 and args.order == "random" and N == 1_000_000) else None,
                "algorithmic_bytes_per_launch": bytes_stage[dom], "avg_launch_ms": dom_ms,
                "whole_step": {"algorithmic_bytes": sum(bytes_stage.values()),
                               "achieved_GBps": sum(bytes_stage.values()) / (ms * 1e-3) / 1e9,
                               "frac": sum(bytes_stage.values()) / (ms * 1e-3) / 1e9 / 8000.0},
                "stage_ms": stage_ms,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = guarded(cpu_baseline, args)
        else:
            out["cpu_baseline"] = None
        if world == 1 and not args.no_tracker:
            out["pose_opt"] = guarded(pose_opt_rate, dev)
        if world == 1 and not args.no_variants:
            # same N and image, the other synthetic inputs of SURVEY.md 8(d): sigma_px -> 0 is the regime of the
            # reference's as-coded kNN scales; "raster" is the Gaussian order of a back-projected depth frame
            out["variants"] = [guarded(variant_rate, dev, N, W, H, s_, o_) for s_, o_ in
                               ((1.0, "raster"), (0.0, "random"), (0.0, "raster"))]
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
