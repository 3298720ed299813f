#!/usr/bin/env python3
"""Headline benchmark: Gaussians/s, forward + backward, of the GsplatLoc render hot path.

One "step" = one full pass of the path BASELINE.json names on a batch of synthetic input:
projection + SH colour -> tile binning + per-tile depth sort -> alpha compositing ("RGB+ED",
the reference's call, /root/reference/src/my_gsplat/model.py:195-213) -> backward from a given
depth-channel gradient to every Gaussian input and to the 4x4 view matrix.  Inputs are resident
in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload R|S|T|X|D]

Workloads (SURVEY.md 8d / BASELINE.json configs): R (default, the metric's configuration) = 1 M random Gaussians,
1200x680, sigma_px = 1; X = 5 M random Gaussians, 1920x1080; S = a 640x480 depth frame back-projected to 102 400
Gaussians with the reference's as-coded kNN scales; T = one Gaussian per pixel of a 640x480 depth frame with
invalid (zero) depths, 307 200 Gaussians; D = one Gaussian per pixel of a 1200x680 depth frame, 816 000 Gaussians, as-coded
scales (the reference's own regime at R size).

N > 1 (launched by torch.distributed.run, one rank per GPU over RCCL): screen-tile rows are
split across ranks, every rank renders and back-propagates its strip, and the 12+4 pose-gradient floats are
summed with ONE all-reduce per step (strong scaling: the frame is fixed, value = Gaussians of the frame / step
time).  Each rank replays one HIP graph per step (library launches only) and then issues the collective.
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

# Idle OpenMP workers (torch's CPU ops, the CPU oracle of the cpu_baseline leg) must sleep, not spin: on a GPU box the
# job owns a CPU *share* (a cgroup quota) of a much larger host, and a few hundred spinning workers burn the quota of a
# scheduler period in a few milliseconds -- the whole process, the thread that launches the HIP graphs included, is
# then frozen until the next period (~85 ms stalls between graph replays: DESIGN.md section 6).
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
os.environ.setdefault("KMP_BLOCKTIME", "0")

import torch  # noqa: E402

WORKLOADS = {
    "R": dict(n=1_000_000, width=1200, height=680, sigma_px=1.0, order="random", kind="random"),
    "X": dict(n=5_000_000, width=1920, height=1080, sigma_px=1.0, order="random", kind="random"),
    "S": dict(n=102_400, width=640, height=480, kind="depth_frame", stride=3, holes=False),
    "T": dict(n=307_200, width=640, height=480, kind="depth_frame", stride=1, holes=True),
    # the reference's own regime at R size: one Gaussian per pixel of a 1200x680 Replica depth frame, raster order,
    # as-coded kNN^2 scales (sigma_px -> 0): ref/src/data/Image.py:29,35, my_gsplat/geometry.py:60-64, cam_params.json:3-4
    "D": dict(n=816_000, width=1200, height=680, kind="depth_frame", stride=1, holes=False),
}
PMC_SUMMARY = os.path.join("profiles", "r04_pmc_traffic.json")   # FETCH_SIZE / WRITE_SIZE / SQ counters (scripts/pmc_summary.py)
ISSUE_MODEL = os.path.join("profiles", "r04_issue_model.json")   # static VALU class mix of the hot loops (scripts/issue_model.py)
# measured issue cost of one wave-instruction on one SIMD, ns (profiles/r04_valu_issue.txt): 2-cycle class (fma / add / mul
# / mov / logic on VGPRs), 4-cycle class (anything with an SGPR operand, v_cndmask, v_cmp, DPP, min / max, packed f32,
# integer bit ops), transcendentals
ISSUE_NS = {"fast": 1.04, "slow": 1.80, "trans": 3.40}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="R")
    ap.add_argument("--n", "--gaussians", dest="n", type=int, default=None,
                    help="(--n is an ambiguous prefix for torch.distributed.run's own parser: use --gaussians there)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--sigma-px", type=float, default=None)
    ap.add_argument("--order", choices=["random", "raster"], default=None)
    ap.add_argument("--staging", choices=["fp32", "fp16"], default=None,
                    help="record precision of the compositing kernels (default: fp16 for workload X, BASELINE.json "
                         "configs[4] 'fp16 compositing'; fp32 otherwise)")
    ap.add_argument("--pose-only", action="store_true", help="skip per-Gaussian gradient outputs")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU baseline and the parity block")
    ap.add_argument("--cpu-sample-div", type=int, default=3,
                    help="fallback cpu_baseline (torch oracle) sample: N/div^2 Gaussians on a (W/div)x(H/div) image")
    ap.add_argument("--no-tracker", action="store_true", help="skip the pose-opt iterations/s side measurement")
    ap.add_argument("--no-variants", action="store_true", help="skip the side measurements of the other workload variants")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of one HIP graph per step")
    ap.add_argument("--collective-with-one-rank", action="store_true",
                    help="dev/test: with --gpus 1, still create an RCCL group of one rank and issue the per-step all-reduce "
                         "(exercises the captured-collective path on a one-GPU box)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="dev: run the N-rank path with every rank on cuda:0 and a gloo (host) all-reduce")
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]
    explicit = any(getattr(args, k) is not None for k in ("n", "width", "height", "sigma_px", "order"))
    for key in ("n", "width", "height", "sigma_px", "order"):
        if getattr(args, key) is None:
            setattr(args, key, wl.get(key, {"sigma_px": 0.0, "order": "raster"}.get(key)))
    # explicit sizes always mean the random-N scene; the depth-frame workloads are selected by name only
    args.kind = "random" if (explicit or wl["kind"] == "random") else "depth_frame"
    if args.staging is None:
        args.staging = "fp16" if (args.workload == "X" and not explicit) else "fp32"
    return args


def host_cpu_share():
    """CPUs this process may really use: the affinity mask, cut down to the cgroup's CPU quota when there is one
    (cgroup v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return n


def throttle_stats():
    """(nr_throttled, throttled ms) of this process's cgroup, or None: how often the scheduler froze the job for
    having used up its CPU quota."""
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            kv = dict(line.split()[:2] for line in open(path) if len(line.split()) >= 2)
            us = int(kv.get("throttled_usec", 0)) or int(kv.get("throttled_time", 0)) // 1000
            return int(kv.get("nr_throttled", 0)), us / 1e3
        except (OSError, ValueError):
            continue
    return None


def trace(msg):
    """GSLOC_BENCH_TRACE=1: synchronise and print a phase marker (to localise an asynchronous GPU fault)."""
    if os.environ.get("GSLOC_BENCH_TRACE"):
        torch.cuda.synchronize()
        print(f"[trace rank {os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)


def algorithmic_bytes(N, I, P, D, n_tiles, full):
    """SURVEY.md 8(d) / BASELINE.md section 3 byte model, per stage (bytes per step)."""
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


def implementation_bin_bytes(N, I, n_tiles):
    """What THIS build's binning moves (no credit for radix passes it does not run): the projection kernel writes
    one 8-byte key per intersection into its tile's bin; the per-tile sort reads the keys and writes the 4-byte
    Gaussian ids; tile sizes and offsets are 8 bytes per tile.  (The two-pass path of the drop-in API also re-reads
    a 16-byte record and a radius per Gaussian in its scatter pass: N * 20 more.)"""
    return I * (8 + 8 + 4) + n_tiles * 8


STAGE_KERNELS = {  # stage timer -> kernels it brackets (names as rocprofv3 prints them, template arguments dropped)
    "project_fwd": ("k_fproject<true, true>", "k_fproject<false, true>"), "bin": ("k_tile_sort",),
    "raster_fwd": ("k_praster_fwd", "k_long_fwd", "k_long_combine", "k_long_map"),
    "raster_bwd": ("k_qraster_bwd", "k_mraster_bwd", "k_tiny_bwd"),
    "project_bwd": ("k_fproject_bwd", "k_freduce_viewmat"),
}


def pmc_traffic(stage):
    """HBM bytes per launch of the stage's kernels from the committed PMC passes of this same command (made by
    scripts/pmc_summary.py from separate --pmc FETCH_SIZE / WRITE_SIZE runs; counters cannot be read from inside
    the process).  None when the file is absent."""
    path = os.path.join(ROOT, PMC_SUMMARY)
    if not os.path.exists(path):
        return None
    summary = json.load(open(path))
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    try:
        from pmc_summary import csrc_sha
        if summary.get("csrc_sha") != csrc_sha():
            return None  # the counters were collected on other kernel sources: not this build's traffic
    except Exception:  # noqa: BLE001
        return None
    ks = summary["kernels"]
    tot = [v["hbm_bytes"] for k, v in ks.items() if any(k.startswith(p) for p in STAGE_KERNELS.get(stage, ()))]
    return sum(tot) if tot else None


def pmc_kernels():
    """Per-kernel counter summary of PMC_SUMMARY while its kernel-source hash matches this build, else None."""
    path = os.path.join(ROOT, PMC_SUMMARY)
    if not os.path.exists(path):
        return None
    summary = json.load(open(path))
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    try:
        from pmc_summary import csrc_sha
        if summary.get("csrc_sha") != csrc_sha():
            return None
    except Exception:  # noqa: BLE001
        return None
    return summary["kernels"]


def issue_bound(stage, launch_ms):
    """The bound the compositing kernels actually run against: VALU issue.  `frac` = the measured
    occupancy of the vector ALUs during the launch (hash-matched SQ-counter summary).  Next to it the instruction counts
    and what they would cost if every 2-cycle instruction found a partner (the static class mix of the kernel's hot loop,
    scripts/issue_model.py, priced with the issue costs of profiles/r04_valu_issue.txt): the gap between the two is why
    removing 4-cycle instructions paid half of what that model promised."""
    ks = pmc_kernels()
    if not ks:
        return None
    names = [k for k in ks if any(k.startswith(p) for p in STAGE_KERNELS.get(stage, ())) and "sq" in ks[k]]
    if not names:
        return None
    k = max(names, key=lambda n: ks[n]["sq"].get("SQ_INSTS_VALU", 0))
    sq = ks[k]["sq"]
    valu = sq.get("SQ_INSTS_VALU")
    mix = None
    try:
        model = json.load(open(os.path.join(ROOT, ISSUE_MODEL)))
        from pmc_summary import csrc_sha
        if model.get("csrc_sha") == csrc_sha():
            mix = next((v for n, v in model["kernels"].items() if k.startswith(n)), None)
    except Exception:  # noqa: BLE001
        mix = None
    simds = 1024.0
    out = {"kernel": k, "valu_insts": valu, "salu_insts": sq.get("SQ_INSTS_SALU"), "lds_insts": sq.get("SQ_INSTS_LDS"),
           "lds_array_busy_frac": (sq["SQ_LDS_IDX_ACTIVE"] / 256.0) / sq["SQ_BUSY_CYCLES_per_se"]
           if sq.get("SQ_LDS_IDX_ACTIVE") and sq.get("SQ_BUSY_CYCLES_per_se") else None,
           "issue_us_if_all_2_cycle": valu * ISSUE_NS["fast"] / simds * 1e-3 if valu else None,
           "issue_us_if_all_4_cycle": valu * ISSUE_NS["slow"] / simds * 1e-3 if valu else None,
           "issue_ns_per_class": ISSUE_NS, "source": f"{PMC_SUMMARY} + {ISSUE_MODEL} (both hash-matched to this build)"}
    if valu and mix:
        tot = float(mix["fast"] + mix["slow"] + mix["trans"])
        ns = (mix["fast"] * ISSUE_NS["fast"] + mix["slow"] * ISSUE_NS["slow"] + mix["trans"] * ISSUE_NS["trans"]) / tot
        out["hot_loop_class_mix"] = {c: mix[c] / tot for c in ("fast", "slow", "trans")}
        out["issue_us_if_every_2_cycle_op_paired"] = valu * ns / simds * 1e-3
        out["frac_if_every_2_cycle_op_paired"] = out["issue_us_if_every_2_cycle_op_paired"] / (launch_ms * 1e3) if launch_ms else None
    # THE measurement: cycles in which a SIMD's vector ALU was occupied by an instruction (SQ_ACTIVE_INST_VALU, counted
    # in units of 4 cycles like SQ_WAVE_CYCLES) over the SIMD-cycles of the launch (1024 SIMDs x the launch's cycles,
    # SQ_BUSY_CYCLES per shader engine), from the same counter run: in the mixed instruction stream of a compositing
    # trip the 2-cycle instructions do NOT pair up -- the kernels average 4.1 cycles per VALU instruction
    if sq.get("SQ_ACTIVE_INST_VALU") and sq.get("SQ_BUSY_CYCLES_per_se") and valu:
        out["valu_busy_frac"] = sq["SQ_ACTIVE_INST_VALU"] * 4.0 / (simds * sq["SQ_BUSY_CYCLES_per_se"])
        out["cycles_per_valu_inst"] = sq["SQ_ACTIVE_INST_VALU"] * 4.0 / valu
        # (two counters of different blocks: the ratio is good to a few per cent -- the forward of round 4's last build
        # reads 1.04 -- so `frac` saturates at 1 and the raw ratio stays next to it)
        out["frac"] = min(1.0, out["valu_busy_frac"])
    return out


def build_scene(args, dev):
    """The synthetic input of the workload (seeded) and the camera it is rendered from."""
    from gsplatloc_amd.synthetic import perturbed_pose, random_scene

    if args.kind == "depth_frame":
        from gsplatloc_amd.synthetic import depth_frame_scene
        wl = WORKLOADS[args.workload]
        return depth_frame_scene(args.width, args.height, stride=wl["stride"], holes=wl["holes"], device=dev)
    sc = random_scene(args.n, args.width, args.height, sigma_px=args.sigma_px, device=dev, order=args.order)
    sc["viewmat"] = torch.linalg.inv(perturbed_pose()).to(dev).contiguous()
    return sc


def cpu_baseline(args, scene, gpu):
    """The CPU oracle (the 'port') timed on the host cores: oracle/csrc/gsplat_oracle.c built for float32 (the
    arithmetic type of the HIP path) with OpenMP on every host core, on the FULL workload (same inputs, same
    gradient outputs), one warm-up step and the best of two timed steps.  The same leg checks the GPU result against
    the oracle's float64 build (the `parity` object): images on all pixels, the pose gradient flip-aware (upstream
    gradient zeroed where the two renders disagree beyond the image tolerance).  If the C library cannot be built,
    the PyTorch oracle on a same-density subsample is timed instead and says so."""
    cores = host_cpu_share()  # not os.cpu_count(): the box gives the job a CPU share of a much larger host
    V = scene["viewmat"].cpu()
    try:
        from oracle import c_oracle
        c_oracle.load("f32")
    except Exception as exc:  # no compiler on the box and no prebuilt library: fall back to the torch oracle
        return cpu_baseline_torch(args, cores, V, reason=f"{type(exc).__name__}: {exc}"), None
    n, w, h = scene["means"].shape[0], args.width, args.height
    arrays = [scene[k].cpu().numpy() for k in ("means", "quats", "scales", "opacities", "sh")] + [V.numpy(), scene["K"].cpu().numpy()]
    v = gpu["v_render"].cpu().numpy()

    def step(precision="f32", v_render=v, threads=min(cores, 64)):
        return c_oracle.rasterization(*arrays, w, h, sh_degree=1, render_mode="RGB+ED", v_render=v_render,
                                      precision=precision, threads=threads)

    # the port's OpenMP loops do not scale to every core of a large host: time a few thread counts within the job's CPU
    # share (one warm-up and two timed steps each) and report the fastest, with the thread count it used
    best = None
    for threads in sorted({min(cores, 64), min(cores, 16)}, reverse=True):
        step(threads=threads)
        for _ in range(2):
            t = time.perf_counter()
            out = step(threads=threads)
            d = time.perf_counter() - t
            if best is None or d < best[0]:
                best = (d, threads)
    dt, used = best
    base = {"value": n / dt, "unit": "Gaussians/s", "cores": used, "kind": "port",
            "sample": f"oracle/csrc/gsplat_oracle.c float32 + OpenMP, fastest of {{64, 16}} threads within the job's CPU share "
                      f"({cores} of the host's {os.cpu_count()} cores) = {used} threads, the full workload (N={n}, {w}x{h}, {out['n_isects']} intersections), "
                      f"fwd+bwd with all gradients, best of 2 after one warm-up step, {dt:.2f} s/step"}
    parity = None
    if gpu.get("render") is not None:
        import numpy as np
        ref = step("f64", None)
        rg, ag = gpu["render"].cpu().double().numpy(), gpu["alphas"].cpu().double().numpy()[..., 0]
        d_rel = np.abs(rg[..., 3] - ref["render"][..., 3]) / np.maximum(np.abs(ref["render"][..., 3]), 1e-3)
        rt, at = (1e-4, 2e-5) if args.staging == "fp32" else (3e-3, 3e-4)
        ok = (np.abs(rg - ref["render"]) <= at + rt * np.abs(ref["render"])).all(-1) & (
            np.abs(ag - ref["alphas"]) <= at + rt * np.abs(ref["alphas"]))
        # pose gradient over three upstream gradients (white noise on the depth channel, seeds 1..3; the first is the
        # timed step's): HIP against the float64 oracle (max over the seeds), HIP against the oracle's own float32 build
        # (like for like) and that build against float64 (the float32 floor of this configuration).  The bound applied is
        # the tests' (tests/parity.py): 1e-4, or min(2 x floor, 8e-4) where the floor itself is above 5e-5.
        from gsplatloc_amd.synthetic import depth_upstream
        worst = vs32 = floor32 = det = 0.0
        for seed in (1, 2, 3):
            vk = v if seed == 1 else depth_upstream(h, w, seed).numpy()  # (the tests' generator, dtype and seeds)
            vm = vk * ok[..., None]
            want, want32 = step("f64", vm), step("f32", vm)
            got = gpu["backward"](torch.from_numpy(vm))
            gv = got.cpu().double().numpy()[:3]
            if gpu.get("backward_deterministic") is not None:  # the same sums in a fixed order (no float atomics)
                gd = gpu["backward_deterministic"](torch.from_numpy(vm)).cpu().double().numpy()[:3]
                det = max(det, float(np.abs(gd - want["v_viewmat"][:3]).max() / np.abs(want["v_viewmat"][:3]).max()))
            w64, w32 = want["v_viewmat"][:3], want32["v_viewmat"][:3].astype(np.float64)
            worst = max(worst, float(np.abs(gv - w64).max() / np.abs(w64).max()))
            vs32 = max(vs32, float(np.abs(gv - w32).max() / np.abs(w32).max()))
            floor32 = max(floor32, float(np.abs(w32 - w64).max() / np.abs(w64).max()))
        from tests.parity import POSE_GRAD_CAPS, pose_grad_bound
        kind = "X" if n > 1_100_000 else ("sigma1" if (args.kind == "random" and args.sigma_px >= 0.5) else "subpixel")
        bound = pose_grad_bound(floor32, kind) if args.staging == "fp32" else None
        parity = {"against": "oracle/csrc/gsplat_oracle.c float64",
                  "tolerance": ("images: 1e-4 relative + 2e-5 absolute (north_star's 1e-4); pose gradient: "
                                f"{bound:.1e} of its largest entry (tests/parity.py POSE_GRAD_CAPS[{kind!r}]: a fixed cap per kind of "
                                "configuration, set from the measured table), flip-aware; north_star's 1e-4 is met by the "
                                "images and by small scenes, NOT by the whole-frame pose gradient at this size")
                               if args.staging == "fp32" else
                               "fp16-staged records: ~1e-3 relative expected (half has 11 significant bits)",
                  "depth_rel_err": {"mean": float(d_rel.mean()), "p99": float(np.quantile(d_rel.reshape(-1)[::7], 0.99)),
                                    "max_over_agreeing_pixels": float(d_rel[ok].max())},
                  "alpha_abs_err": {"mean": float(np.abs(ag - ref["alphas"]).mean()),
                                    "max_over_agreeing_pixels": float(np.abs(ag - ref["alphas"])[ok].max())},
                  "pixels_beyond_tolerance": float(1.0 - ok.mean()),
                  "v_viewmat_rel_err": worst, "v_viewmat_seeds": 3,
                  "v_viewmat_vs_f32_oracle": vs32, "v_viewmat_f32_oracle_vs_f64": floor32,
                  "v_viewmat_deterministic_rel_err": det if gpu.get("backward_deterministic") is not None else None,
                  "v_viewmat_meets_north_star_1e-4": worst < 1e-4,
                  "v_viewmat_within_bound": (worst < bound) if bound is not None else None,
                  "v_viewmat_note": "max over 3 white-noise upstream gradients (a sum of ~1e6 terms of random sign); the "
                                    "oracle's float32 build differs from its float64 build by v_viewmat_f32_oracle_vs_f64",
                  "n_isects": [int(gpu["n_isects"]), int(ref["n_isects"])]}
    return base, parity


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


def backward_name(ctx):
    if getattr(ctx, "tiny", False):
        return "tiny-splat slabs, folded inside the projection backward"
    if getattr(ctx, "deterministic", False):
        return "quadrant walk, per-splat pixel sums by v_mfma_f32_16x16x4_f32, one gradient row per intersection (deterministic)"
    return ("one wave per 8x8 quadrant, each 16-lane DPP row walks its 4x4 block's entries of the quadrant's hit LIST "
            "(written by the forward), DPP reduce-scatter, packed 64-byte atomics")


def placement_name(ctx):
    if getattr(ctx, "order_ids", None) is not None:
        return ("tile order: the context stores its own copies of the Gaussians sorted by the tile of the projected centre "
                "(once per frame, at calibrate(); the sort key keeps the caller's index, lists bit-identical up to relabelling)")
    return "as given by the caller"


def event_stats(ms):
    """median / p10 / p90 of the per-step durations measured with HIP events on the launch stream."""
    if not ms:
        return None
    s = sorted(ms)
    q = lambda f: s[min(len(s) - 1, int(round(f * (len(s) - 1))))]  # noqa: E731
    return {"median": q(0.5), "p10": q(0.1), "p90": q(0.9), "n": len(s)}


def variant_rate(dev, N, W, H, sigma_px, order, steps=100, warmup=5, depth_frame=False, reorder=None):
    """Side measurement: the same step on another variant of the workload (graph replay, all gradients).  Per-step HIP
    events, median over `steps` replays (a wall-clock mean over a handful of replays cannot tell one host stall from a
    slow kernel); the wall-clock mean is reported next to it.  depth_frame: workload D (one Gaussian per pixel of a
    W x H depth frame, as-coded scales) instead of the random-N scene."""
    from gsplatloc_amd.context import RenderContext
    from gsplatloc_amd.synthetic import depth_frame_scene, perturbed_pose, random_scene

    if depth_frame:
        sc = depth_frame_scene(W, H, stride=3 if depth_frame == "S" else 1, holes=False, device=dev)
        viewmat, N = sc["viewmat"], sc["N"]
    else:
        sc = random_scene(N, W, H, sigma_px=sigma_px, device=dev, order=order)
        viewmat = torch.linalg.inv(perturbed_pose()).to(dev).contiguous()
    K = sc["K"].contiguous()
    ctx = RenderContext(N, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=dev, full_grads=True, reorder=reorder)
    inp = (sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["sh"], viewmat, K)
    n_is = ctx.calibrate(*inp)
    from gsplatloc_amd.synthetic import depth_upstream
    v = depth_upstream(H, W, 1).to(dev)
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
    t = time.perf_counter()
    for _ in range(warmup):
        graph.replay()
    torch.cuda.synchronize()
    warm_wall = (time.perf_counter() - t) / max(warmup, 1)
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t = time.perf_counter()
    for e0, e1 in events:
        e0.record()
        graph.replay()
        e1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t) / steps
    ctx.check_capacity()
    ev = event_stats([a.elapsed_time(b) for a, b in events])
    ev["max"] = max(a.elapsed_time(b) for a, b in events)
    dt = ev["median"] * 1e-3
    return {"workload": f"D: {N} Gaussians of a {W}x{H} depth frame, as-coded scales" if depth_frame else "random-N",
            "sigma_px": None if depth_frame else sigma_px, "order": "raster" if depth_frame else order,
            "intersections_per_gaussian": n_is / N, "ms_per_step": ev["median"],
            "step_ms_hip_events": ev, "ms_per_step_wall_mean": wall * 1e3, "warmup_ms_per_step_wall_mean": warm_wall * 1e3,
            "gaussians_per_s": N / dt,
            "backward": backward_name(ctx), "placement": placement_name(ctx)}


def api_rate(dev, context_R_ms=None):
    """The drop-in boundary, timed as the reference uses it (/root/reference/src/my_gsplat/model.py:195-213 inside the loop
    of gs_trainer_total.py:79-267): wall time per  `from gsplat import rasterization`  call with the reference's keyword
    arguments + `.backward()` from the depth channel, allocation of the outputs and the status read-back included, pose
    gradient only (what the tracker consumes) and with the per-Gaussian gradients the reference's parameters would also
    receive; best of three windows after a warm-up; at S (102 400 Gaussians of a 640x480 depth frame) and R.  Beside it
    the same step on a RenderContext as one HIP graph."""
    from gsplat import rasterization
    from gsplatloc_amd.fused import clear_context_cache
    from gsplatloc_amd.synthetic import depth_frame_scene, perturbed_pose, random_scene

    def api(sc, c2w, W, H, full, n, as_called=True, backward=True, warm=300):
        V0 = torch.linalg.inv(c2w).contiguous()
        m = sc["means"].clone().requires_grad_(full)

        def step():
            if as_called:  # the reference's expression: viewmats = torch.linalg.inv(camtoworlds), differentiated through
                c2w_g = c2w.clone().requires_grad_()
                viewmats = torch.linalg.inv(c2w_g)[None]
            else:          # the boundary alone: the view matrix is the leaf
                viewmats = V0.clone().requires_grad_(backward)[None]
            with torch.set_grad_enabled(backward):
                render_colors, render_alphas, info = rasterization(
                    means=m, quats=sc["quats"], scales=sc["scales"], opacities=sc["opacities"], colors=sc["sh"],
                    sh_degree=1, viewmats=viewmats, Ks=sc["K"][None], width=W, height=H, packed=False, absgrad=False,
                    sparse_grad=False, far_plane=1e10, near_plane=1e-2, render_mode="RGB+ED", rasterize_mode="classic")
                assert render_colors.shape[-1] == 4
                if backward:
                    (render_colors[..., 3:4] * 0.5).sum().backward()
                    m.grad = None
        for _ in range(warm):  # (a cold process reads 0.41 ms at S where the loop of a tracker, hundreds of calls per frame,
            step()             # runs at 0.26: host clocks and caches -- warm up as long as a frame's optimisation lasts)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                step()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t) / n * 1e3)
        return best

    out = {"call": "gsplat.rasterization(**kw of ref model.py:195-213) + backward from the depth channel, wall ms per call; "
                   "'boundary': the view matrix is the autograd leaf (the call and its backward alone); 'as the reference "
                   "calls it': torch.linalg.inv(camtoworlds) in front, differentiated through (model.py:202)"}
    for name in ("S", "R"):
        if name == "S":
            W, H = 640, 480
            sc = depth_frame_scene(W, H, stride=3, device=dev)
            c2w = torch.linalg.inv(sc["viewmat"])
        else:
            W, H = 1200, 680
            sc = random_scene(1_000_000, W, H, device=dev)
            c2w = perturbed_pose().to(dev)
        sc = dict(sc, K=sc["K"].contiguous())
        n = 400 if name == "S" else 100
        api(sc, c2w, W, H, False, 30, warm=30)  # (first use: context creation and calibration)
        w = 300 if name == "S" else 60
        out[name] = {"as_the_reference_calls_it_ms": api(sc, c2w, W, H, False, n, warm=w),  # inv(c2w) + call + backward to c2w
                     "boundary_pose_gradient_only_ms": api(sc, c2w, W, H, False, n, as_called=False, warm=w),
                     "boundary_with_gaussian_gradients_ms": api(sc, c2w, W, H, True, n, as_called=False, warm=w),
                     "boundary_forward_only_no_grad_ms": api(sc, c2w, W, H, False, n, as_called=False, backward=False,
                                                             warm=w)}
        if name == "S":
            v = variant_rate(dev, sc["means"].shape[0], W, H, 0.0, "raster", steps=100, depth_frame="S")
            out[name]["render_context_graph_ms"] = v["ms_per_step"]
        elif context_R_ms is not None:
            out[name]["render_context_graph_ms"] = context_R_ms
        clear_context_cache()
    return out


def frame_rate(dev, n_frames=9):
    """Per-frame end-to-end rate of the reference's evaluation protocol (python -m gsplatloc_amd.eval: GT-initialised,
    <= 2000 iterations, early stop; /root/reference/src/my_gsplat/gs_trainer_total.py:45-282, dataset.py:345-383) on a synthetic
    Replica-format 640x480 sequence written to a temporary directory, with the wall time of each phase of a frame."""
    import pathlib
    import shutil
    import tempfile

    from gsplatloc_amd.data.dataset import Parser
    from gsplatloc_amd.eval import evaluate_room
    from gsplatloc_amd.synthetic import write_replica_sequence

    root = pathlib.Path(tempfile.mkdtemp())
    try:
        write_replica_sequence(root, 640, 480, n_frames + 1)
        parser = Parser("Replica", "room0", normalize=True, input_folder=str(root))
        evaluate_room(parser, num_iters=2000, max_frames=2, verbose=False)  # warm-up: library, allocator, first captures
        res = evaluate_room(parser, num_iters=2000, max_frames=None, verbose=False, profile=True)
    finally:
        shutil.rmtree(root, ignore_errors=True)
    ph = res.get("phase_seconds") or {}
    tot = sum(ph.values()) or 1.0
    return {"protocol": "gsplatloc_amd.eval.evaluate_room, synthetic Replica-format sequence 640x480, normalize=True",
            "frames": res["frames"], "frames_per_s": res["frames_per_s"], "mean_iterations_per_frame": res["mean_steps"],
            "ATE_m": res["ATE"], "AAE_deg": res["AAE"],
            "phase_ms_per_frame": {k: v / res["frames"] * 1e3 for k, v in ph.items()},
            "setup_fraction_of_frame": 1.0 - ph.get("optimise", 0.0) / tot}


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
    frame = (pts0, rgb, scales, src, fp["c2w0"].to(dev), fp["c2w1"].to(dev), K)

    def rate(mode):
        gt = GraphTracker(pts0.shape[0], W, H, cfg, device=dev, poll=50, render_mode=mode)
        gt.load_frame(*frame)
        gt.run()
        gt.load_frame(*frame)
        torch.cuda.synchronize()
        t = time.perf_counter()
        res = gt.run()
        torch.cuda.synchronize()
        return res, time.perf_counter() - t

    # the loop that renders expected depth only (what gsplatloc_amd.eval runs: the reference's loss reads no other channel)
    res_ed, dt_ed = rate("ED")
    res, dt = rate("RGB+ED")  # the reference's literal call: the figure of rounds 1-3
    return {"config": "S: depth-map-like frame pair, %d Gaussians, 640x480, 200 iterations, HIP graph" % pts0.shape[0],
            "iters_per_s": res.steps / dt, "ms_per_iter": dt / res.steps * 1e3, "loss_first": res.losses[0],
            "loss_last": res.losses[-1], "eT_init_m": 0.01, "best_eT_m": res.best_eT,
            "render_mode": "RGB+ED (the reference's literal call; RGB is rendered and discarded)",
            "depth_only": {"render_mode": "ED (what gsplatloc_amd.eval runs: same depth, same loss)",
                           "iters_per_s": res_ed.steps / dt_ed, "ms_per_iter": dt_ed / res_ed.steps * 1e3,
                           "loss_last": res_ed.losses[-1], "best_eT_m": res_ed.best_eT}}


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
    torch.set_num_threads(min(host_cpu_share(), 16))  # host-side torch ops are tiny here; never one thread per host core
    throttle0 = throttle_stats()
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.collective_with_one_rank:
        import torch.distributed as dist_mod
        dist = dist_mod
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            if "RANK" not in os.environ:  # started without torch.distributed.run (one rank)
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29655")
                dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
            else:
                dist.init_process_group("nccl", device_id=dev)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"

    from gsplatloc_amd import context as C
    from gsplatloc_amd.parallel import strip_rows

    scene = build_scene(args, dev)
    N, W, H = scene["means"].shape[0], args.width, args.height
    viewmat = scene["viewmat"]
    K = scene["K"].contiguous()
    th = (H + 15) // 16
    full = not args.pose_only
    sc = {k: scene[k] for k in ("means", "quats", "scales", "opacities", "sh")}

    # ---- tile-row strip of this rank (balanced on a calibration pass) -------------------------
    rows = (0, th)
    n_local = N
    if world > 1:
        # once per frame (untimed): balance strips on a full binning pass, keep the Gaussians that can reach
        # this rank's strip (1-tile guard band), the rest never touch its pixels
        from gsplatloc_amd.parallel import gaussians_for_strip
        # (reorder=False: cal.Q0 / cal.radii index the Gaussians below, so they must be in the CALLER's order -- a placed
        # calibration context silently selects the wrong Gaussians; the intersection count of the strip is asserted further down)
        cal = C.RenderContext(N, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=dev, full_grads=False, reorder=False)
        cal.calibrate(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["sh"], viewmat, K)
        rows = strip_rows(cal.offs, cal.tw, cal.th, world)[rank]
        strip_isects_expected = int(cal.offs[rows[1] * cal.tw] - cal.offs[rows[0] * cal.tw])
        idx = cal.gaussians_reaching(rows)
        # the guard band is checked, not assumed (GraphTracker does the same at every poll, where the pose
        # moves): no Gaussian outside the kept set may reach the strip at the pose the timed steps render
        kept = torch.zeros(N, dtype=torch.bool, device=dev)
        kept[idx] = True
        reach = torch.zeros(N, dtype=torch.bool, device=dev)
        reach[cal.gaussians_reaching(rows, guard_tiles=0)] = True
        band_violations = int((reach & ~kept).sum())
        if band_violations:  # (reported in the line rather than raised: a scaling run should still produce its number)
            print(f"[bench] rank {rank}: {band_violations} Gaussians outside the kept set reach strip {rows}", file=sys.stderr)
        for k in sc:
            sc[k] = sc[k][idx].contiguous()
        n_local = int(idx.numel())
        del cal
        trace(f"strip rows {rows}, {n_local} local Gaussians")
    ctx = C.RenderContext(n_local, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=dev, tile_rows=rows, full_grads=full,
                          staging=args.staging)
    n_isects = ctx.calibrate(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["sh"], viewmat, K)
    # what a frame pays ONCE, outside the timed steps (a tracker runs hundreds of steps per frame): the measuring pass that
    # sizes lists and bins, the tile-order placement (a stable sort of N tile indices + five gathers) and the allocations;
    # timed on a second call, when the allocator is warm
    torch.cuda.synchronize()
    t_cal = time.perf_counter()
    n_isects = ctx.calibrate(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["sh"], viewmat, K)
    torch.cuda.synchronize()
    calibrate_ms = (time.perf_counter() - t_cal) * 1e3
    trace(f"calibrated, {n_isects} intersections")
    strip_ok = True
    if world > 1:  # the strip's lists must hold exactly the full frame's entries of its tile rows
        strip_ok = (n_isects == strip_isects_expected) and band_violations == 0
        if not strip_ok:
            print(f"[bench] rank {rank}: strip {rows} has {n_isects} intersections, the full frame {strip_isects_expected} "
                  f"for the same tile rows; {band_violations} guard-band violations", file=sys.stderr)
    from gsplatloc_amd.synthetic import depth_upstream
    v_render = depth_upstream(H, W, 1).to(dev)  # (seed 1 of the parity statement's three: tests/test_gpu_configs.py)
    v_alphas = torch.zeros(H, W, 1, device=dev)
    pose_grad = torch.zeros(16, device=dev)  # this rank's 16 floats of the all-reduce
    host16 = None
    if dist is not None and args.rehearse_on_one_gpu:
        host16 = torch.zeros(16).pin_memory()
    args_in = (sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["sh"], viewmat, K)

    def render_step():
        """The rank's launches of one step: library kernels and memsets only (captured as one HIP graph)."""
        ctx.forward(*args_in)
        grads = ctx.backward(v_render, v_alphas, full=full)
        if dist is not None:  # pack the 12 pose-gradient entries (+4 spare) into the all-reduce buffer
            C.pack_pose_reduce(grads["viewmat"], pose_grad)

    def collective():
        """THE collective of the path: one all-reduce of 16 floats (RCCL on the device buffer; the one-GPU
        rehearsal's gloo group goes through a pinned host buffer)."""
        if host16 is not None:
            host16.copy_(pose_grad, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            dist.all_reduce(host16)
            pose_grad.copy_(host16, non_blocking=True)
        else:
            dist.all_reduce(pose_grad)

    graph = None
    collective_in_graph = False
    side = torch.cuda.Stream()
    if not args.no_graph:
        with torch.cuda.stream(side):
            for _ in range(2):
                render_step()
                if dist is not None:
                    collective()  # (first use creates the communicator: outside any capture)
            torch.cuda.synchronize()
            trace("eager steps done")
            # Over RCCL the all-reduce is captured inside the step's graph (one replay per step, no host dispatch between
            # the render and the collective); the gloo rehearsal keeps it eager.  GSLOC_CAPTURE_COLLECTIVE=0: eager.
            if dist is not None and host16 is None and os.environ.get("GSLOC_CAPTURE_COLLECTIVE", "1") != "0":
                try:
                    g_all = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g_all, stream=side, capture_error_mode="thread_local"):
                        render_step()
                        collective()
                    graph, collective_in_graph = g_all, True
                except Exception as exc:  # noqa: BLE001 - refused: two dispatches per step, as before
                    print(f"[bench] all-reduce not captured ({type(exc).__name__}: {exc}); eager collective", file=sys.stderr)
                    torch.cuda.synchronize()
            if graph is None:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    render_step()
        torch.cuda.synchronize()
        trace("graph captured")
    render = graph.replay if graph is not None else render_step

    def run():
        render()
        if dist is not None and not collective_in_graph:
            collective()

    for _ in range(args.warmup):
        run()
        trace("warm-up step")
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for e0, e1 in events:
        e0.record()
        run()
        e1.record()
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
    step_ms = guarded(lambda: event_stats([a.elapsed_time(b) for a, b in events]))
    trace("timed steps done")

    # ---- per-kernel durations: HIP events on the launch stream, same steps, second pass -------
    stage_ms = {}
    if rank == 0:
        stage_ms = guarded(C.time_stages, ctx, args_in, v_render, v_alphas, full, steps=min(args.steps, 20))
        if "error" in stage_ms:
            stage_ms = {}

    if rank == 0:
        P = W * H
        P_local = P if world == 1 else (min(rows[1] * 16, H) - rows[0] * 16) * W
        bytes_stage = algorithmic_bytes(n_local, n_total, P_local, 4, ctx.n_tiles, full)
        impl_bytes = dict(bytes_stage, bin=implementation_bin_bytes(n_local, n_total, ctx.n_tiles))
        dom = max(stage_ms, key=stage_ms.get) if stage_ms else "raster_bwd"
        dom_ms = stage_ms.get(dom)  # None if the stage timing pass failed
        achieved = bytes_stage[dom] / (dom_ms * 1e-3) / 1e9 if dom_ms else None
        headline = args.workload == "R" and args.kind == "random" and args.sigma_px == 1.0 and args.order == "random"
        if args.kind == "random":
            what = f"{args.workload}: {N} random Gaussians ({args.order} order), {W}x{H}, sigma_px={args.sigma_px}"
        else:
            what = (f"{args.workload}: {N} Gaussians back-projected from a {W}x{H} depth frame (raster order, as-coded "
                    f"kNN scales{', invalid depths' if WORKLOADS[args.workload]['holes'] else ''})")
        out = {
            "metric": "Gaussians/s fwd+bwd @1M splats 1200x680" if (headline and N == 1_000_000) else
                      f"Gaussians/s fwd+bwd @{N} splats {W}x{H} (workload {args.workload})",
            "value": N / (dt / args.steps),
            "unit": "Gaussians/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms,
            "step_ms_hip_events": step_ms,
            "higher_is_better": True,
            "scaling": "strong",  # the frame (N Gaussians, one image) is fixed; more GPUs split its tile rows
            "vs_baseline": None,
            "dtype": "f32" if args.staging == "fp32" else "f32 (f16-staged records)",
            "data": "synthetic",
            "config": {
                "workload": what + ", render_mode=RGB+ED sh_degree=1, backward from a depth-channel gradient, "
                            + ("full per-Gaussian gradients + pose gradient" if full else "pose gradient only"),
                "intersections_per_gaussian": (n_total / N) if world == 1 else None,
                "strip_intersections_rank0": n_total, "tile_rows_rank0": list(rows), "gaussians_rank0": n_local,
                "parallelism": "single GPU" if world == 1 else f"{world} screen-tile strips + 1 all-reduce(16 f32)",
                "guard_band": None if world == 1 else "Gaussians pruned per strip with a 1-tile guard band; checked at the "
                                                      "rendered pose against a projection of all N (rank 0)",
                "strip_lists_match_full_frame_rank0": None if world == 1 else bool(strip_ok),
                "launch": ("hipGraph replay" + (", all-reduce captured in the graph" if collective_in_graph else
                                                 (", eager all-reduce after the replay" if dist is not None else "")))
                          if graph is not None else "eager",
                "backward": backward_name(ctx),
                "gaussian_placement": placement_name(ctx),
                "per_frame_calibrate_ms": calibrate_ms,  # (once per frame, NOT in ms_per_step: see the comment at its timing)
                "binning": ("keys written into per-tile bins by the projection kernel (sizes from calibrate()), "
                            "register sort per tile" if getattr(ctx, "bins", None) is not None
                            else "count, scan, scatter, register sort per tile"),
                "staging": args.staging + (" records for compositing (T and accumulators f32)" if args.staging == "fp16" else ""),
            },
            "roofline": {
                "bound": "hbm", "kernel": dom, "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                "frac": (achieved / 8000.0) if achieved else None,
                "traffic": pmc_traffic(dom) if (world == 1 and headline and N == 1_000_000) else None,
                "traffic_source": f"rocprofv3 --pmc summary {PMC_SUMMARY} of this command, quoted only while its kernel-source "
                                  "hash matches this build (counters cannot be read from inside the process)",
                "algorithmic_bytes_per_launch": bytes_stage[dom], "avg_launch_ms": dom_ms,
                # per stage: the SURVEY.md 8(d) bytes AND what this build moves; a stage whose model bytes would need more
                # than the chip's peak in the time measured is flagged -- the kernel is not doing the model's work (the bin
                # stage: the model prices six radix passes, the build sorts in registers), and its `frac` is the
                # implementation figure
                "stages": {
                    s_: {"ms": stage_ms[s_], "model_bytes": bytes_stage[s_], "implementation_bytes": impl_bytes[s_],
                         "model_frac": bytes_stage[s_] / (stage_ms[s_] * 1e-3) / 8e12,
                         "frac": impl_bytes[s_] / (stage_ms[s_] * 1e-3) / 8e12,
                         **({"flag": "model bytes / time exceeds the HBM peak: the kernel does not move the model's bytes; "
                                     "frac counts what it moves"}
                            if bytes_stage[s_] / (stage_ms[s_] * 1e-3) / 8e12 > 1.0 else {})}
                    for s_ in stage_ms if stage_ms.get(s_)},
                "whole_step": {"model": "SURVEY.md 8(d) bytes, the bin stage counted as what this build moves (keys written "
                                        "once by the projection, sorted in registers, ids written): no credit for radix "
                                        "passes it does not run",
                               "algorithmic_bytes": sum(impl_bytes.values()),
                               "achieved_GBps": sum(impl_bytes.values()) / (ms * 1e-3) / 1e9,
                               "frac": sum(impl_bytes.values()) / (ms * 1e-3) / 1e9 / 8000.0},
                "whole_step_survey_model": {
                    "note": "SURVEY.md 8(d) as written, six radix passes in the bin stage included; quoted for continuity "
                            "with rounds 1-3, not a statement about this build's traffic",
                    "algorithmic_bytes": sum(bytes_stage.values()),
                    "frac": sum(bytes_stage.values()) / (ms * 1e-3) / 1e9 / 8000.0},
                # what the compositing kernels are actually bound by
                "issue": issue_bound(dom, dom_ms) if (world == 1 and headline and N == 1_000_000) else None,
                "issue_other_compositing_stage": issue_bound("raster_fwd" if dom == "raster_bwd" else "raster_bwd",
                                                             stage_ms.get("raster_fwd" if dom == "raster_bwd" else "raster_bwd"))
                if (world == 1 and headline and N == 1_000_000) else None,
                "stage_ms": stage_ms,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            def gpu_backward(vm):
                ctx.forward(*args_in)
                grads = ctx.backward(vm.float().to(dev).contiguous(), v_alphas, full=full)
                torch.cuda.synchronize()
                return grads["viewmat"].clone()

            det_ctx = []

            def gpu_backward_det(vm):  # same sums, fixed order (deterministic=True): built on first use, fp32 records only
                if not det_ctx:
                    dc = C.RenderContext(n_local, W, H, "RGB+ED", sh_degree=1, K_sh=4, device=dev, full_grads=False,
                                         deterministic=True, reorder=False)
                    dc.calibrate(*args_in)
                    det_ctx.append(dc)
                dc = det_ctx[0]
                dc.forward(*args_in)
                grads = dc.backward(vm.float().to(dev).contiguous(), v_alphas, full=False)
                torch.cuda.synchronize()
                return grads["viewmat"].clone()

            def baseline_and_parity():
                ctx.forward(*args_in)
                torch.cuda.synchronize()
                gpu = {"render": getattr(ctx, "render", None), "alphas": getattr(ctx, "alphas", None),
                       "v_render": v_render, "backward": gpu_backward, "n_isects": n_total,
                       "backward_deterministic": gpu_backward_det if (args.staging == "fp32" and N <= 1_100_000
                                                                      and hasattr(ctx, "render")) else None}
                if gpu["render"] is not None:
                    gpu["render"], gpu["alphas"] = gpu["render"].clone(), gpu["alphas"].clone()
                return cpu_baseline(args, dict(scene, K=K), gpu)

            res = guarded(baseline_and_parity)
            if isinstance(res, tuple):
                out["cpu_baseline"], out["parity"] = res
            else:
                out["cpu_baseline"] = res
        else:
            out["cpu_baseline"] = None
        if world == 1 and not args.no_tracker:
            out["pose_opt"] = guarded(pose_opt_rate, dev)
            # the drop-in boundary and the per-frame end-to-end figure
            out["api"] = guarded(api_rate, dev, ms if (headline and N == 1_000_000) else None)
            out["frame"] = guarded(frame_rate, dev)
        if world == 1 and not args.no_variants:
            # same N and image, the other synthetic inputs of SURVEY.md 8(d): sigma_px -> 0 is the regime of the
            # reference's as-coded kNN scales; "raster" is the Gaussian order of a back-projected depth frame
            out["variants"] = [guarded(variant_rate, dev, args.n, W, H, s_, o_) for s_, o_ in
                               ((1.0, "raster"), (0.0, "random"), (0.0, "raster"))]
            # the headline input WITHOUT the tile-order placement (Gaussians gathered where the caller put them)
            out["variants"].append(guarded(variant_rate, dev, args.n, W, H, 1.0, "random", reorder=False))
            out["variants"].append(guarded(variant_rate, dev, W * H, W, H, 0.0, "raster", depth_frame=True))
        th1 = throttle_stats()
        try:  # BENCH_r02 recorded one process still alive when the command returned: name whatever this process started
            import psutil
            kids = [f"{c.pid}:{c.name()}" for c in psutil.Process().children(recursive=True)]
        except Exception:  # noqa: BLE001
            kids = None
        out["host"] = {"cpu_share": host_cpu_share(), "host_cores": os.cpu_count(), "child_processes_at_end": kids,
                       "cgroup_throttled": None if (throttle0 is None or th1 is None) else
                       {"periods": th1[0] - throttle0[0], "ms": th1[1] - throttle0[1]}}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
