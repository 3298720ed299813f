"""bench.py's driver logic on the CPU (tests/bench_fake.py swaps the GPU pieces for host stand-ins): the contract of
the JSON line at N = 1, and the N = 2 path launched exactly as the harness launches it (torch.distributed.run, here
over gloo) -- strips, the per-step collective, barrier + max over ranks, rank 0 prints one line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "bench_fake.py")
SMALL = ["--gaussians", "20000", "--width", "320", "--height", "240", "--steps", "3", "--warmup", "1", "--no-graph",
         "--no-cpu-baseline", "--no-tracker", "--no-variants"]

REQUIRED = {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int, "warmup": int, "ms_per_step": float,
            "higher_is_better": bool, "scaling": str, "dtype": str, "data": str, "config": dict, "roofline": dict}


def _check(line, n_gpus):
    d = json.loads(line)
    for key, typ in REQUIRED.items():
        assert isinstance(d[key], typ), (key, d[key])
    assert "vs_baseline" in d and d["vs_baseline"] is None and "cpu_baseline" in d
    assert d["n_gpus"] == n_gpus and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "Gaussians/s" and d["scaling"] == "strong" and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert set(r) >= {"achieved", "frac", "traffic", "stage_ms"} and set(r["stage_ms"]) == {
        "project_fwd", "bin", "raster_fwd", "raster_bwd", "project_bwd"}
    assert d["value"] == pytest.approx(20000 / (d["ms_per_step"] * 1e-3), rel=1e-6)
    return d


def test_single_process_line_follows_the_contract():
    res = subprocess.run([sys.executable, FAKE] + SMALL, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = _check(lines[0], 1)
    assert d["config"]["parallelism"] == "single GPU" and d["config"]["launch"] == "eager"


def test_two_ranks_as_the_harness_launches_them():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29533", FAKE, "--gpus", "2"] + SMALL
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]  # rank 0 only
    d = _check(lines[0], 2)
    assert d["cpu_baseline"] is None and "2 screen-tile strips" in d["config"]["parallelism"]
    assert d["config"]["strip_lists_match_full_frame_rank0"] is True
    rows = d["config"]["tile_rows_rank0"]
    assert rows[0] == 0 and 0 < rows[1] < 15  # rank 0 owns the top strip of the 15 tile rows


def test_failing_side_measurements_do_not_cost_the_headline_line():
    """Without a GPU the tracker and variant side measurements must fail -- and be reported as errors inside the
    line, next to a real cpu_baseline (the C oracle runs on the host)."""
    args = ["--gaussians", "20000", "--width", "320", "--height", "240", "--steps", "3", "--warmup", "1", "--no-graph"]
    res = subprocess.run([sys.executable, FAKE] + args, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = _check(lines[0], 1)
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "Gaussians/s" and cb["value"] > 0 and cb["cores"] >= 1
    assert "gsplat_oracle.c" in cb["sample"] and "N=20000" in cb["sample"]
    assert "error" in d["pose_opt"] and all("error" in v for v in d["variants"]) and len(d["variants"]) == 5
    assert "error" in d["api"] and "error" in d["frame"]  # (the boundary and per-frame side measurements need the GPU too)
