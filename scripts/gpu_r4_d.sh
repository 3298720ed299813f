#!/usr/bin/env bash
# Round 4: third ubench batch + the default bench run (api / frame / parity objects) + GPU tests touched so far.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 scripts/ubench/valu.bin > gpurun_out/r04_valu_issue_3.txt 2>&1
grep "SIMD=8" gpurun_out/r04_valu_issue_3.txt | tail -16 | cut -c1-84
timeout -k 10 900 python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err; rc=$?
if [ $rc -ne 0 ]; then tail -20 gpurun_out/r04_bench_default.err; exit $rc; fi
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r04_bench_default.json") if l.startswith("{")][-1])
print(d["ms_per_step"], d["value"], d["step_ms_hip_events"])
r = d["roofline"]
print("roofline", {k: r[k] for k in ("kernel", "achieved", "frac")}, "whole", r["whole_step"]["frac"], "survey", r["whole_step_survey_model"]["frac"])
print("stages", {k: (round(v["ms"], 4), round(v["frac"], 3), round(v["model_frac"], 3), "FLAG" if "flag" in v else "") for k, v in r["stages"].items()})
print("issue", r.get("issue"))
for k in ("parity", "cpu_baseline", "pose_opt", "api", "frame"):
    print(k, json.dumps(d.get(k))[:1200])
for v in d.get("variants") or []:
    print("variant", {k: v.get(k) for k in ("workload", "sigma_px", "order", "ms_per_step", "placement", "error")})
PY
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_properties.py tests/test_gpu_perf_guard.py tests/test_gpu_eval.py -q -s > gpurun_out/t_cfg.log 2>&1; rc=$?
tail -4 gpurun_out/t_cfg.log; grep -a "parity\]" gpurun_out/t_cfg.log | cut -c1-420
if [ $rc -ne 0 ]; then grep -a "^E  " gpurun_out/t_cfg.log | head -20; fi
exit $rc
