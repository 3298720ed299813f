#!/usr/bin/env bash
# Round 4, last step: the default bench line with the committed PMC summary / issue model in place (their source hashes
# match the build), tracker iteration rates, the one-GPU estimate of the strip scaling curve.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python bench.py > gpurun_out/bench_R_final.json 2> gpurun_out/bench_R_final.err || { tail -5 gpurun_out/bench_R_final.err; exit 1; }
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/bench_R_final.json") if l.startswith("{")][-1])
r = d["roofline"]
print(d["ms_per_step"], d["value"], "frac", r["frac"], "traffic", r["traffic"], "issue frac", r.get("issue", {}).get("frac"), "cpu", d["cpu_baseline"]["value"])
PY
bash scripts/gpu_r4_trk.sh
timeout -k 10 300 python scripts/trk_render_mode.py 2>/dev/null | tee gpurun_out/r04_tracker_render_mode.txt
( timeout -k 10 300 python scripts/strip_scaling.py 1.0 random R; timeout -k 10 400 python scripts/strip_scaling.py 1.0 random X ) > gpurun_out/r04_strip_scaling_estimate.txt 2>&1
grep "^world" gpurun_out/r04_strip_scaling_estimate.txt
