#!/usr/bin/env bash
# Round 4: whole GPU suite, then frame / api objects of the bench.
set -o pipefail
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.jsonl
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/t_all.log 2>&1
rc=$?; tail -6 gpurun_out/t_all.log
if [ $rc -ne 0 ]; then grep -a "^E  \|^FAILED" gpurun_out/t_all.log | head -30; fi
if grep -aq "Memory access fault" gpurun_out/t_all.log; then echo "GPU FAULT in the test run"; exit 1; fi
timeout -k 10 600 python bench.py --no-cpu-baseline --no-variants > gpurun_out/r04_bench_side.json 2> gpurun_out/r04_bench_side.err; rc2=$?
if [ $rc2 -ne 0 ]; then tail -20 gpurun_out/r04_bench_side.err; exit $rc2; fi
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r04_bench_side.json") if l.startswith("{")][-1])
print(d["ms_per_step"])
for k in ("pose_opt", "api", "frame"):
    print(k, json.dumps(d.get(k))[:900])
PY
exit $rc
