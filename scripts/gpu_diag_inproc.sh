#!/usr/bin/env bash
# One GPU call: the full default bench (every leg) under a rocprofv3 kernel trace; where do the variants' steps go?
set -o pipefail
mkdir -p gpurun_out
R=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$R"
timeout -k 10 600 rocprofv3 --kernel-trace -d gpurun_out/kt_full -o kt --output-format csv -- python3 bench.py > gpurun_out/kt_full.log 2>&1
rc=$?; if [ $rc -ge 124 ]; then exit $rc; fi
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/kt_full.log") if l.startswith("{")][-1])
print(d["ms_per_step"], d["step_ms_hip_events"])
for v in d["variants"]: print(v)
PY
python3 scripts/trace_tail.py $(find gpurun_out/kt_full -name "*kernel_trace.csv" | head -1)
