#!/usr/bin/env bash
# The full default bench line (every leg): do the variants still see host stalls?  (cgroup throttle counters in "host")
set -o pipefail
mkdir -p gpurun_out
cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc
show() { python3 - "$1" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("   main", round(d["ms_per_step"], 4), d["step_ms_hip_events"], "host", d.get("host"))
print("   cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "parity", d.get("parity", {}).get("v_viewmat_rel_err"))
print("   pose_opt", d.get("pose_opt", {}).get("iters_per_s"))
for v in d.get("variants", []):
    if "error" in v: print("   ", v); continue
    print("   ", v.get("sigma_px"), v.get("order"), "median", round(v["ms_per_step"], 4), "wall", round(v["ms_per_step_wall_mean"], 4), "warm wall", round(v["warmup_ms_per_step_wall_mean"], 3), "max", round(v["step_ms_hip_events"]["max"], 3))
PY
}
for i in 1 2; do
  echo "== bench.py (run $i)"
  timeout -k 10 400 python3 bench.py > gpurun_out/legs$i.json 2> gpurun_out/legs$i.err; rc=$?
  if [ $rc -ge 124 ]; then exit $rc; fi
  tail -2 gpurun_out/legs$i.err
  show gpurun_out/legs$i.json
done
echo "== bench.py --workload D"
timeout -k 10 400 python3 bench.py --workload D --no-variants > gpurun_out/legsD.json 2> gpurun_out/legsD.err; rc=$?
if [ $rc -ge 124 ]; then exit $rc; fi
tail -2 gpurun_out/legsD.err; show gpurun_out/legsD.json
