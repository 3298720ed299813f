#!/usr/bin/env bash
# Copy the evidence of the last scripts/gpu_final.sh run from gpurun_out/ (scratch) into profiles/ (tracked).
set -e
R=${1:-r04}
for w in R D S T X; do python3 - "$w" "$R" <<'PY'
import sys
w, r = sys.argv[1], sys.argv[2]
l = [x for x in open(f"gpurun_out/bench_{w}.json") if x.startswith("{")][-1]
open(f"profiles/{r}_bench_{w}.json", "w").write(l)
PY
done
cp gpurun_out/prof_R/st_kernel_stats.csv profiles/${R}_bench_R_1M_1200x680_kernel_stats.csv
cp gpurun_out/prof_D/st_kernel_stats.csv profiles/${R}_bench_D_816k_depth_frame_kernel_stats.csv
cp gpurun_out/prof_X/st_kernel_stats.csv profiles/${R}_bench_X_5M_1920x1080_fp16_kernel_stats.csv
cp gpurun_out/prof_trk/st_kernel_stats.csv profiles/${R}_tracker_S_102k_kernel_stats.csv
cp gpurun_out/pmc_traffic.json profiles/${R}_pmc_traffic.json
python3 scripts/issue_model.py profiles/${R}_issue_model.json
cp gpurun_out/parity_report.jsonl profiles/${R}_parity_report.jsonl
python3 scripts/parity_table.py profiles/${R}_parity_report.jsonl
