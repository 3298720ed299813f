#!/usr/bin/env bash
# Round evidence: SQ counters of the compositing kernels (eager bench run) and the backward's trip statistics (a
# -DGSL_G16_STATS build made on the box, then the normal build restored).
set -o pipefail
mkdir -p gpurun_out
bash scripts/gpu_pmc.sh | grep -a "k_\|pmc rc" | cut -c1-400 | tee gpurun_out/pmc_sq_summary.txt
touch gsplatloc_amd/csrc/raster_g16.hip; make -C gsplatloc_amd/csrc EXTRA=-DGSL_G16_STATS 2>&1 | grep -E "error" -A5
python3 scripts/g16_stats.py 1.0 2>/dev/null | tee gpurun_out/g16_stats.txt
python3 scripts/g16_stats.py 0.0 2>/dev/null | tee -a gpurun_out/g16_stats.txt
