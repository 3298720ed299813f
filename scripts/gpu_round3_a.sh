#!/usr/bin/env bash
# Round-3 measurements (a): strip-scaling estimate, pile, tracker iteration rates.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python3 scripts/strip_scaling.py 1.0 random 2>/dev/null | tee gpurun_out/strips.log
timeout -k 10 200 python3 scripts/pile_bench.py 2>/dev/null | tee gpurun_out/pile.log
for c in S T R; do timeout -k 10 400 python3 scripts/bench_tracker.py $c graph 2>/dev/null | tee -a gpurun_out/tracker_rates.jsonl | cut -c1-400; done
