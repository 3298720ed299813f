#!/usr/bin/env bash
# Round 4: tracker iterations/s at S / T / R-size depth frames, tiny against general backward.
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r04_tracker_iterations.jsonl
for cfg in S T R; do
  for bwd in auto general; do
    GSLOC_BWD=$bwd timeout -k 10 300 python scripts/bench_tracker.py $cfg graph 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); g=d['graph']
print('$cfg $bwd', d['N'], round(g['iters_per_s']), 'it/s', round(g['ms_per_iter'],4), 'ms', 'best_eT', g['best_eT'])
d['backward']='$bwd'; open('gpurun_out/r04_tracker_iterations.jsonl','a').write(json.dumps(d)+'\n')
"
  done
done
