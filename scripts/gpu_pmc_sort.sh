#!/usr/bin/env bash
# Dev: what is k_tile_sort waiting for?  Three counter passes at R (eager launches).
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="--no-cpu-baseline --no-tracker --no-variants --no-graph --steps 3 --warmup 1"
pass() { timeout -k 10 300 rocprofv3 --pmc $2 --kernel-trace -d gpurun_out/pmc_$1 -o $1 --output-format csv -- python3 bench.py $B > gpurun_out/pmc_$1.log 2>&1; echo "pass $1 rc=$?"; }
pass s1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"
pass s2 "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
pass s3 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum GRBM_GUI_ACTIVE"
python3 - <<'PY'
import csv, glob, collections
for d in ("pmc_s1", "pmc_s2", "pmc_s3"):
    rows = []
    for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gsl::", "")[:34]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen:
            seen.add((k, r["Dispatch_Id"])); cnt[k] += 1
    for k in agg:
        if k.startswith(("k_tile_sort", "k_fproject<true, true", "k_praster", "k_qraster_bwd<4, true, 1")):
            print(f"{k:34s} n={cnt[k]:2d} " + " ".join(f"{c}={v / cnt[k]:.3g}" for c, v in sorted(agg[k].items())))
PY
