#!/usr/bin/env python3
"""Condense rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into profiles/<tag>_pmc_traffic.json.

    python scripts/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [<SQ csv> ...]

Further counter_collection.csv files (SQ_* passes: instruction counts, LDS array cycles) are folded in as `sq` per kernel
(mean per launch, summed over the chip as rocprofv3 reports them); bench.py's roofline.issue reads them.

Per kernel: mean bytes per launch.  FETCH_SIZE and WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE
counts a 128-B request as 64 B, so fetch bytes are doubled (MI355X_MICROARCH.md, "HBM [CDNA4]"); the guide
calibrates that factor on wide streaming reads only, so for the 16-B gathers of the compositing kernels the
doubled figure is an upper estimate.
"""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict


def csrc_sha():
    """Hash of the kernel sources the counters were collected on: bench.py only quotes roofline.traffic from a summary
    whose hash matches the library it runs (a committed constant goes stale the moment a kernel changes)."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gsplatloc_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.h"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def per_kernel(path, counter):
    acc = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"].split("(")[0]
            name = name.replace("void ", "").replace("gsl::", "")
            acc[name].append(float(row["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def all_counters(path):
    acc = defaultdict(lambda: defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("gsl::", "")
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    sq = defaultdict(dict)
    for path in sys.argv[4:]:
        for k, cs in all_counters(path).items():
            sq[k].update(cs)
    out = {}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("k_"):
            continue
        fb = fetch.get(k, (0.0, 0))[0] * 1024.0 * 2.0
        wb = write.get(k, (0.0, 0))[0] * 1024.0
        out[k] = {"fetch_bytes": fb, "write_bytes": wb, "hbm_bytes": fb + wb,
                  "launches": max(fetch.get(k, (0, 0))[1], write.get(k, (0, 0))[1])}
        if k in sq:
            out[k]["sq"] = dict(sq[k])
            if "SQ_BUSY_CYCLES" in sq[k]:  # (summed over the 32 shader engines of the chip)
                out[k]["sq"]["SQ_BUSY_CYCLES_per_se"] = sq[k]["SQ_BUSY_CYCLES"] / 32.0
    json.dump({"csrc_sha": csrc_sha(), "note": "mean per launch; fetch = FETCH_SIZE KiB x 1024 x 2 (gfx950 correction), write = WRITE_SIZE KiB x 1024",
               "kernels": out}, open(sys.argv[3], "w"), indent=1)
    for k, v in out.items():
        print(f"{k:48s} fetch {v['fetch_bytes']/1e6:9.1f} MB  write {v['write_bytes']/1e6:9.1f} MB  x{v['launches']}")


if __name__ == "__main__":
    main()
