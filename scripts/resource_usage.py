#!/usr/bin/env python3
"""Static resource table of every kernel: hipcc -Rpass-analysis=kernel-resource-usage on csrc/*.hip
(no GPU needed).  python scripts/resource_usage.py > profiles/r01_kernel_resources.txt"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gsplatloc_amd", "csrc")
rows = []
for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
    out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fPIC", "--offload-arch=gfx950",
                          "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"],
                         capture_output=True, text=True, cwd=CSRC).stderr
    cur = None
    for line in out.splitlines():
        m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|"
                      r"VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            name = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(.*", "", name).replace("void ", "").replace("gsl::", "")
            cur = {"kernel": name, "file": os.path.basename(src)}
            rows.append(cur)
        elif cur is not None:
            cur[k.split(" [")[0]] = v
print(f"{'kernel':44s} {'file':14s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'spill':>6s} {'LDS B':>7s} {'waves/SIMD':>10s}")
for r in rows:
    print(f"{r['kernel'][:44]:44s} {r['file']:14s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('TotalSGPRs', '?'):>5s} "
          f"{r.get('ScratchSize', '?'):>8s} {r.get('VGPRs Spill', '?'):>6s} {r.get('LDS Size', '?'):>7s} {r.get('Occupancy', '?'):>10s}")
