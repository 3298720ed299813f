#!/usr/bin/env python3
"""Static VALU class mix of the compositing kernels' hot loops, priced with the measured issue costs of gfx950.

    python scripts/issue_model.py [out.json]        (needs hipcc only: no GPU)

profiles/r04_valu_issue.txt: with >= 2 waves per SIMD a wave64 VALU instruction issues in 2 cycles (~1.04 ns) if it is
one of v_fma/fmac/fmamk/fmaak/add/sub/mul_f32, v_mov_b32, v_and/or/xor/bitop3_b32, v_add/sub_u32, v_lshrrev_b32 with
VGPR / inline / literal operands only; in 4 cycles (~1.8 ns) if it has an SGPR operand or is anything else (v_cndmask,
v_cmp, DPP, SDWA, min/max/med3, packed f32, ffbl, bfe, lshl_add, cvt, readlane ...); v_exp / v_rcp / v_rsq / v_sqrt take 8
(~3.4 ns).  The hot loop of a kernel = its basic blocks that contain a v_exp_f32 (the alpha evaluation of a trip).
bench.py multiplies the kernel's measured SQ_INSTS_VALU by the mix-weighted cost: roofline.issue.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from pmc_summary import csrc_sha  # noqa: E402

FAST = ("v_fma_f32", "v_fmac_f32", "v_fmamk_f32", "v_fmaak_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32",
        "v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_bitop3_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
        "v_lshrrev_b32", "v_not_b32")
TRANS = ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rcp_iflag_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32")
SGPR = re.compile(r"(?<![\w.])(s\d+|s\[\d+:\d+\]|vcc|vcc_lo|vcc_hi|exec|m0|scc)(?![\w])")

# (source file, kernel-name prefix as rocprofv3 prints it, mangled-name needle)
KERNELS = [
    ("raster_px.hip", "k_praster_fwd<4, true, 0>", "k_praster_fwdILi4ELb1ELi0E"),
    ("raster_g16.hip", "k_qraster_bwd<4, true, 1, false>", "k_qraster_bwdILi4ELb1ELi1ELb0E"),
]


def classify(op, operands):
    if not op.startswith("v_"):
        return None
    base = re.sub(r"_(e32|e64)$", "", op)
    if base.endswith("_dpp") or base.endswith("_sdwa") or "_dpp" in op or "_sdwa" in op:
        return "slow"
    if base in TRANS:
        return "trans"
    if base in FAST:
        srcs = operands.split(",", 1)[1] if "," in operands else ""
        return "slow" if SGPR.search(srcs) else "fast"
    return "slow"


def kernel_blocks(lines, needle):
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and needle in l)
    blocks, cur = {"entry": []}, "entry"
    for l in lines[start + 1:]:
        if l.startswith("\t.end_amdhsa_kernel") or re.match(r"^\.Lfunc_end", l):
            break
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            continue
        t = l.strip()
        if not t or t.startswith((";", ".")):
            continue
        t = t.split(";")[0].strip()
        parts = t.split(None, 1)
        blocks[cur].append((parts[0], parts[1] if len(parts) > 1 else ""))
    return blocks


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r04_issue_model.json")
    res = {"csrc_sha": csrc_sha(), "issue_ns": {"fast": 1.04, "slow": 1.80, "trans": 3.40},
           "note": "VALU instructions of the basic blocks that contain a v_exp_f32 (the trips), by issue class", "kernels": {}}
    for src, name, needle in KERNELS:
        with tempfile.TemporaryDirectory() as td:
            asm = os.path.join(td, "k.s")
            subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fno-slp-vectorize", "--offload-arch=gfx950", "-S", "--cuda-device-only",
                            os.path.join(ROOT, "gsplatloc_amd", "csrc", src), "-o", asm], check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            blocks = kernel_blocks(open(asm).read().splitlines(), needle)
        mix = {"fast": 0, "slow": 0, "trans": 0}
        detail = {}
        nblk = 0
        for label, ins in blocks.items():
            if not any(op.startswith("v_exp_f32") for op, _ in ins):
                continue
            nblk += 1
            for op, operands in ins:
                c = classify(op, operands)
                if c:
                    mix[c] += 1
                    key = re.sub(r"_(e32|e64)$", "", op)
                    detail[key] = detail.get(key, 0) + 1
        tot = sum(mix.values())
        ns = (mix["fast"] * 1.04 + mix["slow"] * 1.80 + mix["trans"] * 3.40) / max(tot, 1)
        res["kernels"][name] = dict(mix, hot_blocks=nblk, valu_per_hot_block=tot / max(nblk, 1), mean_issue_ns=ns,
                                    ops=dict(sorted(detail.items(), key=lambda kv: -kv[1])))
        print(f"{name:40s} hot blocks {nblk}  VALU {tot}  fast {mix['fast']} slow {mix['slow']} trans {mix['trans']}  "
              f"mean {ns:.2f} ns/instr")
    json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
