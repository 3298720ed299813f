"""Register / LDS budget of the hot kernels, from hipcc's resource remarks (no GPU needed): the compositing kernels
are occupancy-sensitive (round 2: k_mraster_bwd from 120 to 130 registers cost 17 % on hardware; round 3: the
per-quadrant backward at 9.8 KB of LDS -- 4 waves/SIMD -- ran 264 us, at 8.4 KB 257 us), so the budget is pinned here."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gsplatloc_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


def _resources(src, *extra):
    cmd = [HIPCC, "-O3", "-std=c++17", "-fno-slp-vectorize", "--offload-arch=gfx950", "--cuda-device-only", "-c", src, "-o", os.devnull,
           "-Rpass-analysis=kernel-resource-usage", *extra]
    res = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    out, cur = {}, None
    for line in res.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+(VGPRs|AGPRs|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|ScratchSize \[bytes/lane\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split(" ")[0]] = int(m.group(2))
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_compositing_kernels_keep_their_occupancy():
    fused = _resources("fused.hip")
    px = _resources("raster_px.hip")
    g16 = _resources("raster_g16.hip")
    # RGB+ED backward, depth-only upstream gradient (GsplatLoc's loss): one wave per workgroup, >= 4.75 waves/SIMD by
    # LDS (19 workgroups of <= 8.5 KB per CU) and <= 96 registers
    bwd = [v for k, v in g16.items() if "k_qraster_bwdILi4ELb1ELi1ELb0" in k]
    assert len(bwd) == 1
    b = bwd[0]
    assert b["ScratchSize"] == 0 and b["VGPRs"] + b.get("AGPRs", 0) <= 96 and b["LDS"] <= 8704, b
    full = [v for k, v in g16.items() if "k_qraster_bwdILi4ELb1ELi4ELb0" in k][0]  # full-colour tiles
    assert full["ScratchSize"] == 0 and full["VGPRs"] <= 128 and full["LDS"] <= 12 * 1024, full
    det = [v for k, v in fused.items() if "k_mraster_bwdILi4ELb1ELb1" in k][0]  # deterministic variant: may be slower,
    assert det["ScratchSize"] == 0 and det["LDS"] <= 64 * 1024, det            # must not spill, two workgroups per CU
    fwd = [v for k, v in px.items() if "k_praster_fwdILi4ELb1" in k][0]
    assert fwd["ScratchSize"] == 0 and fwd["Occupancy"] >= 6, fwd


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_sort_and_projection_kernels_stay_in_registers():
    binning = _resources("binning.hip")
    sort = {k: v for k, v in binning.items() if "k_tile_sort" in k}
    assert len(sort) == 3  # one tile per wave (lists up to 1024 / up to 2048 keys in registers), one tile per workgroup
    for k, s in sort.items():
        # no spills; LDS (32 KB, four workgroups per CU) for the long-list path / the workgroup kernel's merge buffers.  The
        # 16-keys-per-lane instance (frames whose lists stay below 1024 keys: workload R) must keep four waves per SIMD -- a
        # frame of 3 225 tiles is then resident at once; the 32-keys-per-lane network costs a wave (141 VGPR)
        assert s["ScratchSize"] == 0 and s["LDS"] <= 32 * 1024 + 64, (k, s)
        assert s["Occupancy"] >= (3 if "k_tile_sortILi5" in k else 4), (k, s)
    fused = _resources("fused.hip")
    for k, v in fused.items():
        if "k_fproject" in k or "k_ftile_scan" in k:
            assert v["ScratchSize"] == 0, (k, v)
        if "k_fprojectILb" in k:   # forward projection (binned or not): 512-thread workgroups must fit twice per CU
            assert v["VGPRs"] <= 128, (k, v)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_the_eight_group_variant_of_the_backward_still_builds():
    """-DGSL_NG=8 (eight 4x2 half-block lists per quadrant instead of four 4x4 block lists) was measured slower and is
    not what the library is built with (NOTES.md), but the forward's hit words and the backward's body stay generic over
    it: the variant must keep compiling, without scratch."""
    g16 = _resources("raster_g16.hip", "-DGSL_NG=8")
    b = [v for k, v in g16.items() if "k_qraster_bwdILi4ELb1ELi1ELb0" in k]
    assert len(b) == 1 and b[0]["ScratchSize"] == 0 and b[0]["LDS"] <= 9 * 1024, b
    px = _resources("raster_px.hip", "-DGSL_NG=8")
    assert [v for k, v in px.items() if "k_praster_fwdILi4ELb1" in k][0]["ScratchSize"] == 0
