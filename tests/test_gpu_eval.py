"""GPU: the reference's evaluation protocol end to end on a synthetic Replica-format sequence, through the CLI
(`python -m gsplatloc_amd.eval`): every frame pair is tracked from the ground-truth pose of frame i
(/root/reference/src/my_gsplat/gs_trainer_total.py:49-63), at most 2000 iterations, early stop with patience 200
after step 100 and the error read at the minimum-loss iterate (:160-185), ATE / AAE = RMSE over the frames
(eval/utils.py:113-119).  The levels asserted are those of the reference's Replica table
(/root/reference/docs/res.json:20-23: ATE 1.6e-4 .. 2.4e-4 m, AAE ~0.01 deg): ATE <= 2e-4 m, AAE <= 0.01 deg.
The CPU ICP baseline of BASELINE.json configs[0] runs on the same files and is reported beside it
(/root/reference/res.json:2-147 has GICP at centimetres on Replica)."""
import json
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


from gsplatloc_amd.synthetic import write_replica_sequence as _write_sequence  # noqa: E402


def test_sequence_ate_with_the_reference_protocol(tmp_path, repo_root):
    W, H, n = 640, 480, 5
    _write_sequence(tmp_path, W, H, n)
    out = tmp_path / "res.json"
    cmd = [sys.executable, "-m", "gsplatloc_amd.eval", "--dataset", "Replica", "--rooms", "room0", "--root", str(tmp_path),
           "--num-iters", "2000", "--out", str(out), "--verbose"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=repo_root)
    assert res.returncode == 0, res.stderr[-3000:]
    r = json.loads(out.read_text())["room0"]["gsplatloc_amd"]
    print("[eval] gsplatloc_amd", json.dumps(r))
    print(res.stdout[-1500:])
    assert r["frames_with_result"] == r["frames"] == n - 1
    # the ICP baseline on the same files (CPU)
    icp_out = tmp_path / "icp.json"
    icp = subprocess.run([sys.executable, "-m", "gsplatloc_amd.icp_eval", "--dataset", "Replica", "--rooms", "room0", "--root",
                          str(tmp_path), "--method", "GICP", "--stride", "4", "--out", str(icp_out)], capture_output=True,
                         text=True, timeout=900, cwd=repo_root)
    assert icp.returncode == 0, icp.stderr[-3000:]
    b = json.loads(icp_out.read_text())["room0"]
    print("[eval] GICP baseline", json.dumps(b))
    assert r["ATE"] <= 2e-4, r
    assert r["AAE"] <= 0.01, r
    # (on this noise-free box room GICP is at its best -- perfect planes, no occlusion -- so the two are only reported
    #  side by side; the reference's claim against GICP is about real scans, res.json:2-147)
    assert b["ATE"] < 0.01 and b["frames"] == n


def test_tum_format_sequence_with_invalid_depth(tmp_path, repo_root):
    """The same protocol through the TUM reader (timestamp association, quaternion poses, crop_edge, depth scale 5000) on
    frames with patches of invalid (zero) depth -- BASELINE.json configs[2]'s kind of input: the Gaussians of the
    invalid pixels sit at the previous camera's origin and are culled by the near plane or pile up in one tile; the
    target depth has holes the mask (depth != 0) must respect.  Level asserted: the reference's TUM table is at
    centimetres (docs/res.json: fr1/desk ATE 0.0103 m); on noise-free synthetic depth the tracker stays below 1 mm."""
    from gsplatloc_amd.synthetic import write_tum_sequence
    W, H, n = 640, 480, 5
    write_tum_sequence(tmp_path, W, H, n)
    out = tmp_path / "res_tum.json"
    cmd = [sys.executable, "-m", "gsplatloc_amd.eval", "--dataset", "TUM", "--rooms", "freiburg1_desk", "--root",
           str(tmp_path), "--num-iters", "2000", "--out", str(out), "--verbose"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=repo_root)
    assert res.returncode == 0, res.stderr[-3000:]
    r = json.loads(out.read_text())["freiburg1_desk"]["gsplatloc_amd"]
    print("[eval] gsplatloc_amd on the TUM-format sequence", json.dumps(r))
    print(res.stdout[-1200:])
    assert r["frames_with_result"] == r["frames"] == n - 1
    assert r["ATE"] <= 1e-3, r
    assert r["AAE"] <= 0.05, r
