"""CPU ICP baseline (libgsloc_icp.so, include/gsloc_icp.h): k-NN against scipy, normal/covariance
estimation on analytic surfaces, registration of a synthetic room under a known motion for every
registration type, the Scan2ScanICP / sequence plumbing, and header <-> binding <-> library agreement.
No fixture of the reference covers small_gicp: parity unpinned (see the header)."""
import os
import re
from types import SimpleNamespace

import numpy as np
import pytest

import gsplatloc_amd.small_gicp as sg
from gsplatloc_amd.component import Scan2ScanICP
from gsplatloc_amd.icp_eval import rotation_error, run_icp_sequence, translation_error


@pytest.fixture(scope="module", autouse=True)
def _built():
    if not os.path.exists(sg.library_path()):
        sg.build_library()


def _rot(axis, deg):
    axis = np.asarray(axis, dtype=np.float64)
    axis /= np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    a = np.deg2rad(deg)
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K


def _pose(axis, deg, t):
    T = np.eye(4)
    T[:3, :3] = _rot(axis, deg)
    T[:3, 3] = t
    return T


def _room(rng, n, noise=0.002):
    """Points on the six walls of a 6 x 4 x 3 m box seen from inside (world frame)."""
    u = rng.uniform(-1, 1, (n, 3))
    f = rng.integers(0, 6, n)
    for k in range(3):
        u[f == 2 * k, k] = -1
        u[f == 2 * k + 1, k] = 1
    return u * np.array([3.0, 2.0, 1.5]) + np.array([0.3, 0.1, 0.2]) + rng.normal(scale=noise, size=(n, 3))


def _to_camera(c2w, pts_w):
    w2c = np.linalg.inv(c2w)
    return pts_w @ w2c[:3, :3].T + w2c[:3, 3]


def test_header_binding_and_library_agree(repo_root):
    txt = open(os.path.join(repo_root, "include", "gsloc_icp.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    declared = sorted(set(re.findall(r"\b(gsl_icp_[a-z_0-9]+)\s*\(", txt)))
    assert declared == sg.exported_symbols()
    lib = sg.load_library()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.gsl_icp_version().decode().startswith("gsloc_icp")


@pytest.mark.parametrize("k", [1, 5, 20])
def test_knn_matches_ckdtree(k):
    from scipy.spatial import cKDTree

    rng = np.random.default_rng(1)
    pts = rng.normal(size=(5000, 3))
    pts[100:110] = pts[100]  # duplicates: ties resolve to the smaller index
    tree = sg.KdTree(sg.PointCloud(pts), num_threads=4)
    q = np.concatenate([rng.normal(size=(700, 3)), pts[:300]])
    idx, d2 = tree.batch_knn_search(q, k, num_threads=4)
    dd, ii = cKDTree(pts).query(q, k)
    dd, ii = dd.reshape(len(q), k), ii.reshape(len(q), k)
    np.testing.assert_allclose(d2, dd ** 2, rtol=0, atol=1e-12)
    distinct = (np.diff(dd, axis=1) > 1e-12).all(axis=1) if k > 1 else np.ones(len(q), dtype=bool)
    distinct &= ~np.isin(ii, np.arange(100, 110)).any(axis=1)
    assert (idx[distinct] == ii[distinct]).all()
    assert (np.diff(d2, axis=1) >= 0).all()
    # among exact duplicates the smaller index comes first
    first, _ = tree.batch_knn_search(pts[105][None], 1)
    assert first[0, 0] == 100


def test_knn_small_and_empty_clouds():
    tree = sg.KdTree(sg.PointCloud(np.array([[0.0, 0, 0], [1, 0, 0], [0, 2, 0]])))
    idx, d2 = tree.batch_knn_search(np.array([[0.1, 0, 0]]), 5)
    assert idx[0].tolist() == [0, 1, 2, -1, -1] and np.isinf(d2[0, 3:]).all()
    empty = sg.KdTree(sg.PointCloud(np.zeros((0, 3))))
    idx, _ = empty.batch_knn_search(np.zeros((2, 3)), 1)
    assert (idx == -1).all()
    with pytest.raises(RuntimeError, match="bad argument"):
        tree.batch_knn_search(np.zeros((1, 3)), 0)


def test_normals_and_regularised_covariances():
    rng = np.random.default_rng(2)
    n = 4000
    plane = np.c_[rng.uniform(-1, 1, (n, 2)), np.full(n, 2.0)]  # z = 2 in front of the sensor
    c = sg.PointCloud(plane)
    sg.estimate_normals_covariances(c, sg.KdTree(c, 4), num_neighbors=20, num_threads=4)
    nrm = c.normals()[:, :3]
    np.testing.assert_allclose(nrm, np.tile([0, 0, -1.0], (n, 1)), atol=1e-9)  # flipped towards the origin
    cov = c.covs()[:, :3, :3]
    ev = np.linalg.eigvalsh(cov)
    np.testing.assert_allclose(ev, np.tile([1e-3, 1, 1], (n, 1)), atol=1e-9)
    np.testing.assert_allclose(np.einsum("nij,nj->ni", cov, nrm), 1e-3 * nrm, atol=1e-9)
    # sphere: normals are radial and point inwards (towards the sensor at the centre)
    s = rng.normal(size=(6000, 3))
    s = 3.0 * s / np.linalg.norm(s, axis=1, keepdims=True)
    cs = sg.PointCloud(s)
    sg.estimate_normals_covariances(cs, None, 15, 4)
    cosang = np.einsum("ni,ni->n", cs.normals()[:, :3], -s / 3.0)
    assert cosang.min() > 0.99
    with pytest.raises(RuntimeError, match="normals/covariances missing"):
        sg.align(sg.PointCloud(plane), sg.PointCloud(plane), None, registration_type="GICP")


@pytest.mark.parametrize("method,tol_t,tol_r", [("ICP", 1.5e-2, 0.4), ("PLANE_ICP", 1e-3, 0.02), ("GICP", 1e-3, 0.02)])
def test_registration_recovers_a_known_motion(method, tol_t, tol_r):
    rng = np.random.default_rng(3)
    T_true = _pose([0.2, 1.0, 0.3], 2.0, [0.03, -0.02, 0.04])  # source frame -> target frame
    tgt = _room(rng, 30000)
    src = _to_camera(T_true, _room(rng, 30000))  # a different sampling of the same walls
    ct, cs = sg.PointCloud(tgt), sg.PointCloud(src)
    tt = sg.KdTree(ct, 4)
    sg.estimate_normals_covariances(ct, tt, 20, 4)
    sg.estimate_normals_covariances(cs, sg.KdTree(cs, 4), 20, 4)
    r = sg.align(ct, cs, tt, np.eye(4), 0.5, method, num_threads=4)
    assert r.converged and r.num_inliers > 29000 and r.iterations <= 20
    assert translation_error(r.T_target_source, T_true) < tol_t
    assert rotation_error(r.T_target_source, T_true) < tol_r
    # a good initial guess converges at once and a tight gate rejects nothing it should keep
    r2 = sg.align(ct, cs, tt, T_true, 0.1, method, num_threads=4)
    assert r2.converged and r2.iterations <= 3 and r2.num_inliers > 29000
    # thread count does not change the answer (block-ordered reduction)
    r1 = sg.align(ct, cs, tt, np.eye(4), 0.5, method, num_threads=1)
    np.testing.assert_array_equal(r1.T_target_source, r.T_target_source)


def test_voxel_downsampling_and_preprocess():
    rng = np.random.default_rng(4)
    pts = rng.uniform(0, 1, (20000, 3))
    down = sg.voxelgrid_sampling(pts, 0.25)
    assert down.size() == 64  # every voxel of the 4x4x4 grid is hit
    p = down.points()[:, :3]
    assert ((p // 0.25) == np.floor(p / 0.25)).all()
    cells = np.floor(pts / 0.25).astype(int)
    key = cells[:, 0] + 4 * cells[:, 1] + 16 * cells[:, 2]
    want = np.stack([pts[key == k].mean(0) for k in range(64)])  # ascending key order: x fastest
    np.testing.assert_allclose(p, want, atol=1e-12)
    cloud, tree = sg.preprocess_points(pts, 0.25, num_neighbors=10, num_threads=2)
    assert cloud.size() == 64 and cloud.normals().shape == (64, 4) and tree.cloud is cloud


def test_scan2scan_sequence_tracks_a_trajectory():
    rng = np.random.default_rng(5)
    poses = [_pose([0, 1, 0], 1.5 * i, [0.02 * i, 0.0, 0.03 * i]) for i in range(4)]
    frames = [SimpleNamespace(points=_to_camera(c2w, _room(rng, 20000)), pose=c2w) for c2w in poses]
    res = run_icp_sequence(frames, "GICP", max_corresponding_distance=0.3, num_threads=4)
    assert res["frames"] == 4 and len(res["eT"]) == 3
    assert res["ATE"] < 2e-3 and res["AAE"] < 0.05
    # the tracker object itself: T_world_camera accumulates the relative motions from the first pose
    icp = Scan2ScanICP(max_corresponding_distance=0.3, registration_type="PLANE_ICP", num_threads=4)
    assert icp.align(frames[0].points, poses[0]) is poses[0]
    for f in frames[1:]:
        est = icp.align(f.points, None, np.identity(4))
    assert translation_error(est, poses[-1]) < 5e-3 and rotation_error(est, poses[-1]) < 0.1
    with pytest.raises(NotImplementedError):
        Scan2ScanICP(registration_type="HYBRID", implementation="open3d")


def test_icp_evaluation_on_a_replica_format_sequence(tmp_path, capsys):
    """BASELINE.json configs[0] end to end on the CPU: Replica-format files -> reader -> scan-to-scan GICP ->
    ATE/AAE report (the plumbing of icps_eval.py), and the command-line entry."""
    import json

    from gsplatloc_amd import icp_eval
    from gsplatloc_amd.data import Replica
    from tests.test_data import write_replica

    write_replica(tmp_path, name="office0", W=160, H=120, n=4)
    data = Replica("office0", input_folder=tmp_path)
    assert data[0].points.shape == (160 * 120, 3)
    res = run_icp_sequence(data, "GICP", max_corresponding_distance=0.2, num_threads=4)
    assert res["frames"] == 4 and len(res["eT"]) == 3
    # 1 cm / 0.3 degree steps inside a 6 m room sampled at 160x120: centimetre-level registration is the bar here
    assert res["ATE"] < 2e-2 and res["AAE"] < 0.5, res
    plane = run_icp_sequence(data, "PLANE_ICP", max_corresponding_distance=0.2, num_threads=4, max_images=3)
    assert plane["frames"] == 3
    out = tmp_path / "icp.json"
    icp_eval.main(["--dataset", "Replica", "--rooms", "office0", "--root", str(tmp_path), "--method", "GICP",
                   "--threads", "4", "--max-images", "3", "--out", str(out)])
    report = json.loads(out.read_text())
    assert report["office0"]["method"] == "GICP" and report["office0"]["frames"] == 3
    assert "office0" in capsys.readouterr().out
