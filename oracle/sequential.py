"""Sequential (loop-per-pixel, loop-per-Gaussian) restatement with HAND-DERIVED
backward passes -- the literal form of the algorithm the HIP kernels implement.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  PARITY UNPINNED (gsplat
1.3.0 sources are absent; the published algorithm is restated).

Pure-Python loops over numpy scalars: use for tiny cases only.  Its job is to
tie the vectorised autograd oracle (``gsplat_oracle``) to the kernel-order
formulas: compositing forward/backward (rasterize_to_pixels fwd/bwd, IDX:14378,
IDX:14279) and the projection vjp to the view matrix (IDX:14270).
"""
from __future__ import annotations

import numpy as np

ALPHA_MAX, ALPHA_MIN, T_STOP = 0.9990000128746033, 0.003921568859368563, 1e-4  # float32 literals of gsplat


def composite_fwd(means2d, conics, colors, opac, W, H, tile, offsets, flat_ids, dtype=np.float64):
    """Per-pixel front-to-back loop (A.3).  Returns colors[H,W,D], alpha[H,W], last_ids[H,W]."""
    f = dtype
    D = colors.shape[1]
    th, tw = offsets.shape
    offs = list(offsets.reshape(-1)) + [len(flat_ids)]
    out = np.zeros((H, W, D), f)
    alpha_img = np.zeros((H, W), f)
    last = np.zeros((H, W), np.int32)
    for i in range(H):
        for j in range(W):
            t = (i // tile) * tw + (j // tile)
            px, py = f(j) + f(0.5), f(i) + f(0.5)
            T = f(1.0)
            acc = np.zeros(D, f)
            cur = 0
            for idx in range(offs[t], offs[t + 1]):
                g = flat_ids[idx]
                dx, dy = f(means2d[g, 0]) - px, f(means2d[g, 1]) - py
                a, b, c = (f(v) for v in conics[g])
                sigma = f(0.5) * (a * dx * dx + c * dy * dy) + b * dx * dy
                al = min(f(ALPHA_MAX), f(opac[g]) * np.exp(-sigma))
                if sigma < 0 or al < f(ALPHA_MIN):
                    continue
                nT = T * (f(1.0) - al)
                if nT <= f(T_STOP):
                    break
                acc += colors[g].astype(f) * (al * T)
                cur = idx
                T = nT
            out[i, j] = acc
            alpha_img[i, j] = f(1.0) - T
            last[i, j] = cur
    return out, alpha_img, last


def composite_bwd(means2d, conics, colors, opac, W, H, tile, offsets, flat_ids,
                  alpha_img, last, v_out, v_alpha, dtype=np.float64):
    """Back-to-front replay (A.4).  Returns v_means2d, v_conics, v_colors, v_opac."""
    f = dtype
    N, D = colors.shape
    th, tw = offsets.shape
    offs = list(offsets.reshape(-1)) + [len(flat_ids)]
    v_m = np.zeros((N, 2), f)
    v_c = np.zeros((N, 3), f)
    v_col = np.zeros((N, D), f)
    v_o = np.zeros(N, f)
    for i in range(H):
        for j in range(W):
            t = (i // tile) * tw + (j // tile)
            px, py = f(j) + f(0.5), f(i) + f(0.5)
            T_final = f(1.0) - f(alpha_img[i, j])
            T = T_final
            buf = np.zeros(D, f)
            vc = v_out[i, j].astype(f)
            va = f(v_alpha[i, j])
            for idx in range(min(int(last[i, j]), offs[t + 1] - 1), offs[t] - 1, -1):
                g = flat_ids[idx]
                dx, dy = f(means2d[g, 0]) - px, f(means2d[g, 1]) - py
                a, b, c = (f(v) for v in conics[g])
                sigma = f(0.5) * (a * dx * dx + c * dy * dy) + b * dx * dy
                vis = np.exp(-sigma)
                o = f(opac[g])
                al = min(f(ALPHA_MAX), o * vis)
                if sigma < 0 or al < f(ALPHA_MIN):
                    continue
                ra = f(1.0) / (f(1.0) - al)
                T = T * ra
                fac = al * T
                v_col[g] += fac * vc
                v_al = np.dot(colors[g].astype(f) * T - buf * ra, vc) + T_final * ra * va
                if o * vis <= f(ALPHA_MAX):
                    v_sigma = -o * vis * v_al
                    v_c[g] += np.array([0.5 * v_sigma * dx * dx, v_sigma * dx * dy, 0.5 * v_sigma * dy * dy], f)
                    v_m[g] += np.array([v_sigma * (a * dx + b * dy), v_sigma * (b * dx + c * dy)], f)
                    v_o[g] += vis * v_al
                buf += colors[g].astype(f) * fac
    return v_m, v_c, v_col, v_o


def _quat_to_R(q):
    q = q / np.linalg.norm(q)
    w, x, y, z = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)],
    ])


def project_bwd_one(mean, quat, scale, V, K, W, H, eps2d, v_mean2d, v_depth, v_conic):
    """Hand-derived vjp of the fused projection for ONE valid Gaussian (A.5).
    Returns v_mean[3], v_covar[3,3] (world), v_R[3,3], v_t[3] (this Gaussian's share of v_viewmat)."""
    fx, fy = K[0, 0], K[1, 1]
    R, t = V[:3, :3], V[:3, 3]
    Rq = _quat_to_R(quat)
    M = Rq * scale[None, :]
    S = M @ M.T
    mc = R @ mean + t
    Sc = R @ S @ R.T
    x, y, z = mc
    limx, limy = 1.3 * 0.5 * W / fx, 1.3 * 0.5 * H / fy
    rz = 1.0 / z
    rz2 = rz * rz
    tx = z * min(limx, max(-limx, x * rz))
    ty = z * min(limy, max(-limy, y * rz))
    J = np.array([[fx * rz, 0.0, -fx * tx * rz2], [0.0, fy * rz, -fy * ty * rz2]])
    cov2 = J @ Sc @ J.T + eps2d * np.eye(2)
    inv = np.linalg.inv(cov2)
    Vc = np.array([[v_conic[0], 0.5 * v_conic[1]], [0.5 * v_conic[1], v_conic[2]]])
    v_cov2 = -inv @ Vc @ inv
    v_Sc = J.T @ v_cov2 @ J
    v_J = v_cov2 @ J @ Sc.T + v_cov2.T @ J @ Sc
    v_mc = np.array([fx * rz * v_mean2d[0], fy * rz * v_mean2d[1],
                     -(fx * x * v_mean2d[0] + fy * y * v_mean2d[1]) * rz2])
    rz3 = rz2 * rz
    if -limx <= x * rz <= limx:
        v_mc[0] += -fx * rz2 * v_J[0, 2]
    else:
        v_mc[2] += -fx * rz3 * v_J[0, 2] * tx
    if -limy <= y * rz <= limy:
        v_mc[1] += -fy * rz2 * v_J[1, 2]
    else:
        v_mc[2] += -fy * rz3 * v_J[1, 2] * ty
    v_mc[2] += (-fx * rz2 * v_J[0, 0] - fy * rz2 * v_J[1, 1]
                + 2.0 * fx * tx * rz3 * v_J[0, 2] + 2.0 * fy * ty * rz3 * v_J[1, 2])
    v_mc[2] += v_depth
    v_R = np.outer(v_mc, mean) + v_Sc @ R @ S.T + v_Sc.T @ R @ S
    v_t = v_mc.copy()
    v_mean = R.T @ v_mc
    v_S = R.T @ v_Sc @ R
    return v_mean, v_S, v_R, v_t


def covar_to_quat_scale_vjp(quat, scale, v_S):
    """vjp of Sigma=(R S)(R S)^T to (quat, scale), quat normalised inside (A.5)."""
    qn = np.linalg.norm(quat)
    w, x, y, z = quat / qn
    R = _quat_to_R(quat)
    M = R * scale[None, :]
    v_M = (v_S + v_S.T) @ M
    v_Rm = v_M * scale[None, :]
    v_scale = np.array([R[:, k] @ v_M[:, k] for k in range(3)])
    vR = v_Rm
    v_qn = np.array([
        2 * (x * (vR[2, 1] - vR[1, 2]) + y * (vR[0, 2] - vR[2, 0]) + z * (vR[1, 0] - vR[0, 1])),
        2 * (-2 * x * (vR[1, 1] + vR[2, 2]) + y * (vR[1, 0] + vR[0, 1]) + z * (vR[2, 0] + vR[0, 2]) + w * (vR[2, 1] - vR[1, 2])),
        2 * (x * (vR[1, 0] + vR[0, 1]) - 2 * y * (vR[0, 0] + vR[2, 2]) + z * (vR[2, 1] + vR[1, 2]) + w * (vR[0, 2] - vR[2, 0])),
        2 * (x * (vR[2, 0] + vR[0, 2]) + y * (vR[2, 1] + vR[1, 2]) - 2 * z * (vR[0, 0] + vR[1, 1]) + w * (vR[1, 0] - vR[0, 1])),
    ])
    qh = quat / qn
    v_quat = (v_qn - np.dot(v_qn, qh) * qh) / qn
    return v_quat, v_scale
