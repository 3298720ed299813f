"""Rotation parametrisations (mirror of /root/reference/src/my_gsplat/transform.py).
The two kornia.geometry conversions the reference calls (transform.py:65-66,84) are restated
here for wxyz quaternions, kornia 0.7 conventions."""
import torch
from torch import Tensor
from torch.nn import functional as F


def normalize_quaternion(quaternion: Tensor, eps: float = 1e-12) -> Tensor:
    return F.normalize(quaternion, p=2.0, dim=-1, eps=eps)


def _quaternion_to_rotation_matrix(quaternion: Tensor) -> Tensor:
    q = normalize_quaternion(quaternion)
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    tx, ty, tz = 2.0 * x, 2.0 * y, 2.0 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    one = torch.ones_like(w)
    m = torch.stack((one - (tyy + tzz), txy - twz, txz + twy,
                     txy + twz, one - (txx + tzz), tyz - twx,
                     txz - twy, tyz + twx, one - (txx + tyy)), dim=-1)
    return m.reshape(quaternion.shape[:-1] + (3, 3))


def rotation_6d_to_matrix(d6: Tensor) -> Tensor:
    """Role of transform.py:7-29 (Zhou et al.'s 6D rotation): the two 3-vectors of `d6` are orthonormalised
    (Gram-Schmidt) into the first two ROWS of the matrix, the third row is their cross product."""
    first, second = d6[..., 0:3], d6[..., 3:6]
    row0 = first / first.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    second = second - row0 * (row0 * second).sum(dim=-1, keepdim=True)
    row1 = second / second.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    return torch.stack([row0, row1, torch.linalg.cross(row0, row1, dim=-1)], dim=-2)


def matrix_to_rotation_6d(matrix: Tensor) -> Tensor:
    """Role of transform.py:32-47: the first two rows, flattened."""
    return matrix[..., 0:2, :].flatten(start_dim=-2).clone()


def quat_to_rotation_matrix(quaternion: Tensor) -> Tensor:
    """transform.py:50-66: normalise, then wxyz quaternion -> [3,3]."""
    return _quaternion_to_rotation_matrix(normalize_quaternion(quaternion))


def rotation_matrix_to_quaternion(rotation_matrix: Tensor, eps: float = 1e-8) -> Tensor:
    """transform.py:69-84 (kornia rotation_matrix_to_quaternion, returns wxyz)."""
    tiny = torch.finfo(rotation_matrix.dtype).tiny

    def sdiv(n, d):
        return n / torch.clamp(d, min=tiny)

    v = rotation_matrix.reshape(rotation_matrix.shape[:-2] + (9,))
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.chunk(v, 9, dim=-1)
    trace = m00 + m11 + m22

    def c0():
        sq = torch.sqrt(trace + 1.0 + eps) * 2.0
        return torch.cat((0.25 * sq, sdiv(m21 - m12, sq), sdiv(m02 - m20, sq), sdiv(m10 - m01, sq)), -1)

    def c1():
        sq = torch.sqrt(1.0 + m00 - m11 - m22 + eps) * 2.0
        return torch.cat((sdiv(m21 - m12, sq), 0.25 * sq, sdiv(m01 + m10, sq), sdiv(m02 + m20, sq)), -1)

    def c2():
        sq = torch.sqrt(1.0 + m11 - m00 - m22 + eps) * 2.0
        return torch.cat((sdiv(m02 - m20, sq), sdiv(m01 + m10, sq), 0.25 * sq, sdiv(m12 + m21, sq)), -1)

    def c3():
        sq = torch.sqrt(1.0 + m22 - m00 - m11 + eps) * 2.0
        return torch.cat((sdiv(m10 - m01, sq), sdiv(m02 + m20, sq), sdiv(m12 + m21, sq), 0.25 * sq), -1)

    w2 = torch.where(m11 > m22, c2(), c3())
    w1 = torch.where((m00 > m11) & (m00 > m22), c1(), w2)
    return torch.where(trace > 0.0, c0(), w1)
