"""
Small geometric helpers with the reference's names and argument meaning (torchdrivesim/utils.py).  They are plain torch
ops used by host-side plumbing (spawn/despawn, observation helpers); the hot path inlines the same arithmetic in HIP.
"""
import collections
from typing import Tuple

import numpy as np
import torch
from torch import Tensor

Resolution = collections.namedtuple('Resolution', ['width', 'height'])


def normalize_angle(angle):
    """Shift by a multiple of 2 pi into [-pi, pi) (utils.py:32-38); floats, numpy arrays and tensors."""
    return (angle + np.pi) % (2 * np.pi) - np.pi


def rotation_matrix(theta: Tensor) -> Tensor:
    """Counter-clockwise 2x2 rotation for angles of shape Sx1 -> Sx2x2 (utils.py:40-53)."""
    c, s = torch.cos(theta), torch.sin(theta)
    return torch.stack([torch.cat([c, -s], dim=-1), torch.cat([s, c], dim=-1)], dim=-2)


def rotate(v: Tensor, angle: Tensor) -> Tensor:
    """Rotate Sx2 points counter-clockwise by Sx1 angles (utils.py:56-69)."""
    return torch.matmul(rotation_matrix(angle), v.unsqueeze(-1)).squeeze(-1)


def relative(origin_xy: Tensor, origin_psi: Tensor, target_xy: Tensor, target_psi: Tensor) -> Tuple[Tensor, Tensor]:
    """Pose of the target in the frame of the origin (utils.py:72-79)."""
    return rotate(target_xy - origin_xy, -origin_psi), normalize_angle(target_psi - origin_psi)


def transform(points: Tensor, pose: Tensor) -> Tensor:
    """Points given relative to a pose (x, y, yaw) -> absolute positions (utils.py:82-96)."""
    xy = pose[..., :2].unsqueeze(-2).expand_as(points)
    psi = pose[..., 2:3].unsqueeze(-2).expand_as(points[..., :1])
    return rotate(points, psi) + xy


def is_inside_polygon(point: Tensor, polygon: Tensor) -> Tensor:
    """Point-in-convex-polygon for BxPx2 points and BxNx2 polygons of either winding (utils.py:99-122)."""
    batch_dims = polygon.dim() - 2
    assert batch_dims >= 0 and polygon.shape[:batch_dims] == point.shape[:batch_dims]
    for _ in point.shape[batch_dims:-1]:
        polygon = polygon.unsqueeze(-3)
    nxt = polygon.roll(-1, dims=-2)
    a = nxt[..., 1] - polygon[..., 1]
    b = polygon[..., 0] - nxt[..., 0]
    c = -a * polygon[..., 0] - b * polygon[..., 1]
    right = a * point[..., None, 0] + b * point[..., None, 1] + c >= 0
    return right.all(dim=-1) | (~right).all(dim=-1)


def isin(x: Tensor, y: Tensor) -> Tensor:
    assert y.dim() == 1
    return (x[..., None] == y).any(-1)


def assert_equal(x, y):
    assert x == y, f'{x} != {y}'
