"""Camera transforms: look, look_at, perspective, get_points_from_angles.

Reference: neural_renderer/look.py:7-45, look_at.py:7-46, perspective.py:5-19,
get_points_from_angles.py:6-24.  The three vertex transforms are one HIP kernel
(csrc/geometry.hip:k_project) with its own backward; the camera basis is rebuilt per thread with the
reference's operation order (normalize = x / (|x| + 1e-5)).
"""
import math

import numpy as np
import torch

from sdn_hip import const_f32, ops


_width_cache = {}


def _vec(x, bs, device, default):
    if x is None:
        x = default
    if not isinstance(x, torch.Tensor):
        x = const_f32(x, device)
    x = x.to(device=device, dtype=torch.float32)
    if x.dim() == 1:
        x = x[None, :]
    if x.shape[0] != bs:
        x = x.expand(bs, 3)
    return x.contiguous()


def look(vertices, eye, direction=None, up=None):
    """"Look" transformation of vertices: (v - eye) @ [x_axis; y_axis; z_axis]^T (look.py:7-45)."""
    assert (vertices.dim() == 3)
    bs, dev = vertices.shape[0], vertices.device
    eye = _vec(eye, bs, dev, None)
    direction = _vec(direction, bs, dev, [0, 0, 1])
    up = _vec(up, bs, dev, [0, 1, 0])
    return ops.ProjectVertices.apply(vertices, ops.CAMERA_LOOK, eye, direction, up, None, 0)


def look_at(vertices, eye, at=None, up=None):
    """"Look at" transformation of vertices (look_at.py:7-46)."""
    assert (vertices.dim() == 3)
    bs, dev = vertices.shape[0], vertices.device
    eye = _vec(eye, bs, dev, None)
    at = _vec(at, bs, dev, [0, 0, 0])
    up = _vec(up, bs, dev, [0, 1, 0])
    return ops.ProjectVertices.apply(vertices, ops.CAMERA_LOOK_AT, eye, at, up, None, 0)


def perspective_width(angle, bs, device):
    """[bs] tensor of tan(angle / 180 * 3.1416) (perspective.py:10-13; 3.1416, not pi)."""
    if isinstance(angle, torch.Tensor):
        a = angle.to(device=device, dtype=torch.float32) / 180. * 3.1416
        return torch.tan(a).reshape(-1).expand(bs).contiguous()
    if isinstance(angle, (list, tuple, np.ndarray)):
        # one angle per batch element, each evaluated like the scalar case (host float32); the device copy is kept per
        # value list (a frame's objects keep their angles over the iterations of the optimisation loop)
        key = (tuple(float(a) for a in np.asarray(angle).reshape(-1)), bs, str(device))
        t = _width_cache.get(key)
        if t is None:
            if len(_width_cache) > 256:
                _width_cache.clear()
            w = np.asarray([ops.perspective_width(a) for a in key[0]], dtype=np.float32)
            t = _width_cache[key] = const_f32(w, device).expand(bs).contiguous()
        return t
    return torch.full((bs,), float(ops.perspective_width(angle)), dtype=torch.float32, device=device)


def perspective(vertices, angle=30.):
    assert (vertices.dim() == 3)
    width = perspective_width(angle, vertices.shape[0], vertices.device)
    return ops.ProjectVertices.apply(vertices, ops.CAMERA_NONE, None, None, None, width, 0)


def get_points_from_angles(distance, elevation, azimuth, degrees=True):
    if isinstance(distance, float) or isinstance(distance, int):
        if degrees:
            elevation = math.radians(elevation)
            azimuth = math.radians(azimuth)
        return (
            distance * math.cos(elevation) * math.sin(azimuth),
            distance * math.sin(elevation),
            -distance * math.cos(elevation) * math.cos(azimuth))
    else:
        if degrees:
            elevation = torch.deg2rad(elevation)
            azimuth = torch.deg2rad(azimuth)
        return torch.stack([
            distance * torch.cos(elevation) * torch.sin(azimuth),
            distance * torch.sin(elevation),
            -distance * torch.cos(elevation) * torch.cos(azimuth),
        ]).t()
