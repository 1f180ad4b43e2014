"""cross and lighting (neural_renderer/cross.py:8-61, lighting.py:8-52)."""
import numpy as np
import torch

from sdn_hip import ops


def cross(a, b):
    """Row-wise cross product of two [n, 3] tensors (cross.py:25-38); autograd gives cross.py:47-53."""
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != 3 or b.shape[1] != 3 or a.shape[0] != b.shape[0]:
        raise ValueError('cross expects two [n, 3] tensors')
    c0 = a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1]
    c1 = a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2]
    c2 = a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]
    return torch.stack([c0, c1, c2], dim=1)


def _color(x, bs, device):
    if not isinstance(x, torch.Tensor):
        from sdn_hip import const_f32
        x = const_f32(x, device)
    x = x.to(device=device, dtype=torch.float32)
    if x.dim() == 1:
        x = x[None, :].expand(bs, 3)
    return x


def lighting(
        faces, textures, intensity_ambient=0.5, intensity_directional=0.5, color_ambient=(1, 1, 1),
        color_directional=(1, 1, 1), direction=(0, 1, 0)):
    bs, nf = faces.shape[:2]
    dev = faces.device
    color_ambient = _color(color_ambient, bs, dev)
    color_directional = _color(color_directional, bs, dev)
    direction = _color(direction, bs, dev)

    light = torch.zeros((bs, nf, 3), dtype=torch.float32, device=dev)
    if intensity_ambient != 0:
        light = light + intensity_ambient * color_ambient[:, None, :]
    if intensity_directional != 0:
        normals = ops.FaceNormals.apply(faces)  # normalize(cross(v0 - v1, v2 - v1)), lighting.py:37-41
        d = direction[:, None, :]
        cos = torch.relu((normals[:, :, 0] * d[:, :, 0] + normals[:, :, 1] * d[:, :, 1]) + normals[:, :, 2] * d[:, :, 2])
        light = light + intensity_directional * (color_directional[:, None, :] * cos[:, :, None])
    return textures * light[:, :, None, None, None, :]
