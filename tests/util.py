"""Shared helpers of the test-suite (input generators, comparison utilities)."""
import numpy as np


def random_soup(rng, bs, nf, scale, zlo=0.5, zhi=5.0):
    """Random triangle soup in NDC: centres uniform in the view, vertex offsets ~ N(0, scale)."""
    c = rng.uniform(-0.95, 0.95, (bs, nf, 1, 2))
    xy = c + rng.normal(0, scale, (bs, nf, 3, 2))
    z = rng.uniform(zlo, zhi, (bs, nf, 3, 1))
    return np.concatenate([xy, z], -1).astype(np.float32)


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.int32) if a.dtype == np.float32 else a


def biteq(a, b):
    return a.shape == b.shape and np.array_equal(bits(a), bits(b))


def posed_mesh(verts, faces, theta=0.6, scale=(3.9, 1.5, 1.6), translation=(2.0, 1.0, -12.0), focal=725.0,
               render_size=384):
    """numpy restatement of the SURVEY section 8(d) config-2 pose: PerspectiveTransform with zoom_to
    (derender3d/models/transforms.py:103-158).  Returns (posed vertices [1,V,3] float32, viewing angle in degrees)."""
    v = verts.astype(np.float64) * np.asarray(scale)
    a, c = np.cos(theta / 2), np.sin(theta / 2)  # quaternion (a, 0, c, 0)
    b = d = 0.0
    T = np.array([[a * a + b * b - c * c - d * d, 2 * b * c - 2 * a * d, 2 * b * d + 2 * a * c],
                  [2 * b * c + 2 * a * d, a * a - b * b + c * c - d * d, 2 * c * d - 2 * a * b],
                  [2 * b * d - 2 * a * c, 2 * c * d + 2 * a * b, a * a - b * b - c * c + d * d]])
    v = v @ T.T + np.asarray(translation)
    x0, y0, z0 = translation
    x = v[:, 0] - x0 / z0 * v[:, 2]
    y = v[:, 1] - y0 / z0 * v[:, 2]
    z = v[:, 2]
    zoom_to = render_size / (2.0 * focal)
    zoom = np.min(np.abs(z) / np.maximum(np.abs(x), np.abs(y))) * zoom_to
    z = z / zoom
    angle = np.arctan(render_size / (2.0 * focal)) / np.pi * 180
    return np.stack([x, y, z], 1)[None].astype(np.float32), angle
