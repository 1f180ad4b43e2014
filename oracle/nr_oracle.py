"""torch-CPU restatement of the neural_renderer host ops and of the derender3d render bridge.

TEST INFRASTRUCTURE ONLY -- never imported by the product (3d-sdn_amd/).

What is restated (file:line under /root/reference/geometric/):
  normalize           chainer.functions.normalize as used at neural_renderer/look.py:29-31
                      (third-party Chainer 4.1.0, environment.yml:8: x / (||x||_2 + 1e-5))
  cross               neural_renderer/cross.py:25-57
  look / look_at      neural_renderer/look.py:7-45, look_at.py:7-46
  perspective         neural_renderer/perspective.py:5-19  (note 3.1416, not pi)
  vertices_to_faces   neural_renderer/vertices_to_faces.py:4-21
  lighting            neural_renderer/lighting.py:8-52
  Rasterize           neural_renderer/rasterize.py:19-894 (kernels in raster_oracle.c)
  rasterize_rgbad &c  neural_renderer/rasterize.py:897-1057 (flip, 2x2 average pool)
  NRRenderer          neural_renderer/renderer.py:11-110 + derender3d/models/renderer.py:19-127
  SDNRenderer         derender3d/models/renderer.py:130-272 (x flip, normal sign fix)

All arithmetic is float32 on the CPU with an explicit left-to-right operation order and no fused
multiply-add, so that the HIP kernels (compiled with -ffp-contract=off) can reproduce the
post-projection coordinates bit for bit; only tan() of the viewing angle is a transcendental, and
both sides take it from the host (numpy float32).
Gradients of the host ops come from torch autograd on these same expressions.
"""
import math

import numpy as np
import torch

from . import raster_np

DEFAULT_IMAGE_SIZE = 256
DEFAULT_ANTI_ALIASING = True
DEFAULT_NEAR = 0.1
DEFAULT_FAR = 100
DEFAULT_EPS = 1e-4
DEFAULT_BACKGROUND_COLOR = (0, 0, 0)


def _f32(x):
    if isinstance(x, torch.Tensor):
        return x.to(torch.float32)
    return torch.tensor(np.asarray(x, dtype=np.float32))


def sum3(a, b, c):
    return (a + b) + c


class _ExactSqrt(torch.autograd.Function):
    """Correctly rounded float32 sqrt (numpy); torch's CPU sqrt goes through SLEEF and is off by an ulp on
    ~1 % of inputs, which would make the oracle, not the kernel, the inexact side."""

    @staticmethod
    def forward(ctx, x):
        y = torch.from_numpy(np.sqrt(x.detach().numpy()))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return g / (2 * y)


def normalize(x, eps=1e-5):
    """[n,3] -> x / (sqrt(x0^2+x1^2+x2^2) + eps).  Gradient at the zero vector is NaN, as in Chainer."""
    n = _ExactSqrt.apply(sum3(x[:, 0:1] * x[:, 0:1], x[:, 1:2] * x[:, 1:2], x[:, 2:3] * x[:, 2:3])) + eps
    return x / n


def cross(a, b):
    c0 = a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1]
    c1 = a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2]
    c2 = a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]
    return torch.stack([c0, c1, c2], dim=1)


def _apply_rotation(vertices, eye, x_axis, y_axis, z_axis):
    # (v - eye) @ r^T with r rows = (x_axis, y_axis, z_axis); explicit sum order
    v = vertices - eye[:, None, :]
    out = []
    for axis in (x_axis, y_axis, z_axis):
        a = axis[:, None, :]
        out.append(sum3(v[:, :, 0] * a[:, :, 0], v[:, :, 1] * a[:, :, 1], v[:, :, 2] * a[:, :, 2]))
    return torch.stack(out, dim=2)


def look(vertices, eye, direction=None, up=None):
    assert vertices.ndim == 3
    bs = vertices.shape[0]
    direction = _f32([0, 0, 1] if direction is None else direction)
    up = _f32([0, 1, 0] if up is None else up)
    eye = _f32(eye)
    if eye.ndim == 1:
        eye = eye[None, :]
    if direction.ndim == 1:
        direction = direction[None, :]
    if up.ndim == 1:
        up = up[None, :]
    n = max(direction.shape[0], up.shape[0])
    direction = direction.expand(n, 3)
    up = up.expand(n, 3)
    z_axis = normalize(direction)
    x_axis = normalize(cross(up, z_axis))
    y_axis = normalize(cross(z_axis, x_axis))
    return _apply_rotation(vertices, eye.expand(bs, 3), x_axis.expand(bs, 3), y_axis.expand(bs, 3),
                           z_axis.expand(bs, 3))


def look_at(vertices, eye, at=None, up=None):
    assert vertices.ndim == 3
    bs = vertices.shape[0]
    at = _f32([0, 0, 0] if at is None else at)
    up = _f32([0, 1, 0] if up is None else up)
    eye = _f32(eye)
    if eye.ndim == 1:
        eye = eye[None, :].expand(bs, 3)
    if at.ndim == 1:
        at = at[None, :].expand(bs, 3)
    if up.ndim == 1:
        up = up[None, :].expand(bs, 3)
    z_axis = normalize(at - eye)
    x_axis = normalize(cross(up, z_axis))
    y_axis = normalize(cross(z_axis, x_axis))
    return _apply_rotation(vertices, eye, x_axis, y_axis, z_axis)


def perspective_width(angle):
    """tan(angle / 180. * 3.1416) in float32 on the host (perspective.py:10-13)."""
    a = np.float32(angle)
    a = np.float32(a / np.float32(180.))
    a = np.float32(a * np.float32(3.1416))
    return np.float32(np.tan(a, dtype=np.float32))


def perspective(vertices, angle=30.):
    assert vertices.ndim == 3
    if isinstance(angle, torch.Tensor):
        a = angle.to(torch.float32) / 180. * 3.1416
        width = torch.tan(a).reshape(-1, 1).expand(vertices.shape[0], 1)
    else:
        width = torch.full((vertices.shape[0], 1), float(perspective_width(angle)), dtype=torch.float32)
    z = vertices[:, :, 2]
    x = vertices[:, :, 0] / z / width
    y = vertices[:, :, 1] / z / width
    return torch.stack([x, y, z], dim=2)


def vertices_to_faces(vertices, faces):
    assert vertices.ndim == 3 and faces.ndim == 3
    assert vertices.shape[0] == faces.shape[0]
    assert vertices.shape[2] == 3 and faces.shape[2] == 3
    bs, nv = vertices.shape[:2]
    idx = faces.long() + (torch.arange(bs, dtype=torch.long) * nv)[:, None, None]
    return vertices.reshape(bs * nv, 3)[idx]


def lighting(faces, textures, intensity_ambient=0.5, intensity_directional=0.5, color_ambient=(1, 1, 1),
             color_directional=(1, 1, 1), direction=(0, 1, 0)):
    bs, nf = faces.shape[:2]
    color_ambient = _f32(color_ambient)
    color_directional = _f32(color_directional)
    direction = _f32(direction)
    if color_ambient.ndim == 1:
        color_ambient = color_ambient[None, :].expand(bs, 3)
    if color_directional.ndim == 1:
        color_directional = color_directional[None, :].expand(bs, 3)
    if direction.ndim == 1:
        direction = direction[None, :].expand(bs, 3)
    light = torch.zeros(bs, nf, 3, dtype=torch.float32)
    if intensity_ambient != 0:
        light = light + intensity_ambient * color_ambient[:, None, :]
    if intensity_directional != 0:
        f = faces.reshape(bs * nf, 3, 3)
        v10 = f[:, 0] - f[:, 1]
        v12 = f[:, 2] - f[:, 1]
        normals = normalize(cross(v10, v12)).reshape(bs, nf, 3)
        d = direction[:, None, :]
        cos = torch.relu(sum3(normals[:, :, 0] * d[:, :, 0], normals[:, :, 1] * d[:, :, 1],
                              normals[:, :, 2] * d[:, :, 2]))
        light = light + intensity_directional * (color_directional[:, None, :] * cos[:, :, None])
    return textures * light[:, :, None, None, None, :]


class Rasterize(torch.autograd.Function):
    """chainer Function `Rasterize` (rasterize.py:19-894) as a torch Function on CPU tensors."""

    @staticmethod
    def forward(ctx, faces, textures, image_size, near, far, eps, background_color, return_rgb, return_alpha,
                return_depth, unsafe, impl):
        st = raster_np.forward(
            faces.detach().numpy(), None if textures is None else textures.detach().numpy(), image_size, near, far,
            eps, background_color if background_color is not None else (0, 0, 0), return_rgb, return_alpha,
            return_depth, unsafe=unsafe, impl=impl)
        ctx.st = st
        ctx.has_textures = textures is not None
        rgb = torch.from_numpy(st.rgb_map) if return_rgb else None
        alpha = torch.from_numpy(st.alpha_map.copy()) if return_alpha else None
        depth = torch.from_numpy(st.depth_map.copy()) if return_depth else None
        return rgb, alpha, depth

    @staticmethod
    def backward(ctx, g_rgb, g_alpha, g_depth):
        st = ctx.st
        gf, gt = raster_np.backward(
            st, None if g_rgb is None else g_rgb.contiguous().numpy(),
            None if g_alpha is None else g_alpha.contiguous().numpy(),
            None if g_depth is None else g_depth.contiguous().numpy())
        gft = torch.from_numpy(gf)
        gtt = torch.from_numpy(gt) if (gt is not None and ctx.has_textures) else None
        return (gft, gtt) + (None,) * 10


def pool2x2(x):
    """average_pooling_2d(x, 2, 2) on [..., H, W]: ((a + b) + c) + d, then * 0.25."""
    s = ((x[..., 0::2, 0::2] + x[..., 0::2, 1::2]) + x[..., 1::2, 0::2]) + x[..., 1::2, 1::2]
    return s * 0.25


def rasterize_rgbad(faces, textures=None, image_size=DEFAULT_IMAGE_SIZE, anti_aliasing=DEFAULT_ANTI_ALIASING,
                    near=DEFAULT_NEAR, far=DEFAULT_FAR, eps=DEFAULT_EPS, background_color=DEFAULT_BACKGROUND_COLOR,
                    return_rgb=True, return_alpha=True, return_depth=True, unsafe=False, impl='oracle'):
    size = image_size * 2 if anti_aliasing else image_size
    rgb, alpha, depth = Rasterize.apply(faces, textures, size, near, far, eps, background_color, return_rgb,
                                        return_alpha, return_depth, unsafe, impl)
    if return_rgb:
        rgb = rgb.permute(0, 3, 1, 2).flip(2)
    if return_alpha:
        alpha = alpha.flip(1)
    if return_depth:
        depth = depth.flip(1)
    if anti_aliasing:
        if return_rgb:
            rgb = pool2x2(rgb)
        if return_alpha:
            alpha = pool2x2(alpha)
        if return_depth:
            depth = pool2x2(depth)
    return {'rgb': rgb if return_rgb else None, 'alpha': alpha if return_alpha else None,
            'depth': depth if return_depth else None}


def rasterize(faces, textures, image_size=DEFAULT_IMAGE_SIZE, anti_aliasing=DEFAULT_ANTI_ALIASING,
              near=DEFAULT_NEAR, far=DEFAULT_FAR, eps=DEFAULT_EPS, background_color=DEFAULT_BACKGROUND_COLOR, **kw):
    return rasterize_rgbad(faces, textures, image_size, anti_aliasing, near, far, eps, background_color, True, False,
                           False, **kw)['rgb']


def rasterize_silhouettes(faces, image_size=DEFAULT_IMAGE_SIZE, anti_aliasing=DEFAULT_ANTI_ALIASING,
                          near=DEFAULT_NEAR, far=DEFAULT_FAR, eps=DEFAULT_EPS, **kw):
    return rasterize_rgbad(faces, None, image_size, anti_aliasing, near, far, eps, None, False, True, False,
                           **kw)['alpha']


def rasterize_depth(faces, image_size=DEFAULT_IMAGE_SIZE, anti_aliasing=DEFAULT_ANTI_ALIASING, near=DEFAULT_NEAR,
                    far=DEFAULT_FAR, eps=DEFAULT_EPS, **kw):
    return rasterize_rgbad(faces, None, image_size, anti_aliasing, near, far, eps, None, False, False, True,
                           **kw)['depth']


class NRRenderer(object):
    """nr.Renderer (neural_renderer/renderer.py:11-110) plus the `_Renderer` overrides of
    derender3d/models/renderer.py:19-127 (`up` vector for camera_mode 'look', render_normal)."""

    def __init__(self):
        self.image_size = 256
        self.anti_aliasing = True
        self.background_color = [0, 0, 0]
        self.fill_back = True
        self.perspective = True
        self.viewing_angle = 30
        self.eye = [0, 0, -(1. / math.tan(math.radians(self.viewing_angle)) + 1)]
        self.camera_mode = 'look_at'
        self.camera_direction = [0, 0, 1]
        self.up = None
        self.near = 0.1
        self.far = 100
        self.light_intensity_ambient = 0.5
        self.light_intensity_directional = 0.5
        self.light_color_ambient = [1, 1, 1]
        self.light_color_directional = [1, 1, 1]
        self.light_direction = [0, 1, 0]
        self.rasterizer_eps = 1e-3
        self.raster_kw = {}

    def _fill_back(self, faces):
        return torch.cat((faces, faces.flip(2)), dim=1) if self.fill_back else faces

    def _camera(self, vertices):
        if self.camera_mode == 'look_at':
            vertices = look_at(vertices, self.eye)
        elif self.camera_mode == 'look':
            vertices = look(vertices, self.eye, self.camera_direction, self.up)
        if self.perspective:
            vertices = perspective(vertices, angle=self.viewing_angle)
        return vertices

    def render_silhouettes(self, vertices, faces):
        faces = self._fill_back(faces)
        vertices = self._camera(vertices)
        faces = vertices_to_faces(vertices, faces)
        # rasterize_silhouettes is called WITHOUT near/far/eps: module defaults (renderer.py:57)
        return rasterize_silhouettes(faces, self.image_size, self.anti_aliasing, **self.raster_kw)

    def render_depth(self, vertices, faces):
        faces = self._fill_back(faces)
        vertices = self._camera(vertices)
        faces = vertices_to_faces(vertices, faces)
        return rasterize_depth(faces, self.image_size, self.anti_aliasing, **self.raster_kw)

    def face_normals(self, vertices, faces_filled):
        fn = vertices_to_faces(vertices, faces_filled)
        bs, nf = fn.shape[:2]
        fn = fn.reshape(bs * nf, 3, 3)
        v10 = fn[:, 0] - fn[:, 1]
        v12 = fn[:, 2] - fn[:, 1]
        return normalize(cross(v10, v12)).reshape(bs, nf, 3)

    def render_normal(self, vertices, faces):
        faces = self._fill_back(faces)
        normals = self.face_normals(vertices, faces)
        textures = normals[:, :, None, None, None, :].repeat(1, 1, 2, 2, 2, 1)
        vertices = self._camera(vertices)
        faces = vertices_to_faces(vertices, faces)
        return rasterize(faces, textures, self.image_size, self.anti_aliasing, self.near, self.far,
                         self.rasterizer_eps, self.background_color, **self.raster_kw)

    def render(self, vertices, faces, textures):
        if self.fill_back:
            faces = torch.cat((faces, faces.flip(2)), dim=1)
            textures = torch.cat((textures, textures.permute(0, 1, 4, 3, 2, 5)), dim=1)
        faces_lighting = vertices_to_faces(vertices, faces)
        textures = lighting(faces_lighting, textures, self.light_intensity_ambient, self.light_intensity_directional,
                            self.light_color_ambient, self.light_color_directional, self.light_direction)
        vertices = self._camera(vertices)
        faces = vertices_to_faces(vertices, faces)
        return rasterize(faces, textures, self.image_size, self.anti_aliasing, self.near, self.far,
                         self.rasterizer_eps, self.background_color, **self.raster_kw)


class RenderType:
    RGB = 0
    Silhouette = 1
    Depth = 2
    Normal = 3


class SDNRenderer(object):
    """derender3d.models.renderer.Renderer (renderer.py:216-272) on CPU tensors."""

    def __init__(self, image_size=256, viewing_angle=30):
        self.image_size = image_size
        self.viewing_angle = viewing_angle
        self.eye = torch.tensor([0., 0., 0.])
        self.camera_mode = 'look'
        self.camera_direction = torch.tensor([0., 0., -1.])
        self.camera_up = torch.tensor([0., 1., 0.])
        self.raster_kw = {}

    def __call__(self, vertices, faces, textures=None, render_type=RenderType.RGB):
        r = NRRenderer()
        r.image_size = self.image_size
        r.viewing_angle = self.viewing_angle
        r.raster_kw = self.raster_kw
        vertices = vertices * torch.tensor([-1., 1., 1.])  # renderer.py:243
        bs = len(vertices)
        r.eye = self.eye[None, :].expand(bs, -1)
        r.camera_mode = self.camera_mode
        r.camera_direction = self.camera_direction[None, :].expand(bs, -1)
        r.up = self.camera_up[None, :].expand(bs, -1)
        if render_type == RenderType.RGB:
            images = r.render(vertices, faces, textures)
        elif render_type == RenderType.Silhouette:
            images = r.render_silhouettes(vertices, faces)[:, None]
        elif render_type == RenderType.Depth:
            images = r.render_depth(vertices, faces)[:, None]
        else:
            images = r.render_normal(vertices, faces)
            x, y, z = torch.unbind(images, dim=1)
            images = torch.stack([-x, y, z], dim=1)  # renderer.py:268-270
        return images
