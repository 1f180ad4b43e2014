"""nr.Renderer (neural_renderer/renderer.py:11-110): attribute bag + render_silhouettes / render_depth / render."""
import math

import torch

from sdn_hip import ops

from . import camera
from .rasterize import rasterize, rasterize_depth, rasterize_silhouettes
from .shading import lighting


class Renderer(object):
    def __init__(self):
        # rendering
        self.image_size = 256
        self.anti_aliasing = True
        self.background_color = [0, 0, 0]
        self.fill_back = True

        # camera
        self.perspective = True
        self.viewing_angle = 30
        self.eye = [0, 0, -(1. / math.tan(math.radians(self.viewing_angle)) + 1)]
        self.camera_mode = 'look_at'
        self.camera_direction = [0, 0, 1]
        self.near = 0.1
        self.far = 100

        # light
        self.light_intensity_ambient = 0.5
        self.light_intensity_directional = 0.5
        self.light_color_ambient = [1, 1, 1]  # white
        self.light_color_directional = [1, 1, 1]  # white
        self.light_direction = [0, 1, 0]  # up-to-down

        # rasterization
        self.rasterizer_eps = 1e-3

    # -- helpers shared by the three entry points -------------------------------------------------------------
    def _camera_up(self):
        return None

    def project(self, vertices):
        """viewpoint + perspective transformation as ONE kernel (renderer.py:43-51 are three chainer ops)."""
        bs, dev = vertices.shape[0], vertices.device
        mode = ops.CAMERA_NONE
        eye = direction = up = None
        if self.camera_mode == 'look_at':
            mode = ops.CAMERA_LOOK_AT
            eye = camera._vec(self.eye, bs, dev, None)
            direction = camera._vec(None, bs, dev, [0, 0, 0])
            up = camera._vec(None, bs, dev, [0, 1, 0])
        elif self.camera_mode == 'look':
            mode = ops.CAMERA_LOOK
            eye = camera._vec(self.eye, bs, dev, None)
            direction = camera._vec(self.camera_direction, bs, dev, [0, 0, 1])
            up = camera._vec(self._camera_up(), bs, dev, [0, 1, 0])
        width = camera.perspective_width(self.viewing_angle, bs, dev) if self.perspective else None
        if mode == ops.CAMERA_NONE and width is None:
            return vertices
        return ops.ProjectVertices.apply(vertices, mode, eye, direction, up, width, 0)

    def gather(self, vertices, faces):
        """vertices_to_faces with fill_back folded in (renderer.py:41,54)."""
        return ops.GatherFaces.apply(vertices, faces, bool(self.fill_back))

    # -- public API ------------------------------------------------------------------------------------------------
    def render_silhouettes(self, vertices, faces):
        faces = self.gather(self.project(vertices), faces)
        return rasterize_silhouettes(faces, self.image_size, self.anti_aliasing)

    def render_depth(self, vertices, faces):
        faces = self.gather(self.project(vertices), faces)
        return rasterize_depth(faces, self.image_size, self.anti_aliasing)

    def render(self, vertices, faces, textures):
        if self.fill_back:
            textures = torch.cat((textures, textures.permute(0, 1, 4, 3, 2, 5)), dim=1)
        faces_lighting = self.gather(vertices, faces)
        textures = lighting(
            faces_lighting,
            textures,
            self.light_intensity_ambient,
            self.light_intensity_directional,
            self.light_color_ambient,
            self.light_color_directional,
            self.light_direction)
        faces = self.gather(self.project(vertices), faces)
        return rasterize(
            faces, textures, self.image_size, self.anti_aliasing, self.near, self.far, self.rasterizer_eps,
            self.background_color)
