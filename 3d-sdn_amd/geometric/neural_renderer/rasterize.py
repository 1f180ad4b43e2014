"""Rasterization entry points (neural_renderer/rasterize.py:19-1062) on the MI355X tile rasterizer.

`rasterize_rgbad` and friends keep the reference's signatures and return shapes
(rgb [bs,3,is,is], alpha [bs,is,is], depth [bs,is,is]; vertically flipped, 2x2 averaged when
anti-aliasing).  Everything from face setup to the pooled maps is one C-ABI call
(sdn_rasterize_fwd, csrc/raster_fwd.hip); gradients come from sdn_rasterize_bwd.

Default semantics are those of the reference's deterministic ("safe") kernels K2 + K3.  `use_unsafe_rasterizer(True)` /
NEURAL_RENDERER_UNSAFE=1 (what the reference's scripts/env.sh:11 exports, rasterize.py:13-16) select the coverage rule and
barycentric arithmetic of its default kernel K1 (rasterize.py:102-236: pixel-space scanlines over the x-sorted vertices) --
since r04 as a deterministic HIP path (SDN_K1_COVERAGE, csrc/raster_fwd.hip: k_raster_tiles_k1): the one thing K1 leaves to
thread scheduling, which face wins an exact depth tie, goes to the lowest face index.
"""
import os

import torch

from sdn_hip import ops

DEFAULT_IMAGE_SIZE = 256
DEFAULT_ANTI_ALIASING = True
DEFAULT_NEAR = 0.1
DEFAULT_FAR = 100
DEFAULT_EPS = 1e-4
DEFAULT_BACKGROUND_COLOR = (0, 0, 0)
USE_UNSAFE_IMPLEMENTATION = False
if 'NEURAL_RENDERER_UNSAFE' in os.environ and int(os.environ['NEURAL_RENDERER_UNSAFE']):    # rasterize.py:15-16
    USE_UNSAFE_IMPLEMENTATION = True
    ops.set_k1_coverage(True)


def rasterize_rgbad(
        faces,
        textures=None,
        image_size=DEFAULT_IMAGE_SIZE,
        anti_aliasing=DEFAULT_ANTI_ALIASING,
        near=DEFAULT_NEAR,
        far=DEFAULT_FAR,
        eps=DEFAULT_EPS,
        background_color=DEFAULT_BACKGROUND_COLOR,
        return_rgb=True,
        return_alpha=True,
        return_depth=True,
):
    """
    Generate RGB, alpha channel, and depth images from faces and textures (for RGB).

    Args:
        faces (torch.Tensor): [batch size, number of faces, 3 (vertices), 3 (XYZ)], float32, CUDA.
        textures (torch.Tensor): [batch size, number of faces, ts, ts, ts, 3 (RGB)].
        image_size (int): Width and height of rendered images.
        anti_aliasing (bool): 2x super-sampling.
        near, far (float): depth range.   eps (float): epsilon of the approximate gradient.
        background_color (tuple | tensor): [3] or [bs, 3].

    Returns:
        dict: {'rgb': [bs, 3, is, is] | None, 'alpha': [bs, is, is] | None, 'depth': [bs, is, is] | None}
    """
    rgb, alpha, depth = ops.RasterizeMaps.apply(
        faces, textures, image_size, anti_aliasing, near, far, eps, background_color, return_rgb, return_alpha,
        return_depth, None, False)
    return {
        'rgb': rgb if return_rgb else None,
        'alpha': alpha if return_alpha else None,
        'depth': depth if return_depth else None,
    }


def rasterize(
        faces,
        textures,
        image_size=DEFAULT_IMAGE_SIZE,
        anti_aliasing=DEFAULT_ANTI_ALIASING,
        near=DEFAULT_NEAR,
        far=DEFAULT_FAR,
        eps=DEFAULT_EPS,
        background_color=DEFAULT_BACKGROUND_COLOR,
):
    """RGB images [batch size, 3, image_size, image_size] (rasterize.py:977-1005)."""
    return rasterize_rgbad(
        faces, textures, image_size, anti_aliasing, near, far, eps, background_color, True, False, False)['rgb']


def rasterize_silhouettes(
        faces,
        image_size=DEFAULT_IMAGE_SIZE,
        anti_aliasing=DEFAULT_ANTI_ALIASING,
        near=DEFAULT_NEAR,
        far=DEFAULT_FAR,
        eps=DEFAULT_EPS,
):
    """Alpha channels [batch size, image_size, image_size] (rasterize.py:1008-1031)."""
    return rasterize_rgbad(faces, None, image_size, anti_aliasing, near, far, eps, None, False, True, False)['alpha']


def rasterize_depth(
        faces,
        image_size=DEFAULT_IMAGE_SIZE,
        anti_aliasing=DEFAULT_ANTI_ALIASING,
        near=DEFAULT_NEAR,
        far=DEFAULT_FAR,
        eps=DEFAULT_EPS,
):
    """Depth images [batch size, image_size, image_size] (rasterize.py:1034-1057)."""
    return rasterize_rgbad(faces, None, image_size, anti_aliasing, near, far, eps, None, False, False, True)['depth']


class Rasterize(object):
    """Callable with the constructor of the chainer Function (rasterize.py:19-37).  Calling it returns the raw
    (un-flipped, un-pooled) maps the Function returned: rgb [bs,is,is,3], alpha [bs,is,is], depth [bs,is,is]."""

    def __init__(self, image_size, near, far, eps, background_color, return_rgb=False, return_alpha=False,
                 return_depth=False):
        if not any((return_rgb, return_alpha, return_depth)):
            # nothing to draw
            raise Exception
        self.image_size = image_size
        self.near = near
        self.far = far
        self.eps = eps
        self.background_color = background_color
        self.return_rgb = return_rgb
        self.return_alpha = return_alpha
        self.return_depth = return_depth

    def __call__(self, faces, textures=None):
        rgb, alpha, depth = ops.RasterizeMaps.apply(
            faces, textures, self.image_size, False, self.near, self.far, self.eps, self.background_color,
            self.return_rgb, self.return_alpha, self.return_depth, None, False)
        # undo the vertical flip / NCHW layout that rasterize_rgbad applies on top of the Function
        if rgb is not None:
            rgb = rgb.flip(2).permute(0, 2, 3, 1)
        if alpha is not None:
            alpha = alpha.flip(1)
        if depth is not None:
            depth = depth.flip(1)
        return rgb, alpha, depth


def use_unsafe_rasterizer(flag):
    """rasterize.py:1060-1062; here it selects the deterministic K1-coverage path for all later forward calls"""
    global USE_UNSAFE_IMPLEMENTATION
    USE_UNSAFE_IMPLEMENTATION = flag
    ops.set_k1_coverage(flag)
