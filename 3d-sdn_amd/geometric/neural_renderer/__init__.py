"""neural_renderer -- torch-native, MI355X-backed drop-in for the Chainer package of the same name.

Public names and signatures follow /root/reference/geometric/neural_renderer/__init__.py:2-17; arrays
are torch CUDA tensors instead of chainer Variables / cupy arrays, and every op runs on hand-written
HIP kernels through libsdn_hip.so (3d-sdn_amd/csrc).  Put `3d-sdn_amd/geometric` on PYTHONPATH the
way the reference's scripts/env.sh puts `geometric/` there.
"""
from .shading import cross, lighting
from .camera import get_points_from_angles, look, look_at, perspective
from .mesh_io import load_obj, save_obj, Mesh
from .optimizers import Adam
from .rasterize import (
    rasterize_rgbad, rasterize, rasterize_silhouettes, rasterize_depth, use_unsafe_rasterizer, Rasterize)
from .renderer import Renderer
from .vertices_to_faces import vertices_to_faces

__version__ = '1.1.3'
