"""Derenderer3d: encoder -> pose algebra -> per-object FFD decode, perspective transform and render.

Reference: /root/reference/geometric/derender3d/models/__init__.py:18-250 (same constructor, same `forward` /
`render` signatures, same blob keys).  What changes on MI355X:
  * all constants are created on the blob's device (the reference hard-codes `.cuda()`, :110-123,169);
  * each object is rasterized ONCE for silhouette + normal + depth (`Renderer.render_maps`) instead of three
    times (:203-224), with no host round trip (reference: derender3d/models/renderer.py:131-150);
  * mesh templates may be passed in (`objs=`) because two of the eight ShapeNet models the reference hard-codes
    are not shipped with it; the default still reads SHAPENET_ROOT_DIR like the reference.
"""
import numpy as np
import os
import torch

from torch.distributions import Categorical
from torch.nn.modules import Module

import neural_renderer as nr

from derender3d import TargetType
from derender3d.models.derenderer import Derenderer
from derender3d.models.renderer import RenderType, Renderer
from derender3d.models.transforms import FFD, FFDBank, PerspectiveTransform


class ShapenetObj(object):
    root_dir = os.getenv('SHAPENET_ROOT_DIR')

    def __init__(self,
                 class_id=None,
                 obj_id=None,
                 root_dir=None,
                 vertices=None,
                 faces=None):
        if vertices is None:
            root_dir = root_dir if root_dir is not None else ShapenetObj.root_dir
            path = os.path.join(root_dir, class_id, obj_id, 'models', 'model_normalized.obj')
            print('Reading {:s}'.format(path))
            (vertices, faces) = nr.load_obj(path)
            # unit extent per axis, then the (z, y, -x) axis convention of the reference (:29-31)
            vertices = vertices / np.ptp(vertices, axis=0)
            vertices = vertices[:, [2, 1, 0]] * np.asarray([-1, 1, 1], dtype=np.float32)
        self.vertices = torch.as_tensor(np.asarray(vertices), dtype=torch.float32)
        self.faces = torch.as_tensor(np.asarray(faces), dtype=torch.int32)


DEFAULT_OBJS = [
    ('02958343', '137f67657cdc9da5f985cd98f7d73e9a'),
    ('02958343', '5343e944a7753108aa69dfdc5532bb13'),
    ('02958343', '3776e4d1e2587fd3253c03b7df20edd5'),
    ('02958343', '3ba5bce1b29f0be725f689444c7effe2'),
    ('02958343', '53a031dd120e81dc3aa562f24645e326'),
    ('02924116', '7905d83af08a0ca6dafc1d33c05cbcf8'),
    ('02958343', 'a0fe4aac120d5f8a5145cad7315443b3'),
    ('02958343', 'cd7feedd6041209131ac5fb37e6c8324'),
]


class Derenderer3d(Module):
    def __init__(self, mode, image_size, render_size, objs=None):
        super(Derenderer3d, self).__init__()

        self.mode = mode
        self.image_size = image_size
        self.render_size = render_size

        self.derenderer = Derenderer()

        self._force_no_sample = False

        if mode & TargetType.reproject:
            if objs is None:
                objs = [ShapenetObj(class_id=c, obj_id=o) for (c, o) in DEFAULT_OBJS]
            self.objs = objs
            self.ffds = torch.nn.ModuleList([FFD(obj.vertices, constraints=[
                FFD.Constraint.symmetry(axis=FFD.Constraint.Axis.z),
                FFD.Constraint.homogeneity(axis=FFD.Constraint.Axis.y, index=[0, 1]),
            ]) for obj in self.objs])
            self.perspective_transform = PerspectiveTransform()
            self.renderer = Renderer(image_size=render_size)
            self._faces_on = {}
            self._bank = None
            # True: decode + render all objects of a call in one batch of launches (same results as the per-object
            # loop of the reference, derender3d/models/__init__.py:161-224)
            self.batched = True

    def bank(self):
        if self._bank is None:
            bank = FFDBank(list(self.ffds), [obj.faces for obj in self.objs])
            object.__setattr__(self, '_bank', bank)
        dev = next(self.derenderer.parameters()).device
        if self._bank.Bt.device != dev:
            self._bank.to(dev)
        return self._bank

    def _faces(self, index, device):
        key = (index, device)
        f = self._faces_on.get(key)
        if f is None:
            f = self.objs[index].faces.to(device).unsqueeze(dim=0).contiguous()
            self._faces_on[key] = f
        return f

    def forward(self, images, roi_norms, focals):
        _mroi_norms = torch.stack([
            roi_norms[:, 2] + roi_norms[:, 0],
            roi_norms[:, 3] + roi_norms[:, 1],
        ], dim=1) / 2.0
        _droi_norms = torch.stack([
            roi_norms[:, 2] - roi_norms[:, 0],
            roi_norms[:, 3] - roi_norms[:, 1],
        ], dim=1)

        _blob = {
            '_roi_norms': roi_norms,
            '_mroi_norms': _mroi_norms,
            '_droi_norms': _droi_norms,
            '_focals': focals,
        }

        _blob.update(self.derenderer(images, _mroi_norms, _droi_norms))

        if not (self.mode & TargetType.reproject):
            return _blob

        _blob.update(self.render(_blob))
        return _blob

    def render(self, blob):
        _mroi_norms = blob['_mroi_norms']
        _droi_norms = blob['_droi_norms']
        _focals = blob['_focals']
        _theta_deltas = blob['_theta_deltas']
        _translation2ds = blob['_translation2ds']
        _log_scales = blob['_log_scales']
        _log_depths = blob['_log_depths']
        _class_probs = blob['_class_probs']
        _ffd_coeffs = blob['_ffd_coeffs']

        batch_size = len(_focals)
        dev = _theta_deltas.device
        zeros = torch.zeros(batch_size, 1, device=dev)
        ones = torch.ones(batch_size, device=dev)

        # yaw angle -> quaternion about +y (:107-113)
        _thetas = torch.unsqueeze(torch.atan2(_theta_deltas[:, 1], _theta_deltas[:, 0]), dim=1)
        _rotations = torch.cat([
            torch.cos(_thetas / 2),
            zeros,
            torch.sin(_thetas / 2),
            zeros,
        ], dim=1)
        _areas = torch.unsqueeze(_droi_norms[:, 0] * _droi_norms[:, 1], dim=1)

        _scales = torch.exp(_log_scales)
        _depths = torch.sqrt(torch.exp(_log_depths) / _areas)

        # object centre ray; the camera looks down -z with +y up (:119-126)
        _center2ds = _mroi_norms + _translation2ds * _droi_norms
        _translation_units = torch.stack([
            _center2ds[:, 1],
            - _center2ds[:, 0],
            - ones,
        ], dim=1)
        _translation_units = _translation_units / torch.norm(_translation_units, p=2, dim=1, keepdim=True)
        _translations = _depths * _translation_units

        _alphas = - (_thetas - torch.atan(_translations[:, 0:1] / _translations[:, 2:3]))
        _alphas = torch.remainder(_alphas + np.pi, 2 * np.pi) - np.pi

        if self.training and not self._force_no_sample:
            # REINFORCE over the mesh class (:131-134)
            _class_dists = Categorical(_class_probs)
            _class_samples = _class_dists.sample()
            _class_log_probs = _class_dists.log_prob(_class_samples)
        else:
            (_class_max_probs, _class_samples) = torch.max(_class_probs, dim=1)
            _class_log_probs = torch.log(_class_max_probs)

        if self.training:
            _perspective_translation_units = torch.stack([
                _mroi_norms[:, 1],
                - _mroi_norms[:, 0],
                - ones,
            ], dim=1)
            _perspective_translation_units = _perspective_translation_units / torch.norm(
                _perspective_translation_units, p=2, dim=1, keepdim=True)
            _perspective_translations = _depths * _perspective_translation_units
            _zooms = (self.image_size / _focals) / torch.max(_droi_norms, dim=1, keepdim=True)[0]
        else:
            _zoom_tos = self.render_size / (2.0 * _focals)
            _zooms = []

        want_normal = bool(self.mode & TargetType.normal)
        want_depth = bool(self.mode & TargetType.depth)

        if self.batched and type(self.renderer) is Renderer:
            return self._render_batched(blob, locals())

        # one host read for the whole batch instead of int(tensor) / .item() per object (:165,202)
        class_ids = _class_samples.tolist()
        focal_list = _focals.reshape(batch_size, -1)[:, 0].tolist()

        _masks = []
        _normals = []
        _depth_maps = []
        for size in range(batch_size):
            _class_sample = int(class_ids[size])
            _ffd_coeff = _ffd_coeffs[size][_class_sample]

            vertices = self.ffds[_class_sample](_ffd_coeff)
            __vertices = vertices.unsqueeze(dim=0)
            __faces = self._faces(_class_sample, dev)

            __scales = _scales[size].unsqueeze(dim=0)
            __rotations = _rotations[size].unsqueeze(dim=0)
            __translations = _translations[size].unsqueeze(dim=0)

            if self.training:
                __vertices = self.perspective_transform(
                    __vertices,
                    scales=__scales,
                    rotations=__rotations,
                    translations=__translations,
                    perspective_translations=_perspective_translations[size].unsqueeze(dim=0),
                    zooms=_zooms[size].unsqueeze(dim=0),
                )
            else:
                (__vertices, __zooms) = self.perspective_transform(
                    __vertices,
                    scales=__scales,
                    rotations=__rotations,
                    translations=__translations,
                    perspective_translations=__translations,
                    zoom_tos=_zoom_tos[size].unsqueeze(dim=0),
                )
                _zooms.append(__zooms)

            self.renderer.viewing_angle = np.arctan(self.render_size / (2.0 * focal_list[size])) / np.pi * 180
            (__masks, __normals, __depth_maps) = self.renderer.render_maps(
                __vertices, __faces, normal=want_normal, depth=want_depth)
            _masks.append(__masks)
            if want_normal:
                _normals.append(__normals)
            if want_depth:
                _depth_maps.append(__depth_maps)

        if not self.training:
            _zooms = torch.cat(_zooms, dim=0)

        _masks = torch.cat(_masks, dim=0)

        if want_normal:
            _normals = torch.cat(_normals, dim=0)

        if want_depth:
            _depth_maps = torch.cat(_depth_maps, dim=0)

        return self._pack(locals())

    def _render_batched(self, blob, L):
        """Same math as the loop below it, for all objects at once: one FFD decode launch, one batched
        PerspectiveTransform, one rasterization launch set."""
        batch_size = len(L['_focals'])
        dev = L['dev']
        bank = self.bank()
        classes = L['_class_samples']
        coeffs = L['_ffd_coeffs'][torch.arange(batch_size, device=dev), classes]
        vertices, faces = bank.decode(coeffs, classes)
        if self.training:
            vertices = self.perspective_transform(
                vertices, scales=L['_scales'], rotations=L['_rotations'], translations=L['_translations'],
                perspective_translations=L['_perspective_translations'], zooms=L['_zooms'])
            _zooms = L['_zooms']
        else:
            (vertices, _zooms) = self.perspective_transform(
                vertices, scales=L['_scales'], rotations=L['_rotations'], translations=L['_translations'],
                perspective_translations=L['_translations'], zoom_tos=L['_zoom_tos'])
        # per-object viewing angle, computed like np.arctan(render_size / (2 f)) / pi * 180 (:202) in float64
        focal_list = L['_focals'].reshape(batch_size, -1)[:, 0].tolist()  # one host read, like .item() in the loop
        self.renderer.viewing_angle = [np.arctan(self.render_size / (2.0 * f)) / np.pi * 180 for f in focal_list]
        (_masks, _normals, _depth_maps) = self.renderer.render_maps(
            vertices, faces, normal=L['want_normal'], depth=L['want_depth'])
        L = dict(L)
        L.update(_zooms=_zooms, _masks=_masks, _normals=_normals if L['want_normal'] else [],
                 _depth_maps=_depth_maps if L['want_depth'] else [])
        return self._pack(L)

    @staticmethod
    def _pack(L):
        (_thetas, _alphas, _rotations, _scales, _depths, _center2ds, _translations, _class_log_probs, _zooms, _masks,
         _normals, _depth_maps) = [L[k] for k in (
             '_thetas', '_alphas', '_rotations', '_scales', '_depths', '_center2ds', '_translations',
             '_class_log_probs', '_zooms', '_masks', '_normals', '_depth_maps')]
        return {
            '_thetas': _thetas,
            '_alphas': _alphas,
            '_rotations': _rotations,
            '_scales': _scales,
            '_depths': _depths,
            '_center2ds': _center2ds,
            '_translations': _translations,
            '_class_log_probs': _class_log_probs,
            '_zooms': _zooms,
            '_masks': _masks,
            '_normals': _normals,
            '_depth_maps': _depth_maps,
        }
