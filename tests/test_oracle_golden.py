"""Oracle and CPU-runnable host logic against the committed golden fixtures.

tests/golden/raster_golden.npz     outputs of the reference's rasterizer kernels (make_raster_golden.py)
tests/golden/derender_golden.npz   outputs of the reference's derender3d torch code (make_derender_golden.py)
"""
import os

import numpy as np
import pytest
import torch

from oracle import raster_np as rn
from util import biteq

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
RASTER = np.load(os.path.join(GOLD, 'raster_golden.npz'))
DEREN = np.load(os.path.join(GOLD, 'derender_golden.npz'))
CASES = ['soup_small', 'soup_mid', 'slivers', 'cube']


@pytest.mark.parametrize('name', CASES)
def test_oracle_matches_reference_golden(name):
    g = lambda k: RASTER[name + '/' + k]
    st = rn.forward(g('faces'), g('textures'), int(g('image_size')), 0.1, 100, 1e-3, (0.1, 0.2, 0.3), True, True,
                    True, impl='oracle')
    for k in ('face_index_map', 'weight_map', 'depth_map', 'face_inv_map', 'rgb_map', 'alpha_map',
              'sampling_index_map', 'sampling_weight_map'):
        assert biteq(getattr(st, k), g(k)), k
    gf, gt = rn.backward(st, g('g_rgb'), g('g_alpha'), g('g_depth'))
    assert biteq(gf, g('grad_faces'))
    assert biteq(gt, g('grad_textures'))
    st_a = rn.forward(g('faces'), None, int(g('image_size')), 0.1, 100, 1e-4, None, False, True, False, impl='oracle')
    assert biteq(rn.backward(st_a, None, g('g_alpha'), None)[0], g('grad_faces_alpha_only'))


def _ffds():
    from derender3d.models.transforms import FFD
    cons = [FFD.Constraint.symmetry(axis=FFD.Constraint.Axis.z),
            FFD.Constraint.homogeneity(axis=FFD.Constraint.Axis.y, index=[0, 1])]
    return [FFD(torch.tensor(DEREN['template%d_vertices' % k]), constraints=cons) for k in range(2)]


def test_ffd_matches_reference():
    for k, ffd in enumerate(_ffds()):
        np.testing.assert_allclose(ffd.B.numpy(), DEREN['ffd%d_B' % k], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(ffd.P0.numpy(), DEREN['ffd%d_P0' % k], rtol=0, atol=0)
        v = ffd(torch.tensor(DEREN['ffd%d_coeff' % k])).numpy()
        np.testing.assert_allclose(v, DEREN['ffd%d_vertices' % k], rtol=1e-5, atol=2e-6)


def test_ffd_gradient_matches_dense_formula():
    ffd = _ffds()[0]
    c = torch.tensor(DEREN['ffd0_coeff'], requires_grad=True)
    w = torch.linspace(-1, 1, ffd.B.shape[0] * 3).reshape(-1, 3)
    (ffd(c) * w).sum().backward()
    # reference formulation: sum over the [V,3,n,n,n] product
    c2 = torch.tensor(DEREN['ffd0_coeff'], requires_grad=True)
    dP = ffd.constrain(c2)
    V = ((ffd.P0 + dP) * ffd.B).view(-1, 3, 64).sum(dim=2)
    (V * w).sum().backward()
    np.testing.assert_allclose(c.grad.numpy(), c2.grad.numpy(), rtol=1e-4, atol=1e-5)


def test_perspective_transform_matches_reference():
    from derender3d.models.transforms import PerspectiveTransform
    t = lambda k: torch.tensor(DEREN[k])
    pt = PerspectiveTransform()
    out = pt(t('pt_vertices'), scales=t('pt_scales'), rotations=t('pt_rotations'), translations=t('pt_translations'),
             perspective_translations=t('pt_ptranslations'), zooms=t('pt_zooms'))
    np.testing.assert_allclose(out.numpy(), DEREN['pt_train_out'], rtol=1e-6, atol=1e-6)
    v2, z2 = pt(t('pt_vertices'), scales=t('pt_scales'), rotations=t('pt_rotations'),
                translations=t('pt_translations'), perspective_translations=t('pt_translations'),
                zoom_tos=t('pt_zoom_tos'))
    np.testing.assert_allclose(v2.numpy(), DEREN['pt_test_out'], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(z2.numpy(), DEREN['pt_test_zooms'], rtol=1e-6, atol=0)


def test_derenderer3d_pose_math_matches_reference():
    """Derenderer3d.render with a recording renderer: every pose tensor and the vertices handed to the renderer
    equal what the reference's render() produced for the same blob (eval mode)."""
    from derender3d import TargetType
    from derender3d.models import Derenderer3d, ShapenetObj

    objs = [ShapenetObj(vertices=DEREN['template%d_vertices' % k], faces=DEREN['template%d_faces' % k])
            for k in range(2)]
    m = Derenderer3d(mode=TargetType.extend, image_size=256, render_size=384, objs=objs)
    m.eval()

    calls = []

    class Recorder(object):
        viewing_angle = None

        def render_maps(self, vertices, faces, normal=True, depth=True):
            calls.append((vertices.detach().numpy().copy(), float(self.viewing_angle), tuple(faces.shape)))
            z = torch.zeros(1, 1, 8, 8)
            return z, torch.zeros(1, 3, 8, 8), z

    object.__setattr__(m, 'renderer', Recorder())
    blob = {k[4:]: torch.tensor(DEREN[k]) for k in DEREN.files if k.startswith('blob_')}
    res = m.render(blob)
    for k in ('_thetas', '_alphas', '_rotations', '_scales', '_depths', '_center2ds', '_translations',
              '_class_log_probs', '_zooms'):
        np.testing.assert_allclose(res[k].numpy(), DEREN['render' + k], rtol=2e-6, atol=2e-6, err_msg=k)
    assert len(calls) == len(DEREN['render_nverts'])
    for i, (v, ang, fshape) in enumerate(calls):
        nv = int(DEREN['render_nverts'][i])
        assert v.shape == (1, nv, 3)
        # vertices after zoom-to-fit: z is ~O(10..100); compare relatively
        np.testing.assert_allclose(v[0], DEREN['render_vertices'][i, :nv], rtol=2e-5, atol=2e-5)
        assert abs(ang - DEREN['render_viewing_angles'][i]) < 1e-9
