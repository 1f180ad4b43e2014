#!/usr/bin/env python3
"""Generate tests/golden/derender_golden.npz by running the REFERENCE's own derender3d code on the CPU.

Runs only where /root/reference exists (the build container).  Nothing from the reference is copied: its
modules are imported from where they lie, with three harmless shims so that they import without a GPU:
  * `torch.Tensor.cuda` / `torch.nn.Module.cuda` become no-ops (the reference hard-codes `.cuda()`,
    derender3d/models/transforms.py:40,97 and derender3d/models/__init__.py:110-123,169);
  * `neural_renderer`, `chainer`, `torchvision` are empty stub modules (only imported, never called here);
  * `Derenderer3d.render` is driven on a bare instance whose `renderer` is a recorder: it returns zero images and
    records the vertices / viewing angle it was asked to render, which is exactly the geometry decode we pin.
Golden content: FFD decode, PerspectiveTransform (train and test variants), and every pose tensor of
Derenderer3d.render (eval mode) for a fixed seed.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference/geometric'
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.current_device = lambda: 0
    for name in ('chainer', 'chainer.functions', 'neural_renderer', 'torchvision'):
        sys.modules[name] = types.ModuleType(name)
    sys.modules['chainer'].functions = sys.modules['chainer.functions']
    sys.modules['chainer'].Function = object
    sys.modules['neural_renderer'].Renderer = object
    sys.path.insert(0, REF)
    import derender3d.models as ref_models  # noqa: E402  (the reference package)
    from derender3d.models.transforms import FFD, PerspectiveTransform  # noqa: E402

    rng = np.random.default_rng(20260925)
    torch.manual_seed(20260925)
    out = {}

    # two small templates inside the unit box (like ShapenetObj vertices)
    templates = []
    for k, nv in enumerate((150, 230)):
        v = rng.uniform(-0.5, 0.5, (nv, 3)).astype(np.float32)
        f = rng.integers(0, nv, (2 * nv, 3)).astype(np.int32)
        templates.append((v, f))
        out['template%d_vertices' % k] = v
        out['template%d_faces' % k] = f

    constraints = [FFD.Constraint.symmetry(axis=FFD.Constraint.Axis.z),
                   FFD.Constraint.homogeneity(axis=FFD.Constraint.Axis.y, index=[0, 1])]
    ffds = [FFD(torch.tensor(v), constraints=constraints) for v, _ in templates]
    for k, ffd in enumerate(ffds):
        coeff = torch.tensor(rng.normal(0, 0.05, 192).astype(np.float32))
        out['ffd%d_coeff' % k] = coeff.numpy()
        out['ffd%d_vertices' % k] = ffd(coeff).numpy()
        out['ffd%d_B' % k] = ffd.B.numpy()
        out['ffd%d_P0' % k] = ffd.P0.numpy()

    # PerspectiveTransform, both call forms
    pt = PerspectiveTransform()
    verts = torch.tensor(rng.uniform(-0.5, 0.5, (2, 40, 3)).astype(np.float32))
    scales = torch.tensor(rng.uniform(1, 4, (2, 3)).astype(np.float32))
    th = rng.uniform(-3, 3, 2)
    rot = torch.tensor(np.stack([np.cos(th / 2), 0 * th, np.sin(th / 2), 0 * th], 1).astype(np.float32))
    trans = torch.tensor(np.array([[2.0, 1.0, -12.0], [-3.0, 0.5, -20.0]], np.float32))
    ptrans = torch.tensor(np.array([[1.8, 0.9, -11.0], [-2.5, 0.4, -19.0]], np.float32))
    zooms = torch.tensor(np.array([[0.7], [1.3]], np.float32))
    zoom_tos = torch.tensor(np.array([[384 / (2 * 725.0)], [384 / (2 * 725.0)]], np.float32))
    out['pt_vertices'], out['pt_scales'], out['pt_rotations'] = verts.numpy(), scales.numpy(), rot.numpy()
    out['pt_translations'], out['pt_ptranslations'] = trans.numpy(), ptrans.numpy()
    out['pt_zooms'], out['pt_zoom_tos'] = zooms.numpy(), zoom_tos.numpy()
    out['pt_train_out'] = pt(verts, scales=scales, rotations=rot, translations=trans,
                             perspective_translations=ptrans, zooms=zooms).numpy()
    v2, z2 = pt(verts, scales=scales, rotations=rot, translations=trans, perspective_translations=trans,
                zoom_tos=zoom_tos)
    out['pt_test_out'], out['pt_test_zooms'] = v2.numpy(), z2.numpy()

    # Derenderer3d.render (eval mode) on a bare instance with a recording renderer
    class Recorder(object):
        def __init__(self):
            self.calls = []
            self.viewing_angle = None

        def __call__(self, vertices, faces, render_type=None):
            self.calls.append((vertices.detach().numpy().copy(), float(self.viewing_angle), int(render_type)))
            c = 3 if render_type == 3 else 1
            return torch.zeros(1, c, 8, 8)

    class Obj(object):
        pass

    m = object.__new__(ref_models.Derenderer3d)
    torch.nn.Module.__init__(m)
    m.training = False
    m._force_no_sample = False
    m.mode = 15  # TargetType.extend
    m.image_size = 256
    m.render_size = 384
    m.objs = []
    for v, f in templates:
        o = Obj()
        o.vertices, o.faces = torch.tensor(v), torch.tensor(f)
        m.objs.append(o)
    object.__setattr__(m, 'ffds', ffds)
    m.perspective_transform = pt
    object.__setattr__(m, 'renderer', Recorder())
    n = 5
    blob = {
        '_mroi_norms': torch.tensor(rng.uniform(-0.2, 0.2, (n, 2)).astype(np.float32)),
        '_droi_norms': torch.tensor(rng.uniform(0.06, 0.3, (n, 2)).astype(np.float32)),
        '_focals': torch.full((n, 1), 725.0),
        '_theta_deltas': torch.nn.functional.normalize(torch.tensor(rng.normal(size=(n, 2)).astype(np.float32)), dim=1),
        '_translation2ds': torch.tensor(rng.normal(0, 0.1, (n, 2)).astype(np.float32)),
        '_log_scales': torch.tensor(rng.normal(0.8, 0.2, (n, 3)).astype(np.float32)),
        '_log_depths': torch.tensor(rng.normal(1.0, 0.3, (n, 1)).astype(np.float32)),
        '_class_probs': torch.softmax(torch.tensor(rng.normal(size=(n, 2)).astype(np.float32)), dim=1),
        '_ffd_coeffs': torch.tensor(rng.normal(0, 0.03, (n, 2, 192)).astype(np.float32)),
    }
    for k, v in blob.items():
        out['blob' + k] = v.numpy()
    devnull = open(os.devnull, 'w')
    stdout, sys.stdout = sys.stdout, devnull  # the reference prints the device id per object
    try:
        res = m.render(blob)
    finally:
        sys.stdout = stdout
    for k in ('_thetas', '_alphas', '_rotations', '_scales', '_depths', '_center2ds', '_translations',
              '_class_log_probs', '_zooms'):
        out['render' + k] = res[k].numpy()
    calls = m.renderer.calls
    assert len(calls) == 3 * n
    out['render_vertices'] = np.stack([np.pad(c[0][0], ((0, 230 - c[0].shape[1]), (0, 0))) for c in calls[0::3]])
    out['render_nverts'] = np.asarray([c[0].shape[1] for c in calls[0::3]], np.int32)
    out['render_viewing_angles'] = np.asarray([c[1] for c in calls[0::3]], np.float64)
    np.savez_compressed(os.path.join(HERE, 'derender_golden.npz'), **out)
    print('wrote derender_golden.npz with %d arrays' % len(out))


if __name__ == '__main__':
    main()
