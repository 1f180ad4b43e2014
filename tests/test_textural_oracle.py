"""CPU: oracle/textural_oracle.py must reproduce the vectors the REFERENCE's networks produced
(tests/golden/textural_golden.npz, made by tests/golden/make_textural_golden.py from /root/reference/textural/models/
networks.py): forward outputs and the gradients of the recorded scalar loss, for generator, encoder and multiscale
discriminator."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import textural_oracle as to  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden', 'textural_golden.npz')


def load(prefix):
    z = np.load(GOLD)
    pick = lambda kind: {k.split('/', 2)[2]: torch.from_numpy(z[k]) for k in z.files if k.startswith('%s/%s/' % (prefix, kind))}
    return pick('sd'), pick('in'), pick('out'), pick('grad'), pick('gin')


def leaf_params(sd):
    ps = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.endswith('.weight') or k.endswith('.bias')}
    full = dict(sd)
    full.update(ps)
    return full, ps


def check_grads(ps, grads, atol):
    for k, p in ps.items():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        scale = float(grads[k].abs().max()) + 1e-12
        assert float((g - grads[k]).abs().max()) <= atol * max(scale, 1.0), k


def test_generator_matches_reference():
    sd, inp, out, grads, gin = load('G')
    full, ps = leaf_params(sd)
    x = inp['x'].clone().requires_grad_(True)
    y = to.global_generator(full, x, 2, 2)
    assert float((y - out['y']).abs().max()) < 2e-6
    (y * inp['w']).sum().backward()
    assert float((x.grad - gin['x']).abs().max()) < 1e-4 * max(1.0, float(gin['x'].abs().max()))
    check_grads(ps, grads, 2e-4)


def test_encoder_matches_reference():
    sd, inp, out, grads, gin = load('E')
    full, ps = leaf_params(sd)
    x = inp['x'].clone().requires_grad_(True)
    y = to.encoder(full, x, inp['inst'], 2)
    assert float((y - out['y']).abs().max()) < 2e-6
    (y * inp['w']).sum().backward()
    assert float((x.grad - gin['x']).abs().max()) < 1e-4 * max(1.0, float(gin['x'].abs().max()))
    check_grads(ps, grads, 2e-4)


def test_discriminator_matches_reference():
    sd, inp, out, grads, gin = load('D')
    full, ps = leaf_params(sd)
    x = inp['x'].clone().requires_grad_(True)
    res = to.multiscale_discriminator(full, x, 2, 3)
    loss = 0
    for s, scale in enumerate(res):
        assert len(scale) == 5
        for j, f in enumerate(scale):
            assert float((f - out['f%d_%d' % (s, j)]).abs().max()) < 5e-6, (s, j)
            loss = loss + (f * inp['w%d_%d' % (s, j)]).sum()
    loss.backward()
    assert float((x.grad - gin['x']).abs().max()) < 1e-4 * max(1.0, float(gin['x'].abs().max()))
    check_grads(ps, grads, 2e-4)


def test_three_scale_discriminator_matches_reference():
    """`--num_D 3` (BASELINE configs[3]): tests/golden/textural_d3_golden.npz, written by the reference's own
    MultiscaleDiscriminator (make_textural_golden_d3.py) -- all 15 feature maps and every gradient."""
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'textural_d3_golden.npz'))
    pick = lambda kind: {k.split('/', 2)[2]: torch.from_numpy(z[k]) for k in z.files if k.startswith('D3/%s/' % kind)}
    sd, inp, out, grads, gin = pick('sd'), pick('in'), pick('out'), pick('grad'), pick('gin')
    full, ps = leaf_params(sd)
    x = inp['x'].clone().requires_grad_(True)
    res = to.multiscale_discriminator(full, x, 3, 3)
    loss = 0
    assert len(res) == 3
    for s, scale in enumerate(res):
        assert len(scale) == 5
        for j, f in enumerate(scale):
            assert float((f - out['f%d_%d' % (s, j)]).abs().max()) < 5e-6, (s, j)
            loss = loss + (f * inp['w%d_%d' % (s, j)]).sum()
    loss.backward()
    assert float((x.grad - gin['x']).abs().max()) < 1e-4 * max(1.0, float(gin['x'].abs().max()))
    check_grads(ps, grads, 2e-4)


def test_float64_oracle_agrees_with_float32_reference():
    """The fp64 evaluation used as the GPU yardstick stays within fp32 round-off of the reference's fp32 result."""
    sd, inp, out, _, _ = load('G')
    y = to.global_generator(sd, inp['x'].double(), 2, 2)
    assert float((y.float() - out['y']).abs().max()) < 5e-6


def test_local_enhancer_surface_matches_the_reference_module():
    """define_G(..., 'local', ...) (networks.py:156-206): the product module carries exactly the reference's state_dict
    (keys in order, shapes) -- tests/golden/textural_local_golden.npz was written by the reference's own LocalEnhancer;
    the numbers are compared on the GPU (tests/test_gpu_textural.py::test_local_enhancer_against_reference_golden)."""
    import os
    import sys
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tex = os.path.join(root, '3d-sdn_amd', 'textural')
    if tex not in sys.path:
        sys.path.insert(0, tex)
    from models import networks as N
    z = np.load(os.path.join(root, 'tests', 'golden', 'textural_local_golden.npz'))
    ref = {k.split('/', 2)[2]: z[k] for k in z.files if k.startswith('L/sd/')}
    L = N.define_G(6, 3, 4, 'local', n_downsample_global=2, n_blocks_global=2, n_local_enhancers=1, n_blocks_local=2)
    sd = L.state_dict()
    assert list(sd.keys()) == list(ref.keys())
    assert all(tuple(sd[k].shape) == tuple(ref[k].shape) for k in sd)


def test_vgg_loss_restatement_reproduces_the_reference_classes():
    """tests/golden/vgg_golden.npz = the reference's own VGGLoss / Vgg19 (textural/models/networks.py:137-149, 467-497) run in fp64
    on a seeded torchvision-layout VGG19 (tests/golden/make_vgg_golden.py).  The restatement (oracle/textural_oracle.vgg_loss:
    slices relu1_1 ... relu5_1, weights 1/32 ... 1, detached target branch, L1 mean) reproduces loss, gradient and features."""
    import numpy as np
    from oracle import textural_oracle as to
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'vgg_golden.npz'))
    sd = to.vgg19_seeded_state(int(z['seed']))
    assert abs(sum(float(v.double().abs().sum()) for v in sd.values()) - float(z['weight_checksum'])) <= 1e-9 * float(z['weight_checksum'])
    sd64 = {k: v.double() for k, v in sd.items()}
    x = torch.from_numpy(z['x']).requires_grad_(True)
    loss, feats = to.vgg_loss(sd64, x, torch.from_numpy(z['y']))
    loss.backward()
    assert abs(float(loss) - float(z['loss'])) <= 1e-12 * abs(float(z['loss']))
    g = torch.from_numpy(z['grad_x'])
    assert float((x.grad - g).norm() / g.norm()) <= 1e-12
    for i, f in enumerate(feats):
        assert tuple(f.shape) == tuple(z['feat%d_shape' % i])
        s = torch.from_numpy(z['feat%d_sample' % i])
        assert float((f.detach().reshape(-1)[::97] - s).abs().max()) <= 1e-12 * max(float(s.abs().max()), 1.0)
