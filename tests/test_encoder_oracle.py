"""The encoder oracle (oracle/encoder_oracle.py) against tests/golden/encoder_golden.npz -- outputs and gradients of the
REFERENCE's Derenderer class (derenderer.py:7-65) hosted on the restated torchvision ResNet-18 -- and the seeded-weight
contract the GPU test relies on (the product Derenderer draws the same parameters as the reference for the same seed)."""
import os

import numpy as np
import torch

from oracle import encoder_oracle as eo

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'encoder_golden.npz'))
HEADS = ('_theta_deltas', '_translation2ds', '_log_scales', '_log_depths', '_class_probs', '_ffd_coeffs')


def golden_inputs():
    import importlib.util
    spec = importlib.util.spec_from_file_location('make_encoder_golden', os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                                                                     'golden', 'make_encoder_golden.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def seeded_product_derenderer():
    from derender3d.models.derenderer import Derenderer
    torch.manual_seed(int(GOLD['seed']))
    return Derenderer()


def test_seeded_product_weights_are_the_golden_weights():
    m = seeded_product_derenderer()
    sd = m.state_dict()
    assert list(sd.keys()) == [str(k) for k in GOLD['checksum_keys']]
    got = np.array([float(v.double().abs().sum()) for v in sd.values()])
    np.testing.assert_allclose(got, GOLD['checksums'], rtol=1e-6, atol=0)


def test_oracle_reproduces_the_reference_derenderer():
    mk = golden_inputs()
    images, mroi, droi, weights = mk.inputs()
    m = seeded_product_derenderer()           # parameter container only: nothing of the product's forward runs here
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    t = torch.tensor
    ev = eo.derenderer_forward(sd, t(images), t(mroi), t(droi), training=False)
    for k in HEADS:
        np.testing.assert_allclose(ev[k].numpy(), GOLD['eval' + k], rtol=1e-5, atol=1e-6, err_msg=k)
    ps = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and 'running' not in k}
    full = dict(sd)
    full.update(ps)
    tr = eo.derenderer_forward(full, t(images), t(mroi), t(droi), training=True, update_running=True)
    for k in HEADS:
        np.testing.assert_allclose(tr[k].detach().numpy(), GOLD['train' + k], rtol=1e-5, atol=1e-6, err_msg=k)
    sum((tr[k] * t(weights[k])).sum() for k in HEADS).backward()
    for k in mk.GRAD_KEYS:
        g = mk.slice_of(k, ps[k].grad).numpy()
        ref = GOLD['grad/' + k]
        assert np.linalg.norm(g - ref) <= 1e-4 * np.linalg.norm(ref) + 1e-9, k
    for k in ('net.bn1.running_mean', 'net.bn1.running_var', 'net.layer4.1.bn2.running_mean', 'net.layer4.1.bn2.running_var',
              'net.layer2.0.downsample.1.running_var'):
        np.testing.assert_allclose(full[k].numpy(), GOLD['after/' + k], rtol=1e-5, atol=1e-7, err_msg=k)


def test_encoder_has_no_cpu_path():
    import pytest
    m = seeded_product_derenderer()
    with pytest.raises(NotImplementedError):
        m(torch.zeros(2, 3, 64, 64), torch.zeros(2, 2), torch.zeros(2, 2))
