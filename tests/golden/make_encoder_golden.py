#!/usr/bin/env python3
"""Generate tests/golden/encoder_golden.npz: outputs and gradients of the REFERENCE's own Derenderer class
(/root/reference/geometric/derender3d/models/derenderer.py, loaded from where it lies) for seeded weights and inputs.

torchvision is absent from this image: the module the reference asks it for (`torchvision.models.resnet18`) is provided
by oracle/encoder_oracle.py:RefResNet18, a restatement of torchvision 0.2.1's published ResNet-18 (see that file's
header: the backbone itself is unpinned, the Derenderer head is the reference's code).  Weights are NOT stored (47 MB):
both this script and the tests draw them from torch.manual_seed(SEED) -- the product Derenderer creates its layers in the
reference's order, which this script asserts -- and the golden file carries per-tensor checksums to verify it.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SEED = 20260926
GRAD_KEYS = ['_fc3.bias', 'fc1.weight', 'net.fc.bias', 'net.bn1.weight', 'net.bn1.bias', 'net.conv1.weight',
             'net.layer1.0.conv1.weight', 'net.layer2.0.downsample.0.weight', 'net.layer2.0.downsample.1.weight',
             'net.layer3.1.bn2.bias', 'net.layer4.1.bn2.weight', 'net.layer4.0.conv1.weight']


def inputs(n=8, size=64):
    rng = np.random.default_rng(SEED)
    images = rng.normal(size=(n, 3, size, size)).astype(np.float32)
    mroi = rng.uniform(-0.3, 0.3, (n, 2)).astype(np.float32)
    droi = rng.uniform(0.05, 0.4, (n, 2)).astype(np.float32)
    weights = {k: rng.normal(size=s).astype(np.float32) for k, s in (
        ('_theta_deltas', (n, 2)), ('_translation2ds', (n, 2)), ('_log_scales', (n, 3)), ('_log_depths', (n, 1)),
        ('_class_probs', (n, 8)), ('_ffd_coeffs', (n, 8, 192)))}
    return images, mroi, droi, weights


def slice_of(name, g):
    """what is stored of a gradient: all of a small tensor, the leading rows of a big one"""
    g = g.reshape(g.shape[0], -1) if g.dim() > 1 else g
    return g[:16].contiguous() if g.dim() > 1 else g


def main():
    sys.path.insert(0, ROOT)
    for p in (os.path.join(ROOT, '3d-sdn_amd'), os.path.join(ROOT, '3d-sdn_amd', 'geometric')):
        sys.path.insert(0, p)
    os.environ['SDN_ALLOW_RANDOM_INIT'] = '1'
    from oracle import encoder_oracle as eo
    tv = types.ModuleType('torchvision')
    tv.models = types.ModuleType('torchvision.models')
    tv.models.resnet18 = lambda pretrained=False: eo.RefResNet18()
    sys.modules['torchvision'] = tv
    sys.modules['torchvision.models'] = tv.models
    spec = importlib.util.spec_from_file_location('ref_derenderer', '/root/reference/geometric/derender3d/models/derenderer.py')
    ref_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_mod)

    torch.manual_seed(SEED)
    ref = ref_mod.Derenderer()
    del sys.modules['torchvision'], sys.modules['torchvision.models']
    from derender3d.models.derenderer import Derenderer
    torch.manual_seed(SEED)
    prod = Derenderer()
    rs, ps = ref.state_dict(), prod.state_dict()
    assert list(rs.keys()) == list(ps.keys())
    for k in rs:
        assert torch.allclose(rs[k].float(), ps[k].float(), rtol=1e-6, atol=0), k
    out = {'seed': np.int64(SEED)}
    out['checksum_keys'] = np.array(list(rs.keys()))
    out['checksums'] = np.array([float(v.double().abs().sum()) for v in rs.values()])

    images, mroi, droi, weights = inputs()
    ti, tm, td = torch.tensor(images), torch.tensor(mroi), torch.tensor(droi)
    ref.eval()
    with torch.no_grad():
        ev = ref(ti, tm, td)
    sd = {k: v.clone() for k, v in rs.items()}
    fo = eo.derenderer_forward(sd, ti, tm, td, training=False)
    for k in ev:
        assert torch.allclose(ev[k], fo[k], rtol=1e-5, atol=1e-6), k
        out['eval' + k] = ev[k].numpy()
    ref.train()
    tr = ref(ti, tm, td)
    loss = sum((tr[k] * torch.tensor(weights[k])).sum() for k in tr)
    loss.backward()
    for k in tr:
        out['train' + k] = tr[k].detach().numpy()
    params = dict(ref.named_parameters())
    for k in GRAD_KEYS:
        out['grad/' + k] = slice_of(k, params[k].grad).numpy()
    after = ref.state_dict()
    for k in ('net.bn1.running_mean', 'net.bn1.running_var', 'net.layer4.1.bn2.running_mean', 'net.layer4.1.bn2.running_var',
              'net.layer2.0.downsample.1.running_var', 'net.bn1.num_batches_tracked'):
        out['after/' + k] = after[k].numpy()
    np.savez_compressed(os.path.join(HERE, 'encoder_golden.npz'), **out)
    print('wrote encoder_golden.npz (%d arrays, %.0f KB)' % (len(out), os.path.getsize(os.path.join(HERE, 'encoder_golden.npz')) / 1024))


if __name__ == '__main__':
    main()
