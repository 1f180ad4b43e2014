#!/usr/bin/env python3
"""Generate tests/golden/vgg_golden.npz: the reference's OWN VGGLoss / Vgg19 classes on a seeded VGG19.

/root/reference/textural/models/networks.py:137-149 (VGGLoss) and :467-497 (Vgg19: which torchvision `features` indices go
into which slice) are imported as they lie.  `torchvision.models.vgg19(pretrained=True)` -- the only thing they take from the
absent torchvision, and a download besides -- is stubbed by an object whose `.features` is the cfg-'E' Sequential of
torchvision 0.2.1 (Conv2d 3x3 pad 1 + ReLU(inplace) + MaxPool2d(2, 2)) carrying SEEDED weights
(oracle/textural_oracle.vgg19_seeded_state): no pretrained file can exist here.  The run is fp64 (weights rounded to fp32
first), `.cuda()` is the identity.  Stored: the seed, a checksum of the weights, inputs, the loss, its gradient wrt x, the
five slice outputs' shapes and a strided sample of each.  Pins the slicing (relu1_1 ... relu5_1), the weights [1/32 ... 1],
the detach of the target branch and the L1 mean reduction.  Runs only where /root/reference exists.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT]
REF = os.environ.get('SDN_REFERENCE_ROOT', '/root/reference')
SEED = 20260926


def main():
    from oracle import textural_oracle as to
    sd = to.vgg19_seeded_state(SEED)

    def vgg19(pretrained=False):
        assert pretrained
        feats = []
        for lay in to.vgg19_layout():
            if lay[0] == 'conv':
                feats.append(nn.Conv2d(lay[1], lay[2], kernel_size=3, padding=1))
            elif lay[0] == 'relu':
                feats.append(nn.ReLU(inplace=True))
            else:
                feats.append(nn.MaxPool2d(kernel_size=2, stride=2))
        m = types.SimpleNamespace(features=nn.Sequential(*feats))
        m.features.load_state_dict({k[len('features.'):]: v for k, v in sd.items()})
        return m
    tv = types.ModuleType('torchvision')
    tv.models = types.ModuleType('torchvision.models')
    tv.models.vgg19 = vgg19
    sys.modules['torchvision'] = tv
    sys.modules['torchvision.models'] = tv.models
    nn.Module.cuda = lambda self, *a, **k: self
    sys.path.insert(0, os.path.join(REF, 'textural'))
    from models import networks as ref_networks          # the reference module, as it lies
    loss_mod = ref_networks.VGGLoss([0]).double()
    assert all(not p.requires_grad for p in loss_mod.vgg.parameters())
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 3, 48, 80, generator=g, dtype=torch.float64).requires_grad_(True)
    y = torch.randn(2, 3, 48, 80, generator=g, dtype=torch.float64)
    feats = loss_mod.vgg(x)
    loss = loss_mod(x, y)
    loss.backward()
    out = {'seed': np.int64(SEED), 'x': x.detach().numpy(), 'y': y.numpy(), 'loss': np.float64(loss.item()), 'grad_x': x.grad.numpy(),
           'weight_checksum': np.float64(sum(float(v.double().abs().sum()) for v in sd.values()))}
    for i, f in enumerate(feats):
        out['feat%d_shape' % i] = np.asarray(f.shape, np.int64)
        out['feat%d_sample' % i] = f.detach().reshape(-1)[::97].numpy()
    path = os.path.join(HERE, 'vgg_golden.npz')
    np.savez_compressed(path, **out)
    print('loss', loss.item(), '|grad|', float(x.grad.norm()), [tuple(f.shape) for f in feats])
    print('wrote', path, os.path.getsize(path))


if __name__ == '__main__':
    main()
