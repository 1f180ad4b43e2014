#!/usr/bin/env python3
"""Generate tests/golden/textural_golden.npz by running the REFERENCE's own textural networks on the CPU.

Runs only in the build container (needs /root/reference).  The reference module
/root/reference/textural/models/networks.py is imported unmodified; `torchvision` (absent here, used only by its Vgg19
class) is stubbed in sys.modules.  For small instances of every network family the script stores the seeded input, the
complete state_dict, the forward outputs and the gradients of a fixed scalar loss with respect to the input and every
parameter -- the vectors that pin oracle/textural_oracle.py and, on the GPU box, the HIP path.

    python tests/golden/make_textural_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('SDN_REFERENCE_ROOT', '/root/reference')


def load_reference_networks():
    tv = types.ModuleType('torchvision')
    tv.models = types.ModuleType('torchvision.models')
    sys.modules.setdefault('torchvision', tv)
    sys.modules.setdefault('torchvision.models', tv.models)
    spec = importlib.util.spec_from_file_location('ref_textural_networks',
                                                  os.path.join(REF, 'textural', 'models', 'networks.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    class _Np:
        """The reference's Encoder.forward calls np.any(<torch bool tensor>) (networks.py:320), which numpy 2 / torch 2
        reject; only that call is adapted, through the module's `np` name -- the reference file is untouched."""
        def __getattr__(self, name):
            return getattr(np, name)

        @staticmethod
        def any(a, *args, **kw):
            return bool(a.any()) if isinstance(a, torch.Tensor) else np.any(a, *args, **kw)
    mod.np = _Np()
    return mod


def put(out, prefix, net, inputs, outputs, loss):
    loss.backward()
    for k, v in net.state_dict().items():
        out['%s/sd/%s' % (prefix, k)] = v.detach().numpy().copy()
    for k, p in net.named_parameters():
        out['%s/grad/%s' % (prefix, k)] = p.grad.detach().numpy().copy()
    for k, v in inputs.items():
        out['%s/in/%s' % (prefix, k)] = v.detach().numpy().copy()
        if v.grad is not None:
            out['%s/gin/%s' % (prefix, k)] = v.grad.detach().numpy().copy()
    for k, v in outputs.items():
        out['%s/out/%s' % (prefix, k)] = v.detach().numpy().copy()


def main():
    R = load_reference_networks()
    out = {}
    # ---- GlobalGenerator: 6 -> 3 channels, ngf 8, 2 downsamplings, 2 residual blocks, 16 x 24 input
    torch.manual_seed(101)
    G = R.define_G(6, 3, 8, 'global', n_downsample_global=2, n_blocks_global=2)
    x = torch.randn(2, 6, 16, 24, requires_grad=True)
    y = G(x)
    w = torch.randn(y.shape)
    put(out, 'G', G, {'x': x}, {'y': y}, (y * w).sum())
    out['G/in/w'] = w.numpy()
    # ---- Encoder: 3 -> 2 channels, nef 4, 2 downsamplings, with instance pooling
    torch.manual_seed(102)
    E = R.define_G(3, 2, 4, 'encoder', n_downsample_global=2, isTrain=False)
    x = torch.randn(2, 3, 16, 16, requires_grad=True)
    inst = torch.zeros(2, 1, 16, 16)
    inst[0, :, 2:9, 3:12] = 7
    inst[0, :, 10:, :5] = 3
    inst[1, :, :8, 8:] = 7
    y = E(x, inst.clone())
    w = torch.randn(y.shape)
    put(out, 'E', E, {'x': x}, {'y': y}, (y * w).sum())
    out['E/in/inst'] = inst.numpy()
    out['E/in/w'] = w.numpy()
    # ---- MultiscaleDiscriminator: 5 channels in, ndf 8, 3 layers, 2 scales, intermediate features
    torch.manual_seed(103)
    D = R.define_D(5, 8, 3, 'instance', False, 2, True)
    x = torch.randn(2, 5, 32, 40, requires_grad=True)
    res = D(x)
    loss = 0
    outs = {}
    for s, scale in enumerate(res):
        for j, f in enumerate(scale):
            wj = torch.randn(f.shape)
            out['D/in/w%d_%d' % (s, j)] = wj.numpy()
            outs['f%d_%d' % (s, j)] = f
            loss = loss + (f * wj).sum()
    put(out, 'D', D, {'x': x}, outs, loss)
    path = os.path.join(HERE, 'textural_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote %s: %d arrays, %.1f KiB' % (path, len(out), os.path.getsize(path) / 1024))


if __name__ == '__main__':
    main()
