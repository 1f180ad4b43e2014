#!/usr/bin/env python3
"""Generate tests/golden/textural_local_golden.npz: the REFERENCE's LocalEnhancer (textural/models/networks.py:156-206,
`define_G(..., 'local', ...)`) on a seeded input -- state_dict, output, input and parameter gradients.  Same method as
make_textural_golden.py (the reference module imported from where it lies, stub torchvision); a separate file so that
the first golden set stays byte-identical."""
import os

import numpy as np
import torch

from make_textural_golden import HERE, load_reference_networks, put


def main():
    R = load_reference_networks()
    out = {}
    # global trunk at half resolution (ngf 4 * 2 = 8, 2 downsamplings, 2 blocks) + one enhancer level (2 blocks)
    torch.manual_seed(104)
    L = R.define_G(6, 3, 4, 'local', n_downsample_global=2, n_blocks_global=2, n_local_enhancers=1, n_blocks_local=2)
    x = torch.randn(2, 6, 32, 48, requires_grad=True)
    y = L(x)
    w = torch.randn(y.shape)
    put(out, 'L', L, {'x': x}, {'y': y}, (y * w).sum())
    out['L/in/w'] = w.numpy()
    path = os.path.join(HERE, 'textural_local_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote %s: %d arrays, %.1f KiB' % (path, len(out), os.path.getsize(path) / 1024))


if __name__ == '__main__':
    main()
