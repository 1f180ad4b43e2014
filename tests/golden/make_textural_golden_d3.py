#!/usr/bin/env python3
"""Generate tests/golden/textural_d3_golden.npz: the REFERENCE's 3-scale MultiscaleDiscriminator -- the `--num_D 3`
configuration of BASELINE configs[3] (textural/models/networks.py:368-461, `define_D(..., num_D=3, getIntermFeat=True)`)
-- on a seeded input: state_dict, all 15 feature maps, input and parameter gradients.  Same method as
make_textural_golden.py (the reference module imported from where it lies, stub torchvision); a separate file so that the
first golden set stays byte-identical."""
import os

import numpy as np
import torch

from make_textural_golden import HERE, load_reference_networks, put


def main():
    R = load_reference_networks()
    out = {}
    torch.manual_seed(105)
    D = R.define_D(7, 8, 3, 'instance', False, 3, True)
    x = torch.randn(1, 7, 64, 88, requires_grad=True)   # 64 x 88 -> 32 x 44 -> 16 x 22: three pooled scales
    res = D(x)
    assert len(res) == 3 and all(len(s) == 5 for s in res)
    loss = 0
    outs = {}
    for s, scale in enumerate(res):
        for j, f in enumerate(scale):
            wj = torch.randn(f.shape)
            out['D3/in/w%d_%d' % (s, j)] = wj.numpy()
            outs['f%d_%d' % (s, j)] = f
            loss = loss + (f * wj).sum()
    put(out, 'D3', D, {'x': x}, outs, loss)
    path = os.path.join(HERE, 'textural_d3_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote %s: %d arrays, %.1f KiB' % (path, len(out), os.path.getsize(path) / 1024))


if __name__ == '__main__':
    main()
