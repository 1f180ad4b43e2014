"""Micro-benchmark of sdn_conv_wgrad on the three weight-gradient shapes that dominate the generator (development aid;
GPU only).  Prints time and algorithmic TFLOP/s per shape; `--check` compares against an fp64 einsum on a small case."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, '3d-sdn_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)
import sdn_hip  # noqa: E402
from sdn_hip import check, ptr, stream  # noqa: E402
from sdn_hip import convplan as cp  # noqa: E402

_LIBS = {}


def lib():
    """the product library, or the one named by SDN_LAB_LIB (A/B runs against a build of another revision)"""
    path = os.environ.get('SDN_LAB_LIB')
    if not path:
        return sdn_hip.lib()
    if path not in _LIBS:
        L = ctypes.CDLL(path)
        try:
            sdn_hip._declare(L)
        except AttributeError:   # a revision without some of today's entry points: declare what this tool calls
            pass
        _LIBS[path] = L
    return _LIBS[path]

_i8 = ctypes.c_int8

SHAPES = [  # name, N, OH, OW, cout, cin, k, s, p, reflect
    ('res 1024->1024 k3 @24x78', 4, 24, 78, 1024, 1024, 3, 1, 1, 1),
    ('down 256->512 k3 s2 @48x156', 4, 48, 156, 512, 256, 3, 2, 1, 0),
    ('stem 48->64 k7 @384x1248', 4, 384, 1248, 64, 48, 7, 1, 3, 1),
    ('D 256->512 k4 @50x158', 4, 50, 158, 512, 256, 4, 1, 2, 0),
]


def run(name, N, OH, OW, cout, cin, k, s, p, reflect, iters=10, quiet=False):
    dev = 'cuda'
    IH, IW = (OH - 1) * s + k - 2 * p, (OW - 1) * s + k - 2 * p
    Cr, Cc = cp.cpad_pow2(cout), cp.cpad(cin)
    dz = torch.randn(N, OH, OW, Cr, device=dev)
    x = torch.randn(N, IH, IW, Cc, device=dev)
    WL = cp.conv_wgrad(k, s, p, OH, OW)
    ntaps = len(WL.taps)
    dy = (_i8 * ntaps)(*[t[0] for t in WL.taps])
    dx = (_i8 * ntaps)(*[t[1] for t in WL.taps])
    dw = torch.zeros(Cr, ntaps * Cc, device=dev)
    n_tiles = ((Cr + 127) // 128 if Cr > 64 else 1) * ((ntaps * Cc + 127) // 128)
    splits = cp.wgrad_splits(N * WL.QH * WL.QW, n_tiles)

    ws = torch.empty(splits * Cr * ntaps * Cc, device=dev) if '--det' in sys.argv else None   # ordered split-K sums

    def call():
        check(lib().sdn_conv_wgrad(ptr(dz), ptr(x), ptr(dw), N, WL.QH, WL.QW, Cr, IH, IW, Cc, WL.istride, ntaps, dy, dx,
                                   reflect, 0, 0, splits, 3, ptr(ws), ws.numel() * 4 if ws is not None else 0, stream()))
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * N * OH * OW * k * k * cin * cout
    if not quiet:
        print('%-30s splits %4d  %8.3f ms  %7.1f TFLOP/s' % (name, splits, ms, fl / ms / 1e9))
    return ms


def check_small(N=2, OH=12, OW=20, cout=128, cin=32, k=3, s=1, p=1, reflect=0, splits=None):
    """sdn_conv_wgrad against an fp64 einsum on a small case; prints the error and where it sits"""
    import torch.nn.functional as F
    dev = 'cuda'
    torch.manual_seed(0)
    IH, IW = (OH - 1) * s + k - 2 * p, (OW - 1) * s + k - 2 * p
    Cr, Cc = cp.cpad_pow2(cout), cp.cpad(cin)
    dz = torch.zeros(N, OH, OW, Cr, device=dev)
    dz[..., :cout] = torch.randn(N, OH, OW, cout, device=dev)
    x = torch.zeros(N, IH, IW, Cc, device=dev)
    x[..., :cin] = torch.randn(N, IH, IW, cin, device=dev)
    WL = cp.conv_wgrad(k, s, p, OH, OW)
    ntaps = len(WL.taps)
    dy = (_i8 * ntaps)(*[t[0] for t in WL.taps])
    dx = (_i8 * ntaps)(*[t[1] for t in WL.taps])
    dw = torch.zeros(Cr, ntaps * Cc, device=dev)
    n_tiles = ((Cr + 127) // 128 if Cr > 64 else 1) * ((ntaps * Cc + 127) // 128)
    sp = splits if splits is not None else cp.wgrad_splits(N * WL.QH * WL.QW, n_tiles)
    check(lib().sdn_conv_wgrad(ptr(dz), ptr(x), ptr(dw), N, WL.QH, WL.QW, Cr, IH, IW, Cc, WL.istride, ntaps, dy, dx,
                               reflect, 0, 0, sp, 3, None, 0, stream()))
    # reference: dW[o, t, c] = sum_{n, y, x} dz[n, y, x, o] * xpad[n, y*s + ky, x*s + kx, c]
    xn = x.permute(0, 3, 1, 2).double()
    xp = F.pad(xn, (p, p, p, p), mode='reflect' if reflect else 'constant')
    ref = torch.zeros(Cr, ntaps, Cc, dtype=torch.float64, device=dev)
    d64 = dz.double()
    for t, (ky, kx) in enumerate([(a, b) for a in range(k) for b in range(k)]):
        win = xp[:, :, ky:ky + (OH - 1) * s + 1:s, kx:kx + (OW - 1) * s + 1:s]   # [N, Cc, OH, OW]
        ref[:, t, :] = torch.einsum('nyxo,ncyx->oc', d64, win)
    got = dw.double().reshape(Cr, ntaps, Cc)
    err = (got - ref).abs()
    rel = float(err.max() / ref.abs().max())
    print('check N%d %dx%d %d->%d k%d s%d splits %d: rel max err %.3e' % (N, OH, OW, cin, cout, k, s, sp, rel))
    if rel > 1e-3:
        bad = err > 1e-3 * ref.abs().max()
        print('  bad fraction %.3f; by row group of 16: %s' % (float(bad.float().mean()),
              [round(float(bad[g * 16:(g + 1) * 16].float().mean()), 2) for g in range(min(Cr // 16, 8))]))
        print('  by tap: %s' % [round(float(bad[:, t].float().mean()), 2) for t in range(ntaps)])
        print('  by channel (first 16): %s' % [round(float(bad[:, :, c].float().mean()), 2) for c in range(min(Cc, 16))])
        print('  ratio got/ref sample: %s' % (got[0, 0, :4] / ref[0, 0, :4]).tolist())


def main():
    if '--check' in sys.argv:
        check_small()
        check_small(splits=1)
        check_small(N=1, OH=4, OW=8, splits=1)
        check_small(cout=20, cin=3, k=7, p=3, reflect=1)
        return
    import glob
    variants = [('tree', None)]
    if '--ab' in sys.argv:   # every library under lab/ (tools/build_lab_variant.sh, or a build of another revision)
        variants += [(os.path.basename(f)[:-3], f) for f in sorted(glob.glob(os.path.join(ROOT, 'lab', '*.so')))]
    table = {}
    for rep in range(3 if len(variants) > 1 else 1):
        for name, path in variants:
            if path:
                os.environ['SDN_LAB_LIB'] = path
            else:
                os.environ.pop('SDN_LAB_LIB', None)
            for sh in SHAPES:
                ms = run(*sh, quiet=len(variants) > 1)
                table.setdefault(sh[0], {}).setdefault(name, []).append(ms)
    if len(variants) > 1:   # best of the repetitions (the first pass also warms the clocks up)
        print('%-30s' % 'ms (min of 3)' + ''.join('%16s' % n[:15] for n, _ in variants))
        for sh in SHAPES:
            print('%-30s' % sh[0] + ''.join('%16.3f' % min(table[sh[0]][n]) for n, _ in variants))
        print('%-30s' % 'sum' + ''.join('%16.3f' % sum(min(table[sh[0]][n]) for sh in SHAPES) for n, _ in variants))


if __name__ == '__main__':
    main()
