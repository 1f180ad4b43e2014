"""Micro-benchmark of sdn_conv_gemm on forward shapes that dominate the generator / discriminator (development aid, GPU
only); `--ab` also times every library under lab/ (tools/build_lab_variant.sh) in the same process."""
import ctypes
import glob
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

_i8 = ctypes.c_int8
_LIBS = {}


def lib():
    path = os.environ.get('SDN_LAB_LIB')
    if not path:
        return sdn_hip.lib()
    if path not in _LIBS:
        L = ctypes.CDLL(path)
        try:
            sdn_hip._declare(L)
        except AttributeError:
            pass
        _LIBS[path] = L
    return _LIBS[path]


SHAPES = [  # name, N, IH, IW, cin, cout, k, s, p, reflect, in_relu
    ('res 1024->1024 k3 @24x78', 4, 24, 78, 1024, 1024, 3, 1, 1, 1, 1),
    ('down 256->512 k3 s2 @96x312', 4, 96, 312, 256, 512, 3, 2, 1, 0, 1),
    ('stem 48->64 k7 @384x1248', 4, 384, 1248, 48, 64, 7, 1, 3, 1, 0),
    ('D 256->512 k4 @49x157', 4, 49, 157, 256, 512, 4, 1, 2, 0, 0),
    ('D 64->128 k4 s2 @193x625', 4, 193, 625, 64, 128, 4, 2, 2, 0, 0),
]


def run(name, N, IH, IW, cin, cout, k, s, p, reflect, in_relu, iters=10):
    dev = 'cuda'
    Cip, Cop = cp.cpad(cin), cp.cpad_pow2(cout)
    launches, (OH, OW) = cp.conv_fwd(k, s, p, IH, IW)
    L = launches[0]
    x = torch.randn(N, IH, IW, Cip, device=dev)
    w = torch.randn(cout, cin, k, k, device=dev) * 0.05
    rows, Kp = cp.weight_rows(Cop), cp.kpad(len(L.tapidx), Cip)
    tix = torch.tensor(list(L.tapidx), dtype=torch.int32, device=dev)
    packed = torch.empty(2 * rows * Kp, dtype=torch.bfloat16, device=dev)
    check(lib().sdn_conv_pack_weights(ptr(w), cout, cin, cin * k * k, k * k, ptr(tix), len(L.tapidx), Cip, Kp, rows,
                                      ptr(packed), stream()))
    out = torch.empty(N, OH, OW, Cop, device=dev)
    dy = (_i8 * len(L.taps))(*[t[0] for t in L.taps])
    dx = (_i8 * len(L.taps))(*[t[1] for t in L.taps])

    def call():
        check(lib().sdn_conv_gemm(ptr(x), N, IH, IW, Cip, ptr(out), OH, OW, Cop, L.QH, L.QW, L.istride, L.ostride, L.py,
                                  L.px, len(L.taps), dy, dx, reflect, in_relu, ptr(packed), Kp, rows, None, 0, None, 0, 3,
                                  None, 0, stream()))
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters, 2.0 * N * OH * OW * k * k * cin * cout


def main():
    variants = [('tree', None)]
    if '--ab' in sys.argv:
        variants += [(os.path.basename(f)[:-3], f) for f in sorted(glob.glob(os.path.join(ROOT, 'lab', '*.so')))]
    table, flops = {}, {}
    for rep in range(3):
        for name, path in variants:
            if path:
                os.environ['SDN_LAB_LIB'] = path
            else:
                os.environ.pop('SDN_LAB_LIB', None)
            for sh in SHAPES:
                ms, fl = run(*sh)
                flops[sh[0]] = fl
                table.setdefault(sh[0], {}).setdefault(name, []).append(ms)
    print('%-30s' % 'ms (min of 3)' + ''.join('%16s' % n[:15] for n, _ in variants) + '   TFLOP/s (tree)')
    for sh in SHAPES:
        print('%-30s' % sh[0] + ''.join('%16.3f' % min(table[sh[0]][n]) for n, _ in variants)
              + '   %8.1f' % (flops[sh[0]] / min(table[sh[0]]['tree']) / 1e9))
    print('%-30s' % 'sum' + ''.join('%16.3f' % sum(min(table[sh[0]][n]) for sh in SHAPES) for n, _ in variants))


if __name__ == '__main__':
    main()
