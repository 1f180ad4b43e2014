"""Lab for the r04 tiled MFMA kernels (development aid, GPU only): micro-test of the transpose read, then sdn_conv_tile /
sdn_conv_wgrad_tile against sdn_conv_gemm / sdn_conv_wgrad (the parity-verified r01-r03 kernels) on the layer shapes that
dominate the GAN step -- results compared, both timed in the same process.

    python tools/tile_lab.py [--quick] [--json out.json]
"""
import ctypes
import json
import os
import subprocess
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
DEV = 'cuda'
RESULTS = {}
_LIB = [None]


def the_lib():
    """the product library, or the build named by --lib (probe builds: tools/build_lab_variant.sh)"""
    if _LIB[0] is None:
        if '--lib' in sys.argv:
            L = ctypes.CDLL(sys.argv[sys.argv.index('--lib') + 1])
            sdn_hip._declare(L)
            _LIB[0] = L
        else:
            _LIB[0] = sdn_hip.lib()
    return _LIB[0]


def lab_lib():
    so = os.path.join(ROOT, 'lab', 'liblab.so')
    src = os.path.join(ROOT, 'tools', 'lab', 'lab_kernels.hip')
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O2', '-shared', '-fPIC', src, '-o', so])
    return ctypes.CDLL(so)


def test_tr16():
    """Hypothesis (conv_wtile.hip): within a 16-lane group, lane i receives, as element e, the e-th 8-byte row's i-th bf16:
    out[i][e] = element (i % 4) of the row loaded by lane 4 e + i / 4."""
    L = lab_lib()
    L.lab_tr16.argtypes = [ctypes.c_void_p] * 4
    x = torch.arange(4096, dtype=torch.int16, device=DEV)
    # the kernel's address rule for one fragment read: group g -> block (g & 1), key0 = 8 (g >> 1); lane j: row key0 + j / 4
    lane = torch.arange(64)
    g4, j16 = lane // 16, lane % 16
    key = 8 * (g4 // 2) + (j16 // 4)
    slot = (key + 4 * (g4 % 2)) % 32
    addr = ((g4 % 2) * 1024 + slot * 32 + (j16 % 4) * 8).to(torch.int32).to(DEV)
    out = torch.zeros(256, dtype=torch.int16, device=DEV)
    rc = L.lab_tr16(ptr(x), ptr(addr), ptr(out), stream())
    torch.cuda.synchronize()
    assert rc == 0, rc
    got = out.cpu().view(64, 4)
    # expected: lane (g, j) holds channel j of block (g & 1), positions key0 .. key0 + 3  (element index = byte / 2)
    exp = torch.zeros(64, 4, dtype=torch.int16)
    for l in range(64):
        g, j = l // 16, l % 16
        for e in range(4):
            k = 8 * (g // 2) + e
            s = (k + 4 * (g % 2)) % 32
            exp[l, e] = ((g % 2) * 1024 + s * 32) // 2 + j
    ok = bool(torch.equal(got, exp))
    print('tr16 micro-test:', 'layout hypothesis CONFIRMED' if ok else 'MISMATCH')
    if not ok:
        print('lane: got | expected (elements are LDS bf16 indices)')
        for l in range(64):
            print(l, got[l].tolist(), '|', exp[l].tolist(), ' addr', int(addr[l]))
    RESULTS['tr16_ok'] = ok
    return ok


def planes_of(x, relu=False):
    n = x.numel()
    stride = (n + 7) // 8 * 8
    pl = torch.empty(2 * stride, dtype=torch.bfloat16, device=x.device)
    check(the_lib().sdn_split_planes(ptr(x), n, int(relu), ptr(pl), stride, stream()))
    return pl, stride


def timeit(fn, iters):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best


GEMM_SHAPES = [  # name, kind, N, IH, IW, cin, cout, k, s, p, reflect, in_relu, extras
    ('res 1024->1024 k3 @24x78', 'fwd', 4, 24, 78, 1024, 1024, 3, 1, 1, 1, 1, 'stats'),
    ('res dgrad 1024->1024 k3 @24x78', 'dgrad', 4, 24, 78, 1024, 1024, 3, 1, 1, 1, 0, ''),
    ('down 512->1024 k3 s2 @48x156', 'fwd', 4, 48, 156, 512, 1024, 3, 2, 1, 0, 1, 'stats'),
    ('down 256->512 k3 s2 @96x312', 'fwd', 4, 96, 312, 256, 512, 3, 2, 1, 0, 1, ''),
    ('down 64->128 k3 s2 @384x1248', 'fwd', 4, 384, 1248, 64, 128, 3, 2, 1, 0, 1, 'stats'),
    ('down dgrad 256->512 k3 s2 @96x312', 'dgrad', 4, 96, 312, 256, 512, 3, 2, 1, 0, 0, ''),
    ('up convT 1024->512 k3 s2 @24x78', 'convT', 4, 24, 78, 1024, 512, 3, 2, 1, 0, 1, ''),
    ('up convT 128->64 k3 s2 @192x624', 'convT', 4, 192, 624, 128, 64, 3, 2, 1, 0, 1, 'stats'),
    ('D 256->512 k4 @49x157', 'fwd', 4, 49, 157, 256, 512, 4, 1, 2, 0, 0, 'bias'),
    ('D 64->128 k4 s2 @193x625', 'fwd', 4, 193, 625, 64, 128, 4, 2, 2, 0, 0, 'bias,lrelu'),
    ('small 64->64 k3 @20x30', 'fwd', 2, 20, 30, 64, 64, 3, 1, 1, 1, 0, 'bias,stats,acc'),
    ('small 64->128 k3 @21x33 zero pad', 'fwd', 2, 21, 33, 64, 128, 3, 1, 1, 0, 0, 'bias,stats,lrelu'),
    ('small 96->160 k4 @13x19', 'fwd', 1, 13, 19, 96, 160, 4, 1, 2, 0, 1, 'bias'),
    ('small dgrad 128->64 k3 @17x22', 'dgrad', 2, 17, 22, 64, 128, 3, 1, 1, 1, 0, 'acc'),
]


def run_gemm(name, kind, N, IH, IW, cin, cout, k, s, p, reflect, in_relu, extras, iters):
    L = the_lib()
    torch.manual_seed(hash(name) % 1000)
    if kind == 'fwd':
        launches, (OH, OW) = cp.conv_fwd(k, s, p, IH, IW)
        R, C, sr, sc = cout, cin, cin * k * k, k * k
        w = torch.randn(cout, cin, k, k, device=DEV) * 0.05
        pad_mode = reflect
    elif kind == 'convT':
        launches, (OH, OW) = cp.convT_fwd(k, s, p, 1, IH, IW)
        R, C, sr, sc = cout, cin, k * k, cout * k * k          # ConvTranspose2d weight [cin, cout, k, k]
        w = torch.randn(cin, cout, k, k, device=DEV) * 0.05
        pad_mode = 0
    else:   # data gradient of Conv2d(cin -> cout): input = d(out) with `cout` channels over the conv's output grid
        OHc, OWc = cp.conv_out_size(IH, k, s, p), cp.conv_out_size(IW, k, s, p)
        launches, (OH, OW) = cp.conv_dgrad(k, s, p, IH, IW, bool(reflect))
        w = torch.randn(cout, cin, k, k, device=DEV) * 0.05
        R, C, sr, sc = cin, cout, k * k, cin * k * k
        IH, IW, cin, cout = OHc, OWc, cout, cin
        pad_mode = 0
    Cip, Cop = cp.cpad(cin), cp.cpad_pow2(cout)
    x = torch.randn(N, IH, IW, Cip, device=DEV)
    if Cip != cin:
        x[..., cin:] = 0
    bias = torch.randn(Cop, device=DEV) if 'bias' in extras else None
    act = 1 if 'lrelu' in extras else 0
    acc = 'acc' in extras
    outs, stats = {}, {}
    times = {}
    halo_ok = (len(launches) == 1 and launches[0].istride == 1 and launches[0].ostride == 1 and len(launches[0].taps) >= 9
               and Cip % 32 == 0 and Cop > 64 and '--no-halo' not in sys.argv)
    for which in (('old', 'new', 'halo') if halo_ok else ('old', 'new')):
        out = torch.full((N, OH, OW, Cop), 0.5, device=DEV) if acc else torch.zeros(N, OH, OW, Cop, device=DEV)
        st = torch.zeros(N, 8, Cop, 2, dtype=torch.float64, device=DEV) if 'stats' in extras else None
        calls = []
        for Lh in launches:
            if not Lh.taps:
                continue
            tix = torch.tensor(list(Lh.tapidx), dtype=torch.int32, device=DEV)
            nt = len(Lh.taps)
            dy = (_i8 * nt)(*[t[0] for t in Lh.taps])
            dx = (_i8 * nt)(*[t[1] for t in Lh.taps])
            if which == 'old':
                rows, Kp = cp.weight_rows(Cop), cp.kpad(nt, Cip)
                packed = torch.empty(2 * rows * Kp, dtype=torch.bfloat16, device=DEV)
                check(L.sdn_conv_pack_weights(ptr(w), R, C, sr, sc, ptr(tix), nt, Cip, Kp, rows, ptr(packed), stream()))
                calls.append(lambda Lh=Lh, nt=nt, dy=dy, dx=dx, packed=packed, Kp=Kp, rows=rows: check(L.sdn_conv_gemm(
                    ptr(x), N, IH, IW, Cip, ptr(out), OH, OW, Cop, Lh.QH, Lh.QW, Lh.istride, Lh.ostride, Lh.py, Lh.px, nt, dy, dx,
                    pad_mode, in_relu, ptr(packed), Kp, rows, ptr(bias), act, ptr(st), int(acc), 3, None, 0, stream())))
            elif which == 'halo':
                rows = (Cop + 127) // 128 * 128
                packed = torch.empty(2 * rows * nt * Cip, dtype=torch.bfloat16, device=DEV)
                check(L.sdn_conv_pack_weights_kmajor(ptr(w), R, C, sr, sc, ptr(tix), nt, Cip, rows, ptr(packed), stream()))
                pl, pstride = planes_of(x, relu=bool(in_relu))
                calls.append(lambda nt=nt, dy=dy, dx=dx, packed=packed, rows=rows, pl=pl, pstride=pstride: check(
                    L.sdn_conv_halo(ptr(pl), pstride, N, IH, IW, Cip, ptr(out), OH, OW, Cop, nt, dy, dx, pad_mode, ptr(packed),
                                    rows, ptr(bias), act, ptr(st), int(acc), 0, stream())))
            else:
                rows = (Cop + 127) // 128 * 128 if Cop > 64 else 64
                packed = torch.empty(2 * rows * nt * Cip, dtype=torch.bfloat16, device=DEV)
                check(L.sdn_conv_pack_weights_kmajor(ptr(w), R, C, sr, sc, ptr(tix), nt, Cip, rows, ptr(packed), stream()))
                pl, pstride = planes_of(x, relu=bool(in_relu))
                calls.append(lambda Lh=Lh, nt=nt, dy=dy, dx=dx, packed=packed, rows=rows, pl=pl, pstride=pstride: check(
                    L.sdn_conv_tile(ptr(pl), pstride, N, IH, IW, Cip, ptr(out), None, 0, 0, OH, OW, Cop, Lh.QH, Lh.QW,
                                    Lh.istride, Lh.ostride, Lh.py, Lh.px, nt, dy, dx, pad_mode, ptr(packed), rows, ptr(bias),
                                    act, ptr(st), int(acc), stream())))

        def run_all():
            for c in calls:
                c()
        run_all()
        torch.cuda.synchronize()
        outs[which] = out.clone()
        stats[which] = st.clone() if st is not None else None
        if not acc:
            times[which] = timeit(run_all, iters)
    scale = float(outs['old'].abs().max())
    err = float((outs['new'] - outs['old']).abs().max()) / max(scale, 1e-30)
    serr = 0.0
    if stats['old'] is not None:
        so, sn = stats['old'].sum(1), stats['new'].sum(1)
        serr = float(((sn - so).abs() / (so.abs() + 1e-3 * so.abs().max())).max())
    flops = 2.0 * N * OH * OW * k * k * cin * cout / (s * s if kind in ('convT',) or (kind == 'dgrad' and s > 1) else 1)
    to, tn = times.get('old', 0.0), times.get('new', 0.0)
    ok = err < 2e-5 and serr < 1e-5
    if 'halo' in outs:
        herr = float((outs['halo'] - outs['old']).abs().max()) / max(scale, 1e-30)
        hs = 0.0
        if stats['old'] is not None:
            so, sn = stats['old'].sum(1), stats['halo'].sum(1)
            hs = float(((sn - so).abs() / (so.abs() + 1e-3 * so.abs().max())).max())
        th = times.get('halo', 0.0)
        print('%-36s %s  err %.2e  stats %.1e | halo %7.3f ms %6.1f TF | x%.2f vs old' % (
            '   halo', 'ok ' if (herr < 2e-5 and hs < 1e-5) else 'BAD', herr, hs, th, flops / th / 1e9 if th else 0, to / th if th else 0),
            flush=True)
    print('%-36s %s  err %.2e  stats %.1e | old %7.3f ms %6.1f TF | new %7.3f ms %6.1f TF | x%.2f' % (
        name, 'ok ' if ok else 'BAD', err, serr, to, flops / to / 1e9 if to else 0, tn, flops / tn / 1e9 if tn else 0,
        to / tn if tn else 0), flush=True)
    RESULTS.setdefault('gemm', []).append({'name': name, 'ok': ok, 'err': err, 'stats_err': serr, 'old_ms': to, 'new_ms': tn,
                                           'flops': flops})
    return ok


WGRAD_SHAPES = [  # name, kind, N, OH, OW, cout, cin, k, s, p, reflect
    ('res 1024->1024 k3 @24x78', 'conv', 4, 24, 78, 1024, 1024, 3, 1, 1, 1),
    ('down 512->1024 k3 s2 @24x78', 'conv', 4, 24, 78, 1024, 512, 3, 2, 1, 0),
    ('down 256->512 k3 s2 @48x156', 'conv', 4, 48, 156, 512, 256, 3, 2, 1, 0),
    ('down 64->128 k3 s2 @192x624', 'conv', 4, 192, 624, 128, 64, 3, 2, 1, 0),
    ('stem 48->64 k7 @384x1248', 'conv', 4, 384, 1248, 64, 48, 7, 1, 3, 1),
    ('D 256->512 k4 @50x158', 'conv', 4, 50, 158, 512, 256, 4, 1, 2, 0),
    ('D 18->64 k4 s2 @193x625', 'conv', 4, 193, 625, 64, 18, 4, 2, 2, 0),
    ('up convT 1024->512 k3 s2 @24x78', 'convT', 4, 24, 78, 512, 1024, 3, 2, 1, 0),
    ('small 32->48 k3 @13x9', 'conv', 3, 13, 9, 48, 32, 3, 1, 1, 0),
]


def run_wgrad(name, kind, N, OH, OW, cout, cin, k, s, p, reflect, iters):
    L = the_lib()
    torch.manual_seed(hash(name) % 1000)
    if kind == 'conv':
        IH, IW = (OH - 1) * s + k - 2 * p, (OW - 1) * s + k - 2 * p
        Cr, Cc = cp.cpad_pow2(cout), cp.cpad(cin)
        rows = torch.randn(N, OH, OW, Cr, device=DEV)
        gath = torch.randn(N, IH, IW, Cc, device=DEV)
        WL = cp.conv_wgrad(k, s, p, OH, OW)
        GH, GW = IH, IW
        pad_mode = reflect
    else:   # ConvTranspose2d(cin -> cout): rows = the input over the input grid (OH x OW here), gath = d(out)
        GH, GW = cp.convT_out_size(OH, k, s, p, 1), cp.convT_out_size(OW, k, s, p, 1)
        Cr, Cc = cp.cpad_pow2(cin), cp.cpad_pow2(cout)
        rows = torch.randn(N, OH, OW, Cr, device=DEV)
        gath = torch.randn(N, GH, GW, Cc, device=DEV)
        WL = cp.convT_wgrad(k, s, p, OH, OW)
        pad_mode = 0
    nt = len(WL.taps)
    dy = (_i8 * nt)(*[t[0] for t in WL.taps])
    dx = (_i8 * nt)(*[t[1] for t in WL.taps])
    n_tiles = ((Cr + 127) // 128 if Cr > 64 else 1) * ((nt * Cc + 127) // 128)
    splits = cp.wgrad_splits(N * WL.QH * WL.QW, n_tiles)
    dws, times = {}, {}
    rp, rstride = planes_of(rows)
    gp, gstride = planes_of(gath)
    for which in ('old', 'new'):
        dw = torch.zeros(Cr, nt * Cc, device=DEV)
        if which == 'old':
            def call():
                check(L.sdn_conv_wgrad(ptr(rows), ptr(gath), ptr(dw), N, WL.QH, WL.QW, Cr, GH, GW, Cc, WL.istride, nt, dy, dx,
                                       pad_mode, 0, 0, splits, 3, None, 0, stream()))
        else:
            def call():
                check(L.sdn_conv_wgrad_tile(ptr(rp), rstride, ptr(gp), gstride, ptr(dw), N, WL.QH, WL.QW, Cr, GH, GW, Cc,
                                            WL.istride, nt, dy, dx, pad_mode, stream()))
        call()
        torch.cuda.synchronize()
        dws[which] = dw.clone()
        times[which] = timeit(call, iters)
    scale = float(dws['old'].abs().max())
    err = float((dws['new'] - dws['old']).abs().max()) / max(scale, 1e-30)
    flops = 2.0 * N * WL.QH * WL.QW * nt * (cin if kind == 'conv' else cout) * (cout if kind == 'conv' else cin)
    ok = err < 5e-5
    to, tn = times['old'], times['new']
    print('%-36s %s  err %.2e | old %7.3f ms %6.1f TF | new %7.3f ms %6.1f TF | x%.2f' % (
        name, 'ok ' if ok else 'BAD', err, to, flops / to / 1e9, tn, flops / tn / 1e9, to / tn), flush=True)
    RESULTS.setdefault('wgrad', []).append({'name': name, 'ok': ok, 'err': err, 'old_ms': to, 'new_ms': tn, 'flops': flops})
    return ok


def main():
    quick = '--quick' in sys.argv
    iters = 3 if quick else 10
    print(torch.cuda.get_device_name(0))
    try:
        test_tr16()
    except Exception as e:   # noqa: BLE001
        print('tr16 micro-test failed to run:', e)
    if '--one' in sys.argv:   # one shape of each kind, few launches: the target of tools/gpu_pmc_tile.sh
        run_gemm(*GEMM_SHAPES[0], iters=2)
        run_wgrad(*WGRAD_SHAPES[0], iters=2)
        return
    if '--probes' in sys.argv:
        # timing-only builds of k_conv_tile<2, 2> (csrc/conv_tile.hip, SDN_TILE_PROBES): results are wrong by construction
        names = {0: 'product', 1: 'no copies in the K loop', 2: 'no MFMAs'}
        for var in (0, 1, 2, 0):
            os.environ['SDN_TILE_VARIANT'] = str(var)
            print('== variant %d: %s' % (var, names[var]), flush=True)
            for sh in (GEMM_SHAPES[0], GEMM_SHAPES[8]):
                run_gemm(*sh, iters=iters)
        return
    print('---- sdn_conv_tile vs sdn_conv_gemm')
    for sh in GEMM_SHAPES:
        try:
            if '--wgrad-only' not in sys.argv:
                run_gemm(*sh, iters=iters)
        except Exception as e:   # noqa: BLE001
            print('%-36s EXCEPTION %s' % (sh[0], e), flush=True)
    for mode in (('0', '1') if '--wmodes' in sys.argv else (None,)):
      if mode is not None:
        os.environ['SDN_WTILE_MODE'] = mode
      print('---- sdn_conv_wgrad_tile vs sdn_conv_wgrad' + ('' if mode is None else '  (SDN_WTILE_MODE=%s)' % mode))
      for sh in WGRAD_SHAPES:
        try:
            if '--gemm-only' not in sys.argv:
                run_wgrad(*sh, iters=iters)
        except Exception as e:   # noqa: BLE001
            print('%-36s EXCEPTION %s' % (sh[0], e), flush=True)
    if '--json' in sys.argv:
        with open(sys.argv[sys.argv.index('--json') + 1], 'w') as fh:
            json.dump(RESULTS, fh, indent=1)


if __name__ == '__main__':
    main()
