"""Geometry of the textural convolutions as "gather GEMM" launches (pure Python, no GPU needed).

Every Conv2d / ConvTranspose2d of the reference's textural networks (textural/models/networks.py:211-239, 244-283,
286-308, 412-449) -- forward, data gradient and weight gradient -- is expressed as one or more launches of the two MFMA
kernels behind sdn_conv_gemm / sdn_conv_wgrad (include/sdn_hip.h):

    out[n, qy*os + py, qx*os + px, :] = sum_t  in[n, qy*is + dy_t, qx*is + dx_t, :] @ W_t        (gemm)
    dW_t = sum_{n,q} rows[n, q, :]^T  gath[n, q*is + d_t, :]                                     (wgrad)

This module only builds those descriptions; tests/test_convplan.py executes them with numpy and checks them against
torch.nn.functional on the CPU, so the index arithmetic is verified without a GPU.
"""
from collections import namedtuple

# one sdn_conv_gemm call.  taps: [(dy, dx)];  tapidx: index ky*kw + kx of each tap in the kernel window
Launch = namedtuple('Launch', 'QH QW istride ostride py px taps tapidx')
# one sdn_conv_wgrad call
WLaunch = namedtuple('WLaunch', 'QH QW istride taps tapidx')


def cpad(c):
    """Channel padding of an activation tensor: multiple of 16."""
    return (c + 15) // 16 * 16


def cpad_pow2(c):
    """Padding of tensors that pass through the norm / elementwise kernels: power of two >= 16."""
    p = 16
    while p < c:
        p *= 2
    return p


def weight_rows(cop):
    """Rows of a packed weight matrix: the N tile the gemm launcher picks for `cop` output channels."""
    if cop > 64:
        return (cop + 127) // 128 * 128
    return 64 if cop > 32 else 32


def tile_weight_rows(cop):
    """Rows of a K-major weight matrix for sdn_conv_tile: its N tile is 128 for cop > 64, else 64."""
    return (cop + 127) // 128 * 128 if cop > 64 else 64


def kpad(ntaps, ccp):
    return (ntaps * ccp + 31) // 32 * 32


def conv_out_size(i, k, s, p):
    return (i + 2 * p - k) // s + 1


def convT_out_size(i, k, s, p, op):
    return (i - 1) * s - 2 * p + k + op


def conv_fwd(k, s, p, IH, IW):
    """Conv2d(kernel k, stride s, padding p -- zero or reflected) over an IH x IW input."""
    OH, OW = conv_out_size(IH, k, s, p), conv_out_size(IW, k, s, p)
    taps = [(ky - p, kx - p) for ky in range(k) for kx in range(k)]
    return [Launch(OH, OW, s, 1, 0, 0, taps, list(range(k * k)))], (OH, OW)


def _phases(k, s, p, OH, OW):
    """Gather form of a transposed convolution: output position o = s*q + phase reads input q + (phase + p - ky)/s for the
    taps ky with (phase + p - ky) % s == 0."""
    out = []
    for py in range(s):
        for px in range(s):
            QH, QW = (OH - py + s - 1) // s, (OW - px + s - 1) // s
            if QH <= 0 or QW <= 0:
                continue
            taps, idx = [], []
            for ky in range(k):
                if (py + p - ky) % s:
                    continue
                for kx in range(k):
                    if (px + p - kx) % s:
                        continue
                    taps.append(((py + p - ky) // s, (px + p - kx) // s))
                    idx.append(ky * k + kx)
            out.append(Launch(QH, QW, 1, s, py, px, taps, idx))
    return out


def convT_fwd(k, s, p, op, IH, IW):
    """ConvTranspose2d(k, stride s, padding p, output_padding op) as s*s phase launches."""
    OH, OW = convT_out_size(IH, k, s, p, op), convT_out_size(IW, k, s, p, op)
    return _phases(k, s, p, OH, OW), (OH, OW)


def conv_dgrad(k, s, p, IH, IW, reflect):
    """Data gradient of Conv2d: launches gather from d(out) and write d(in).
    reflect: the conv read a ReflectionPad2d(p)-padded input; the gradient is produced on the PADDED grid
    (IH+2p x IW+2p, as for a pad-0 conv) and folded back by sdn_reflect_fold.  Returns (launches, (GH, GW) of the grid
    written)."""
    if reflect:
        GH, GW, pe = IH + 2 * p, IW + 2 * p, 0
    else:
        GH, GW, pe = IH, IW, p
    if s == 1:
        taps = [(pe - ky, pe - kx) for ky in range(k) for kx in range(k)]
        return [Launch(GH, GW, 1, 1, 0, 0, taps, list(range(k * k)))], (GH, GW)
    return _phases(k, s, pe, GH, GW), (GH, GW)


def convT_dgrad(k, s, p, IH, IW):
    """Data gradient of ConvTranspose2d: a strided gather from d(out) over the input grid."""
    taps = [(ky - p, kx - p) for ky in range(k) for kx in range(k)]
    return [Launch(IH, IW, s, 1, 0, 0, taps, list(range(k * k)))], (IH, IW)


def conv_wgrad(k, s, p, OH, OW):
    """Weight gradient of Conv2d: rows = d(out) over the output grid, gathered = the input."""
    taps = [(ky - p, kx - p) for ky in range(k) for kx in range(k)]
    return WLaunch(OH, OW, s, taps, list(range(k * k)))


def convT_wgrad(k, s, p, IH, IW):
    """Weight gradient of ConvTranspose2d: rows = the input over the input grid, gathered = d(out)."""
    taps = [(ky - p, kx - p) for ky in range(k) for kx in range(k)]
    return WLaunch(IH, IW, s, taps, list(range(k * k)))


def wgrad_splits(npos, n_tiles, target_blocks=None):
    """K slices of a weight-gradient launch: enough blocks to fill 256 CUs many times over (the tile count alone
    quantises badly: 576 tiles = 2.25 per CU), at least 32 steps each (measured: 1024 / 64 -> 4096 / 32 = -8 % wgrad time)."""
    import os
    if target_blocks is None:
        target_blocks = int(os.environ.get('SDN_WGRAD_TARGET', '4096'))
    steps = (npos + 31) // 32
    want = max(1, (target_blocks + n_tiles // 2) // max(n_tiles, 1))
    ms = int(os.environ.get('SDN_WGRAD_MINSTEPS', '32'))
    return max(1, min(want, steps // ms if steps >= ms else 1))
