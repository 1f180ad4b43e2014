"""numpy stand-in for the HIP conv kernels (TEST INFRASTRUCTURE): executes the launch descriptions of
sdn_hip/convplan.py with the exact index rules of csrc/conv_gemm.hip / conv_wgrad.hip / conv_norm.hip, in float64, so the
host-side geometry (tap tables, phases, weight packing strides, reflection fold) is verified on the CPU."""
import numpy as np


def resolve(v, n, pad_mode):
    if pad_mode == 1:
        if v < 0:
            v = -v
        if v >= n:
            v = 2 * n - 2 - v
    return v if 0 <= v < n else None


def pack(w_flat, R, C, sr, sc, tapidx, ccp, rows):
    """k_pack_weights: packed[r, t*ccp + c] = w[r*sr + c*sc + tapidx[t]]"""
    out = np.zeros((rows, len(tapidx) * ccp))
    for r in range(R):
        for t, ti in enumerate(tapidx):
            for c in range(C):
                out[r, t * ccp + c] = w_flat[r * sr + c * sc + ti]
    return out


def unpack(dw, R, C, sr, sc, tapidx, ccp, grad_flat):
    for r in range(R):
        for t, ti in enumerate(tapidx):
            for c in range(C):
                grad_flat[r * sr + c * sc + ti] += dw[r, t * ccp + c]


def gemm(x, out, L, pad_mode, packed, in_relu=False, accumulate=False):
    """k_conv_gemm.  x [N, IH, IW, Cip], out [N, OH, OW, Cop] (modified in place), packed [rows, ntaps*Cip]."""
    N, IH, IW, Cip = x.shape
    Cop = out.shape[3]
    xs = np.maximum(x, 0) if in_relu else x
    for n in range(N):
        for qy in range(L.QH):
            for qx in range(L.QW):
                acc = np.zeros(Cop)
                for t, (dy, dx) in enumerate(L.taps):
                    iy = resolve(qy * L.istride + dy, IH, pad_mode)
                    ix = resolve(qx * L.istride + dx, IW, pad_mode)
                    if iy is None or ix is None:
                        continue
                    acc += packed[:Cop, t * Cip:(t + 1) * Cip] @ xs[n, iy, ix]
                oy, ox = qy * L.ostride + L.py, qx * L.ostride + L.px
                if accumulate:
                    out[n, oy, ox] += acc
                else:
                    out[n, oy, ox] = acc


def wgrad(rows, gath, WL, pad_mode, relu_rows=False, relu_gath=False):
    """k_conv_wgrad -> dw [Cr, ntaps*Cc]"""
    N, QH, QW, Cr = rows.shape
    _, GH, GW, Cc = gath.shape
    a = np.maximum(rows, 0) if relu_rows else rows
    b = np.maximum(gath, 0) if relu_gath else gath
    dw = np.zeros((Cr, len(WL.taps) * Cc))
    for n in range(N):
        for qy in range(QH):
            for qx in range(QW):
                for t, (dy, dx) in enumerate(WL.taps):
                    iy = resolve(qy * WL.istride + dy, GH, pad_mode)
                    ix = resolve(qx * WL.istride + dx, GW, pad_mode)
                    if iy is None or ix is None:
                        continue
                    dw[:, t * Cc:(t + 1) * Cc] += np.outer(a[n, qy, qx], b[n, iy, ix])
    return dw


def reflect_fold(gp, H, W, Pd, out=None):
    """k_reflect_fold"""
    N, Hp, Wp, C = gp.shape
    res = np.zeros((N, H, W, C))

    def srcs(y, n):
        s = [y + Pd]
        if 1 <= y <= Pd:
            s.append(Pd - y)
        if n - 1 - Pd <= y <= n - 2:
            s.append(2 * (n - 1) - y + Pd)
        return s
    for y in range(H):
        for x in range(W):
            for ys in srcs(y, H):
                for xs in srcs(x, W):
                    res[:, y, x] += gp[:, ys, xs]
    if out is not None:
        res += out
    return res
