// Per-element bodies of the weight pack / gradient unpack kernels (conv_norm.hip: k_pack_weights, k_unpack_grad; conv_tile.hip:
// k_pack_weights_kmajor), shared with the multi-tensor kernel of fast_program.hip so that a coalesced launch writes the
// bytes the per-tensor launches write.
#pragma once
#include <hip/hip_runtime.h>

namespace sdn {

// Wm[r, k(t, c)] = W[r * sr + c * sc + tapidx[t]] split to bf16 hi / lo in MFMA fragment order (layout: conv_norm.hip)
__device__ __forceinline__ void pack_weights_element(long i, const float* __restrict__ w, int R, int C, long sr, long sc,
                                                     const int* __restrict__ tapidx, int ntaps, int Ccp, int Kp, int rows,
                                                     __bf16* __restrict__ packed)
{
    if (i >= (long)rows * Kp) return;
    const int r = (int)(i / Kp), k = (int)(i % Kp);
    int t = k / Ccp, c = k % Ccp;
    if ((Ccp & 31) == 0 && Kp == ntaps * Ccp) {   // channel-block-major columns (see k_conv_gemm's K order)
        const int step = k >> 5, cb = step / ntaps;
        t = step - cb * ntaps;
        c = cb * 32 + (k & 31);
    }
    float v = 0.f;
    if (r < R && t < ntaps && c < C) v = w[(size_t)r * sr + (size_t)c * sc + tapidx[t]];
    const __bf16 h = (__bf16)v;
    const int lane = (r & 31) + 32 * ((k & 15) >> 3);
    const size_t blk = ((size_t)(r >> 5) * (Kp >> 4) + (k >> 4)) * 2;
    packed[blk * 512 + lane * 8 + (k & 7)] = h;
    packed[(blk + 1) * 512 + lane * 8 + (k & 7)] = (__bf16)(v - (float)h);
}

// K-major: packed[r][step][part][k % 32] for step = cb * ntaps + t, c = cb * 32 + k % 32
__device__ __forceinline__ void pack_weights_kmajor_element(long i, const float* __restrict__ w, int R, int C, long sr, long sc,
                                                            const int* __restrict__ tapidx, int ntaps, int Ccp, int rows,
                                                            __bf16* __restrict__ packed)
{
    const long K = (long)ntaps * Ccp;
    if (i >= (long)rows * K) return;
    const int r = (int)(i / K), k = (int)(i % K);
    const int step = k >> 5, cb = step / ntaps, t = step - cb * ntaps, c = cb * 32 + (k & 31);
    float v = 0.f;
    if (r < R && c < C) v = w[(size_t)r * sr + (size_t)c * sc + tapidx[t]];
    const __bf16 h = (__bf16)v;
    const size_t base = ((size_t)r * (K >> 5) + step) * 64 + (k & 31);
    packed[base] = h;
    packed[base + 32] = (__bf16)(v - (float)h);
}

// grad_w[r * sr + c * sc + tapidx[t]] (+)= dw[r, t * Ccp + c]
__device__ __forceinline__ void unpack_grad_element(long i, const float* __restrict__ dw, int R, int C, long sr, long sc,
                                                    const int* __restrict__ tapidx, int ntaps, int Ccp,
                                                    float* __restrict__ grad_w, int accumulate)
{
    const long ncols = (long)ntaps * Ccp;
    if (i >= (long)R * ncols) return;
    const int r = (int)(i / ncols);
    const int k = (int)(i % ncols);
    const int t = k / Ccp, c = k % Ccp;
    if (c >= C) return;
    float* dst = grad_w + (size_t)r * sr + (size_t)c * sc + tapidx[t];
    *dst = accumulate ? *dst + dw[i] : dw[i];
}

}  // namespace sdn
