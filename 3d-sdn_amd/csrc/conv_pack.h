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

// ---- r05: the same two packs, EIGHT consecutive columns per thread (g = element index / 8).  The per-element bodies above
// spend three runtime integer divisions and two 2-byte stores on every weight (k_weights_multi ran at ~1.3 TB/s: 3 ms of the
// GAN step for 4 GB of traffic); a group of eight aligned columns shares its row, tap and channel block (K and Kp are multiples
// of 32, Ccp of 16), is eight source loads at the parameter's column stride and leaves as two 16-byte stores.  Same bytes.
typedef __attribute__((ext_vector_type(8))) __bf16 pack_bf16x8;

__device__ __forceinline__ void pack_split8(const float (&v)[8], __bf16* hi_dst, __bf16* lo_dst)
{
    pack_bf16x8 h, l;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const __bf16 hh = (__bf16)v[e];
        h[e] = hh;
        l[e] = (__bf16)(v[e] - (float)hh);
    }
    *reinterpret_cast<pack_bf16x8*>(hi_dst) = h;
    *reinterpret_cast<pack_bf16x8*>(lo_dst) = l;
}

__device__ __forceinline__ void pack_weights_group8(long g, const float* __restrict__ w, int R, int C, long sr, long sc,
                                                    const int* __restrict__ tapidx, int ntaps, int Ccp, int Kp, int rows,
                                                    __bf16* __restrict__ packed)
{
    const long i = g * 8;
    if (i >= (long)rows * Kp) return;
    const int r = (int)(i / Kp), k = (int)(i % Kp);       // k % 8 == 0
    int t = k / Ccp, c = k % Ccp;
    if ((Ccp & 31) == 0 && Kp == ntaps * Ccp) {   // channel-block-major columns (see k_conv_gemm's K order)
        const int step = k >> 5, cb = step / ntaps;
        t = step - cb * ntaps;
        c = cb * 32 + (k & 31);
    }
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = 0.f;
    if (r < R && t < ntaps) {
        const float* src = w + (size_t)r * sr + tapidx[t];
#pragma unroll
        for (int e = 0; e < 8; e++)
            if (c + e < C) v[e] = src[(size_t)(c + e) * sc];
    }
    const int lane = (r & 31) + 32 * ((k & 15) >> 3);
    const size_t blk = ((size_t)(r >> 5) * (Kp >> 4) + (k >> 4)) * 2;
    pack_split8(v, packed + blk * 512 + lane * 8, packed + (blk + 1) * 512 + lane * 8);
}

__device__ __forceinline__ void pack_weights_kmajor_group8(long g, const float* __restrict__ w, int R, int C, long sr, long sc,
                                                           const int* __restrict__ tapidx, int ntaps, int Ccp, int rows,
                                                           __bf16* __restrict__ packed)
{
    const long K = (long)ntaps * Ccp;
    const long i = g * 8;
    if (i >= (long)rows * K) return;
    const int r = (int)(i / K), k = (int)(i % K);         // k % 8 == 0
    const int step = k >> 5, cb = step / ntaps, t = step - cb * ntaps, c = cb * 32 + (k & 31);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = 0.f;
    if (r < R) {
        const float* src = w + (size_t)r * sr + tapidx[t];
#pragma unroll
        for (int e = 0; e < 8; e++)
            if (c + e < C) v[e] = src[(size_t)(c + e) * sc];
    }
    __bf16* base = packed + ((size_t)r * (K >> 5) + step) * 64 + (k & 31);
    pack_split8(v, base, base + 32);
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

namespace sdn {

// the unpack with FOUR consecutive columns per thread (g = element index / 4): one 16-byte load of dw, the row / tap arithmetic
// once, four stores at the parameter's column stride (Ccp is a multiple of 16: a group never straddles a tap)
__device__ __forceinline__ void unpack_grad_group4(long g, const float* __restrict__ dw, int R, int C, long sr, long sc,
                                                   const int* __restrict__ tapidx, int ntaps, int Ccp,
                                                   float* __restrict__ grad_w, int accumulate)
{
    typedef __attribute__((ext_vector_type(4))) float pack_f32x4;
    const long ncols = (long)ntaps * Ccp;
    const long i = g * 4;
    if (i >= (long)R * ncols) return;
    const int r = (int)(i / ncols);
    const int k = (int)(i % ncols);
    const int t = k / Ccp, c = k % Ccp;
    if (c >= C) return;
    const pack_f32x4 v = *reinterpret_cast<const pack_f32x4*>(dw + i);
    float* dst = grad_w + (size_t)r * sr + (size_t)c * sc + tapidx[t];
#pragma unroll
    for (int e = 0; e < 4; e++)
        if (c + e < C) dst[(size_t)e * sc] = accumulate ? dst[(size_t)e * sc] + v[e] : v[e];
}

}  // namespace sdn

namespace sdn {

// ---- r05: the gradient unpack as a transpose through LDS.  dw is [R, ntaps * Ccp] (tap-major columns), the parameter gradient
// is addressed r * sr + c * sc + tapidx[t] with the taps INNERMOST (sc = kh * kw for both Conv2d [O, I, kh, kw] rows = O and
// ConvTranspose2d [I, O, kh, kw] rows = I): per row a [ntaps x C] -> [C x sc] transpose.  The per-element / per-group forms read
// dw coalesced and scatter 4-byte stores sc floats apart (66 us for the 1024 x 1024 x 3 x 3 weight = 1.1 TB/s, the largest share
// of k_weights_multi); here a workgroup takes one row and UNPACK_CB consecutive columns: the taps are read tap by tap (coalesced
// along c), parked in LDS as tile[c][tap], and the UNPACK_CB * sc contiguous floats of the destination leave coalesced.
// Positions of the window that no tap of the list addresses are left alone, as before (accumulate 0 = plain stores).
constexpr int UNPACK_CB = 64, UNPACK_MAX_SC = 64;
__host__ __device__ __forceinline__ bool unpack_rows_ok(long sc, int ntaps) { return sc >= 1 && sc <= UNPACK_MAX_SC && ntaps <= UNPACK_MAX_SC; }
__host__ __device__ __forceinline__ long unpack_rows_blocks(int R, int C) { return (long)R * ((C + UNPACK_CB - 1) / UNPACK_CB); }

// blk: this tensor's block index in [0, unpack_rows_blocks); tile: UNPACK_CB * (UNPACK_MAX_SC + 1) floats; inv: UNPACK_MAX_SC ints
__device__ __forceinline__ void unpack_grad_rows(long blk, const float* __restrict__ dw, int R, int C, long sr, long sc,
                                                 const int* __restrict__ tapidx, int ntaps, int Ccp, float* __restrict__ grad_w,
                                                 int accumulate, float* tile, int* inv)
{
    const int cblocks = (C + UNPACK_CB - 1) / UNPACK_CB;
    const int r = (int)(blk / cblocks), c0 = (int)(blk % cblocks) * UNPACK_CB;
    const int tid = threadIdx.x;
    const int isc = (int)sc;
    if (tid < UNPACK_MAX_SC) inv[tid] = -1;
    __syncthreads();
    if (tid < ntaps) {
        const int x = tapidx[tid];
        if (x >= 0 && x < isc) inv[x] = tid;   // (a window position named twice keeps one of them, as the scattered stores did)
    }
    const float* src = dw + (size_t)r * ntaps * Ccp;
    float* dst_row = grad_w + (size_t)r * sr;
    const int cc = tid & (UNPACK_CB - 1);
    const int c = c0 + cc;
    for (int t = tid / UNPACK_CB; t < ntaps; t += 256 / UNPACK_CB) {
        float v = 0.f;
        if (c < C) v = src[(size_t)t * Ccp + c];
        tile[cc * (UNPACK_MAX_SC + 1) + t] = v;
        const int x = tapidx[t];
        if (c < C && (x < 0 || x >= isc)) {   // a tap outside the innermost window: its own scattered store
            float* d = dst_row + (size_t)c * sc + x;
            *d = accumulate ? *d + v : v;
        }
    }
    __syncthreads();
    const int ncols = min(UNPACK_CB, C - c0);
    float* out = dst_row + (size_t)c0 * sc;
    for (int j = tid; j < ncols * isc; j += 256) {
        const int cl = j / isc, x = j - cl * isc;
        const int t = inv[x];
        if (t >= 0) {
            const float v = tile[cl * (UNPACK_MAX_SC + 1) + t];
            out[j] = accumulate ? out[j] + v : v;
        }
    }
}

}  // namespace sdn
