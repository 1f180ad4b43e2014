// Free-form-deformation decode for a batch of objects that use different mesh templates.
//
// Reference: FFD.forward, /root/reference/geometric/derender3d/models/transforms.py:68-99:
//     V[v, c] = sum_{ijk} (P0 + dP)[c, i, j, k] * B[v, i, j, k]
// run once per object from a Python loop, after re-uploading B (3.8 MB) and P0 with `.cuda()` on every call
// (transforms.py:97) and through a [V,3,4,4,4] temporary.  Here the Bernstein bases of ALL templates live on the
// device once, stored coefficient-major  Bt[class][j][v]  so that the 64 lanes of a wave read 64 consecutive
// vertices (coalesced 256 B per coefficient), the 3 x 64 control points of an object sit in LDS, and a whole
// frame's objects are decoded in one launch:  grid = (ceil(Vmax / 256), N).
//
// HBM-bound: reads 64 floats per vertex (the basis), writes 3.  Backward: grad_P[b, c, j] = sum_v g[b, v, c] * Bt[j][v]
// -- one workgroup per (object, coefficient), block reduction in a fixed order (deterministic).
#include "sdn_common.h"

namespace sdn {

constexpr int NCOEF_MAX = 512;
// (Measured and dropped, r06: the workgroups of a template's first object decoding ALL objects of that template, four at a time, so
// that the basis is read once per template instead of once per object -- same bits, and k_ffd_fwd 15.7 -> 28.8 us, k_ffd_bwd 20.6 ->
// 35.4: the 16-object frame shares ~7 templates, i.e. 7 working workgroup columns with 2.3x the work each; the basis re-reads
// were L2 hits that cost less than the lost parallelism.)

__global__ __launch_bounds__(256) void k_ffd_fwd(const float* __restrict__ Bt, const float* __restrict__ P,
                                                  const int32_t* __restrict__ cls, int vmax, int ncoef,
                                                  float* __restrict__ out)
{
    __shared__ float Ps[3 * NCOEF_MAX];
    const int b = blockIdx.y;
    for (int t = threadIdx.x; t < 3 * ncoef; t += 256) Ps[t] = P[(size_t)b * 3 * ncoef + t];
    __syncthreads();
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= vmax) return;
    const float* bt = Bt + (size_t)cls[b] * ncoef * vmax + v;
    float x = 0.f, y = 0.f, z = 0.f;
#pragma unroll 8    // (32 measured in r06: 15.7 -> 23.2 us)
    for (int j = 0; j < ncoef; j++) {
        const float w = bt[(size_t)j * vmax];
        x += Ps[j] * w;
        y += Ps[ncoef + j] * w;
        z += Ps[2 * ncoef + j] * w;
    }
    float* o = out + ((size_t)b * vmax + v) * 3;
    o[0] = x;
    o[1] = y;
    o[2] = z;
}

// r04: a workgroup of 1024 threads takes FFD_JB coefficients of one object (was: 256 threads, one coefficient).  The object's
// vertex gradients -- three quarters of the bytes a (b, j) pair reads -- are fetched once per FFD_JB rows of Bt (L2 traffic per
// frame step 377 -> 165 MB) and the 16 waves keep the chip as full as the 1024 small workgroups did.
constexpr int FFD_BWD_THREADS = 1024, FFD_JB = 4;
__global__ __launch_bounds__(FFD_BWD_THREADS) void k_ffd_bwd(const float* __restrict__ Bt, const int32_t* __restrict__ cls,
                                                              const float* __restrict__ g, int vmax, int ncoef,
                                                              float* __restrict__ grad_P)
{
    constexpr int NW = FFD_BWD_THREADS / 64;
    __shared__ float red[FFD_JB][3][NW];
    const int b = blockIdx.y, j0 = blockIdx.x * FFD_JB;
    const float* bt = Bt + ((size_t)cls[b] * ncoef + j0) * vmax;
    const float* gb = g + (size_t)b * vmax * 3;
    float s[FFD_JB][3];
#pragma unroll
    for (int u = 0; u < FFD_JB; u++) s[u][0] = s[u][1] = s[u][2] = 0.f;
#pragma unroll 2    // (8 measured in r06: no change)
    for (int v = threadIdx.x; v < vmax; v += FFD_BWD_THREADS) {
        const float g0 = gb[3 * v + 0], g1 = gb[3 * v + 1], g2 = gb[3 * v + 2];
#pragma unroll
        for (int u = 0; u < FFD_JB; u++) {
            const float w = (j0 + u < ncoef) ? bt[(size_t)u * vmax + v] : 0.f;
            s[u][0] += g0 * w;
            s[u][1] += g1 * w;
            s[u][2] += g2 * w;
        }
    }
#pragma unroll
    for (int u = 0; u < FFD_JB; u++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            s[u][0] += __shfl_xor(s[u][0], o, 64);
            s[u][1] += __shfl_xor(s[u][1], o, 64);
            s[u][2] += __shfl_xor(s[u][2], o, 64);
        }
        if ((threadIdx.x & 63) == 0) {
            red[u][0][threadIdx.x >> 6] = s[u][0];
            red[u][1][threadIdx.x >> 6] = s[u][1];
            red[u][2][threadIdx.x >> 6] = s[u][2];
        }
    }
    __syncthreads();
    if (threadIdx.x < 3 * FFD_JB) {
        const int u = threadIdx.x / 3, d = threadIdx.x % 3;
        if (j0 + u < ncoef) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NW; w++) t += red[u][d][w];     // fixed order
            grad_P[((size_t)b * 3 + d) * ncoef + j0 + u] = t;
        }
    }
}

// out[i, j] = (base ? base[j] : 0) + sum_k x[i, k] * M[k, j]   (transpose: M[j, k]);  x [n, m], M [m, m] -- the linear
// constraint map of FFD.constrain applied to a frame's coefficient rows.  m is a few hundred: one thread per output, the
// row of x through LDS.
// r06: 64 outputs x 4 quarters of the k range per workgroup (was: one thread per output walking all m products behind each
// other -- 192 dependent-latency loads, 8.4 us for 16 x 192 x 192).  The quarters meet in LDS in a fixed order.
__global__ __launch_bounds__(256) void k_ffd_coefficients(const float* __restrict__ x, const float* __restrict__ M,
                                                          const float* __restrict__ base, int m, int transpose,
                                                          float* __restrict__ out)
{
    extern __shared__ float xrow[];        // [m] the object's row, then [4][64] partial sums
    float* part = xrow + m;
    const int i = blockIdx.y;
    for (int k = threadIdx.x; k < m; k += 256) xrow[k] = x[(size_t)i * m + k];
    __syncthreads();
    const int jj = threadIdx.x & 63, kq = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + jj;
    const int per = (m + 3) / 4, k0 = kq * per, k1 = min(m, k0 + per);
    float acc = 0.0f;
    if (j < m) {
        if (transpose) {
            const float* row = M + (size_t)j * m;
#pragma unroll 8
            for (int k = k0; k < k1; k++) acc = fmaf(xrow[k], row[k], acc);
        } else {
#pragma unroll 8
            for (int k = k0; k < k1; k++) acc = fmaf(xrow[k], M[(size_t)k * m + j], acc);
        }
    }
    part[kq * 64 + jj] = acc;
    __syncthreads();
    if (kq == 0 && j < m)
        out[(size_t)i * m + j] = (base ? base[j] : 0.0f) + ((part[jj] + part[64 + jj]) + (part[128 + jj] + part[192 + jj]));
}

}  // namespace sdn

using namespace sdn;

SDN_API int sdn_ffd_coefficients(const float* x, const float* M, const float* base, int n, int m, int transpose, float* out,
                                 sdnStream stream)
{
    if (!x || !M || !out || n <= 0 || m <= 0 || m > 12288)
        return fail(SDN_EINVAL, "sdn_ffd_coefficients: bad arguments (m <= 12288)");
    hipLaunchKernelGGL(k_ffd_coefficients, dim3(cdiv(m, 64), n), dim3(256), (size_t)(m + 256) * sizeof(float), (hipStream_t)stream,
                       x, M, base, m, transpose, out);
    return check_launch("k_ffd_coefficients");
}

SDN_API int sdn_ffd_decode(const float* Bt, const float* P, const int32_t* cls, int n, int vmax, int ncoef, float* out,
                           sdnStream stream)
{
    if (!Bt || !P || !cls || !out || n <= 0 || vmax <= 0 || ncoef <= 0 || ncoef > NCOEF_MAX)
        return fail(SDN_EINVAL, "sdn_ffd_decode: bad arguments (ncoef <= %d)", NCOEF_MAX);
    hipLaunchKernelGGL(k_ffd_fwd, dim3(cdiv(vmax, 256), n), dim3(256), 0, (hipStream_t)stream, Bt, P, cls, vmax, ncoef,
                       out);
    return check_launch("k_ffd_fwd");
}

SDN_API int sdn_ffd_decode_bwd(const float* Bt, const int32_t* cls, const float* grad_out, int n, int vmax, int ncoef,
                               float* grad_P, sdnStream stream)
{
    if (!Bt || !cls || !grad_out || !grad_P || n <= 0 || vmax <= 0 || ncoef <= 0 || ncoef > NCOEF_MAX)
        return fail(SDN_EINVAL, "sdn_ffd_decode_bwd: bad arguments");
    hipLaunchKernelGGL(k_ffd_bwd, dim3(cdiv(ncoef, FFD_JB), n), dim3(FFD_BWD_THREADS), 0, (hipStream_t)stream, Bt, cls, grad_out, vmax, ncoef,
                       grad_P);
    return check_launch("k_ffd_bwd");
}
