// Device arithmetic of the rasterizer.  Every function reproduces, operation for operation, what the
// reference's kernels compute (file:line under /root/reference/geometric/neural_renderer/), including
// the sub-expressions CUDA evaluates in double because of un-suffixed literals.  The translation unit
// is compiled with -ffp-contract=off and correctly rounded fp32 divide/sqrt, so `a * b + c` below is
// two roundings, like in the CPU oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sdn {

// CUDA float->int conversion: NaN -> 0, saturating (rasterize.py uses implicit conversions).
__device__ __forceinline__ int cvt_i32(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (int)0x80000000;
    return (int)v;
}

// rasterize.py:120,252,307,537
__device__ __forceinline__ bool is_backface(const float f[9])
{
    const float a = (f[7] - f[1]) * (f[3] - f[0]);
    const float b = (f[4] - f[1]) * (f[6] - f[0]);
    return a < b;
}

// rasterize.py:138,258,546:  0.5 * (v * is + is - 1)   (0.5 * x is exact, so float == double path)
__device__ __forceinline__ float ndc_to_pixel(float v, float is_f)
{
    float t = v * is_f;
    t = t + is_f;
    t = t - 1.0f;
    return 0.5f * t;
}

// rasterize.py:292-293: (2. * i + 1 - is) / is in double, rounded to float
__device__ __forceinline__ float pixel_to_ndc(int i, int is)
{
    return (float)((2. * (double)i + 1. - (double)is) / (double)is);
}

// rasterize.py:147-155,261-269.  p[k] = (x,y) of vertex k in pixel coordinates; returns denominator.
__device__ __forceinline__ float face_inverse(const float px[3], const float py[3], float inv[9])
{
    inv[0] = py[1] - py[2];
    inv[1] = px[2] - px[1];
    inv[2] = px[1] * py[2] - px[2] * py[1];
    inv[3] = py[2] - py[0];
    inv[4] = px[0] - px[2];
    inv[5] = px[2] * py[0] - px[0] * py[2];
    inv[6] = py[0] - py[1];
    inv[7] = px[1] - px[0];
    inv[8] = px[0] * py[1] - px[1] * py[0];
    float den = px[2] * (py[0] - py[1]) + px[0] * (py[1] - py[2]);
    den = den + px[1] * (py[2] - py[0]);
#pragma unroll
    for (int k = 0; k < 9; k++) inv[k] = inv[k] / den;
    return den;
}

// rasterize.py:311-313 -- true when pixel (xp,yp) in NDC passes all three edge tests
__device__ __forceinline__ bool inside_ndc(const float f[9], float xp, float yp)
{
    if ((yp - f[1]) * (f[3] - f[0]) < (xp - f[0]) * (f[4] - f[1])) return false;
    if ((yp - f[4]) * (f[6] - f[3]) < (xp - f[3]) * (f[7] - f[4])) return false;
    if ((yp - f[7]) * (f[0] - f[6]) < (xp - f[6]) * (f[1] - f[7])) return false;
    return true;
}

// rasterize.py:316-328 -- clamp through fmax/fmin (NaN loses), renormalise
__device__ __forceinline__ void bary_weights(const float inv[9], int xi, int yi, float w[3])
{
    const float fx = (float)xi, fy = (float)yi;
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float t = inv[3 * k + 0] * fx + inv[3 * k + 1] * fy;
        t = t + inv[3 * k + 2];
        t = fminf(fmaxf(t, 0.0f), 1.0f);
        w[k] = t;
        sum = sum + t;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) w[k] = w[k] / sum;
}

// rasterize.py:331 -- 1. / (w0/z0 + w1/z1 + w2/z2).  A double reciprocal rounded to float equals the
// correctly rounded float reciprocal (53 >= 2*24+2), so float division is exact parity.
__device__ __forceinline__ float persp_depth(const float w[3], float z0, float z1, float z2)
{
    float s = w[0] / z0 + w[1] / z1;
    s = s + w[2] / z2;
    return 1.0f / s;
}

// ---- K1, the reference's DEFAULT ("unsafe") forward kernel, rasterize.py:102-236 (scripts/env.sh:11 selects it): one thread
// per face walks the pixel columns between its leftmost and rightmost vertex and, per column, the rows between two edge
// interpolations.  Its COVERAGE RULE (pixel-space scanline) differs from the safe kernels' NDC edge tests on pixels whose
// centre meets an edge, its barycentrics are evaluated on the x-sorted vertices (different roundings), and depth ties go to
// whichever thread gets the per-pixel spinlock first.  SDN_K1_COVERAGE reproduces rule and arithmetic exactly and settles
// ties as the oracle's serial face order does (lowest face index): a deterministic K1.
// CUDA's double -> int conversion: NaN -> 0, saturating.
__device__ __forceinline__ int cvt_i32_d(double v)
{
    if (v != v) return 0;
    if (v >= 2147483647.0) return 2147483647;
    if (v <= -2147483648.0) return (int)0x80000000;
    return (int)v;
}

struct K1Face {
    float px[3], py[3];     // pixel coordinates of the vertices sorted by x: leftmost, middle, rightmost (:123-142)
    float sa, sb, sc;       // slopes of p0->p1, p1->p2, p0->p2 (:163-176); sa / sb unused when their edge is vertical
    int pi[3];              // pi[l] = original number of sorted vertex l
    int xi_min, xi_max;     // :158-159; no column when xi_min > xi_max
    bool a_ok, b_ok, dead;  // p1x != p0x, p2x != p1x, p0x == p2x ("line, not triangle", :144)
};

__device__ __forceinline__ void k1_order(const float f[9], int pi[3])
{
    if (f[0] < f[3]) {
        pi[0] = (f[6] < f[0]) ? 2 : 0;
        pi[2] = (f[3] < f[6]) ? 2 : 1;
    } else {
        pi[0] = (f[6] < f[3]) ? 2 : 1;
        pi[2] = (f[0] < f[6]) ? 2 : 0;
    }
    pi[1] = 0;
#pragma unroll
    for (int k = 0; k < 3; k++)
        if (pi[0] != k && pi[2] != k) pi[1] = k;
}

__device__ __forceinline__ K1Face k1_setup(const float f[9], int is)
{
    K1Face K;
    k1_order(f, K.pi);
    const float is_f = (float)is;
#pragma unroll
    for (int l = 0; l < 3; l++) {
        // (select by comparisons: K.pi is not a compile-time index)
        const int k = K.pi[l];
        const float x = k == 0 ? f[0] : (k == 1 ? f[3] : f[6]);
        const float y = k == 0 ? f[1] : (k == 1 ? f[4] : f[7]);
        K.px[l] = ndc_to_pixel(x, is_f);
        K.py[l] = ndc_to_pixel(y, is_f);
    }
    K.dead = K.px[0] == K.px[2];
    K.a_ok = K.px[1] - K.px[0] != 0.0f;
    K.b_ok = K.px[2] - K.px[1] != 0.0f;
    K.sa = (K.py[1] - K.py[0]) / (K.px[1] - K.px[0]);
    K.sb = (K.py[2] - K.py[1]) / (K.px[2] - K.px[1]);
    K.sc = (K.py[2] - K.py[0]) / (K.px[2] - K.px[0]);
    K.xi_min = cvt_i32_d(fmax((double)ceilf(K.px[0]), 0.));
    K.xi_max = cvt_i32_d(fmin((double)K.px[2], (double)is - 1.));
    return K;
}

// the rows K1 visits in column xi (:160-181; the slope products as written: divide, multiply, add); empty when
// yi_min > yi_max or xi is outside [xi_min, xi_max]
__device__ __forceinline__ void k1_column(const K1Face& K, int xi, int is, int& yi_min, int& yi_max)
{
    yi_min = 0;
    yi_max = -1;
    if (xi < K.xi_min || xi > K.xi_max) return;
    const float fx = (float)xi;
    float yi1;
    if (fx <= K.px[1]) {
        float s = K.sa * (fx - K.px[0]);
        yi1 = K.a_ok ? s + K.py[0] : K.py[1];
    } else {
        float s = K.sb * (fx - K.px[1]);
        yi1 = K.b_ok ? s + K.py[1] : K.py[1];
    }
    float s2 = K.sc * (fx - K.px[0]);
    const float yi2 = s2 + K.py[0];
    yi_min = cvt_i32_d(fmax(0., (double)ceilf(fminf(yi1, yi2))));
    yi_max = cvt_i32_d(fmin((double)fmaxf(yi1, yi2), (double)is - 1.));
}

// order-preserving map float -> uint32 (all finite floats, negative included)
__device__ __forceinline__ uint32_t ord_bits(float v)
{
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord_unbits(uint32_t o)
{
    const uint32_t b = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(b);
}

// rasterize.py:398-423 -- trilinear sampling of a per-face ts^3 texture.  `texel(isc, k)` fetches channel
// k of texel isc; returns the 8 (index, weight) pairs through out arrays when requested.
struct TexCoord {
    int base[3];
    float frac[3];
};

__device__ __forceinline__ TexCoord texture_coord(const float w[3], float depth, float z0, float z1, float z2,
                                                  int ts, double eps)
{
    TexCoord tc;
    const float z[3] = {z0, z1, z2};
    const double c = (double)(ts - 1) - eps;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float q = depth / z[k];
        const float tif = (float)(((double)w[k] * c) * (double)q);
        const int b = cvt_i32(tif);
        tc.base[k] = b;
        tc.frac[k] = tif - (float)b;
    }
    return tc;
}

__device__ __forceinline__ void texture_corner(const TexCoord& tc, int pn, int ts, int& isc, float& wgt)
{
    float w = 1.0f;
    int idx[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (((pn >> k) & 1) == 0) {
            w = w * (1.0f - tc.frac[k]);
            idx[k] = tc.base[k];
        } else {
            w = w * tc.frac[k];
            idx[k] = tc.base[k] + 1;
        }
    }
    isc = idx[0] * ts * ts + idx[1] * ts + idx[2];
    wgt = w;
}

}  // namespace sdn
