// PerspectiveTransform of derender3d for a whole frame's objects in two passes per direction.
//
// Reference: /root/reference/geometric/derender3d/models/transforms.py:102-158 -- scale, quaternion rotation, translation,
// shear that moves the reference ray (x0, y0, z0) onto the optical axis, zoom-to-fit
//     zoom[b] = min_v |z| / max(|x|, |y|) * zoom_to[b],   z <- z / zoom
// run there as ~25 element-wise torch ops and a batched [V,3]x[3,3] GEMM per call (and twice that in backward).  Here:
//   k_ptf_fwd_a   per vertex: scale, rotate (R(q) built per thread from the 4 quaternion floats), translate, shear; writes
//                 (x, y, z) and reduces  key = float_bits(|z| / max(|x|,|y|)) << 32 | v  with a 64-bit atomicMin
//                 (ratios are >= 0, so their bit patterns order like the values; the vertex index rides along for the
//                 backward pass);
//   k_ptf_fwd_b   per vertex: z /= zoom;  zoom[b] itself is written by vertex 0's thread;
//   k_ptf_bwd_a   per object: S = sum_v gz_v * z_v (z after zoom)  ->  d loss / d zoom = -S / zoom + upstream;
//   k_ptf_bwd_b   per vertex: gradient through zoom (the argmin vertex also receives d zoom), shear, translation,
//                 rotation, scale; the 18 per-object parameter sums (translation 3, reference ray 3, R 9, scale 3) are
//                 block-reduced and added with float atomics; the R gradient is folded into the quaternion's by
//   k_ptf_bwd_c   (one thread per object).
// Arithmetic follows the torch expression order (this file is built without FMA contraction), so vertices agree with
// the torch path to the last bit except for the 3x3 product, whose summation order inside a BLAS GEMM is not defined.
#include "sdn_common.h"

namespace sdn {

struct PtfParams {
    const float* verts;   // [n, V, 3]
    const float* scales;  // [n, 3]
    const float* quat;    // [n, 4]  (a, b, c, d)
    const float* trans;   // [n, 3]
    const float* persp;   // [n, 3]  reference ray (x0, y0, z0)
    const float* zoom_to; // [n]
    const float* zoom_fixed;  // [n] or null: the training form (transforms.py:150): z /= zoom_fixed, no zoom-to-fit
    float* out;           // [n, V, 3]
    float* zooms;         // [n]
    unsigned long long* key;  // [n] min ratio bits << 32 | vertex, then [n, ceil(V / 256)] per-block minima (test-time form)
    int n, V;
};

__device__ __forceinline__ void quat_matrix(const float* q, float T[9])
{
    const float a = q[0], b = q[1], c = q[2], d = q[3];
    T[0] = ((a * a + b * b) - c * c) - d * d;
    T[1] = 2 * b * c - 2 * a * d;
    T[2] = 2 * b * d + 2 * a * c;
    T[3] = 2 * b * c + 2 * a * d;
    T[4] = ((a * a - b * b) + c * c) - d * d;
    T[5] = 2 * c * d - 2 * a * b;
    T[6] = 2 * b * d - 2 * a * c;
    T[7] = 2 * c * d + 2 * a * b;
    T[8] = ((a * a - b * b) - c * c) + d * d;
}

__global__ __launch_bounds__(256) void k_ptf_fwd_a(const PtfParams P)
{
    __shared__ unsigned long long red[4];
    const int b = blockIdx.y;
    const int v = blockIdx.x * 256 + threadIdx.x;
    unsigned long long key = ~0ull;
    if (v < P.V) {
        float T[9];
        quat_matrix(P.quat + 4 * b, T);
        const float* s = P.scales + 3 * b;
        const float* t = P.trans + 3 * b;
        const float* p0 = P.persp + 3 * b;
        const float* vin = P.verts + ((size_t)b * P.V + v) * 3;
        const float u0 = vin[0] * s[0], u1 = vin[1] * s[1], u2 = vin[2] * s[2];
        const float w0 = ((u0 * T[0] + u1 * T[1]) + u2 * T[2]) + t[0];
        const float w1 = ((u0 * T[3] + u1 * T[4]) + u2 * T[5]) + t[1];
        const float w2 = ((u0 * T[6] + u1 * T[7]) + u2 * T[8]) + t[2];
        const float x = w0 - p0[0] / p0[2] * w2;
        const float y = w1 - p0[1] / p0[2] * w2;
        float* o = P.out + ((size_t)b * P.V + v) * 3;
        o[0] = x;
        o[1] = y;
        if (P.zoom_fixed) {
            // given zoom: one pass.  The backward kernels read it back from `key` with an argmin vertex that matches no
            // thread (0xffffffff) and zoom_to = 1.
            const float zoom = P.zoom_fixed[b];
            o[2] = w2 / zoom;
            if (v == 0) {
                P.key[b] = ((unsigned long long)__float_as_uint(zoom) << 32) | 0xffffffffull;
                P.zooms[b] = zoom;
            }
        } else {
            o[2] = w2;
            const float r = fabsf(w2) / fmaxf(fabsf(x), fabsf(y));
            if (r == r) key = ((unsigned long long)__float_as_uint(r) << 32) | (unsigned)v;  // NaN (0/0) never wins
        }
    }
    if (P.zoom_fixed) return;  // uniform over the grid
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(key, o, 64);
        key = other < key ? other : key;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        // one minimum per block, combined by every block of k_ptf_fwd_b: nothing to initialise (r04: was an atomicMin on
        // key[b] behind a hipMemsetAsync -- a 5 us fill launch in a 1 ms frame step)
        unsigned long long k = red[0];
        for (int i = 1; i < 4; i++) k = red[i] < k ? red[i] : k;
        P.key[P.n + (size_t)b * gridDim.x + blockIdx.x] = k;
    }
}

__global__ __launch_bounds__(256) void k_ptf_fwd_b(const PtfParams P)
{
    __shared__ unsigned long long red[4];
    const int b = blockIdx.y;
    const int v = blockIdx.x * 256 + threadIdx.x;
    unsigned long long key = ~0ull;
    for (unsigned i = threadIdx.x; i < gridDim.x; i += 256) {
        const unsigned long long k = P.key[P.n + (size_t)b * gridDim.x + i];
        key = k < key ? k : key;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(key, o, 64);
        key = other < key ? other : key;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = key;
    __syncthreads();
    key = red[0];
    for (int i = 1; i < 4; i++) key = red[i] < key ? red[i] : key;
    if (v >= P.V) return;
    const float zoom = __uint_as_float((unsigned)(key >> 32)) * P.zoom_to[b];
    float* o = P.out + ((size_t)b * P.V + v) * 3;
    o[2] = o[2] / zoom;
    if (v == 0) {
        P.zooms[b] = zoom;
        P.key[b] = key;     // what the backward pass reads
    }
}

struct PtfBwdParams {
    PtfParams f;
    const float* out;      // forward output [n, V, 3] (z after zoom)
    const float* g_out;    // [n, V, 3]
    const float* g_zooms;  // [n] or null
    float* g_verts;        // [n, V, 3]
    float* acc;            // [n, 20]: 0-2 translation, 3-5 reference ray, 6-14 R, 15-17 scale; then [n, 16] block sums of S = sum gz*z
    float* g_scales;       // [n, 3]
    float* g_quat;         // [n, 4]
    float* g_trans;        // [n, 3]
    float* g_persp;        // [n, 3]
    float* g_zoom_to;      // [n]
    int s_parts;           // blocks per object of k_ptf_bwd_a (<= 16)
};

__global__ __launch_bounds__(256) void k_ptf_bwd_a(const PtfBwdParams B)
{
    __shared__ float red[4];
    const int b = blockIdx.y, V = B.f.V;
    float s = 0.f;
    for (int v = blockIdx.x * 256 + threadIdx.x; v < V; v += gridDim.x * 256) {
        const size_t i = ((size_t)b * V + v) * 3 + 2;
        s += B.g_out[i] * B.out[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    // one partial of S per block (ptf_S adds them in block order) and, from block 0, the zeros k_ptf_bwd_b accumulates onto:
    // no memset launch, a fixed summation order for S
    if (threadIdx.x == 0) B.acc[20 * (size_t)B.f.n + 16 * b + blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
    if (blockIdx.x == 0 && threadIdx.x < 20) B.acc[20 * b + threadIdx.x] = 0.f;
}

__device__ __forceinline__ float ptf_S(const PtfBwdParams& B, int b, int nparts)
{
    float s = 0.f;
    for (int i = 0; i < nparts; i++) s += B.acc[20 * (size_t)B.f.n + 16 * b + i];
    return s;
}

__global__ __launch_bounds__(256) void k_ptf_bwd_b(const PtfBwdParams B)
{
    __shared__ float red[4][18];
    const int b = blockIdx.y, V = B.f.V;
    const int v = blockIdx.x * 256 + threadIdx.x;
    float part[18];
#pragma unroll
    for (int k = 0; k < 18; k++) part[k] = 0.f;
    if (v < V) {
        float T[9];
        quat_matrix(B.f.quat + 4 * b, T);
        const float* sc = B.f.scales + 3 * b;
        const float* p0 = B.f.persp + 3 * b;
        const unsigned long long key = B.f.key[b];
        const float rmin = __uint_as_float((unsigned)(key >> 32));
        const int vstar = (int)(unsigned)(key & 0xffffffffull);
        const float zoom = rmin * B.f.zoom_to[b];
        // d loss / d zoom: z_out = Z / zoom  =>  -sum gz * Z / zoom^2 = -S / zoom with S = sum gz * z_out
        float dzoom = -ptf_S(B, b, B.s_parts) / zoom;
        if (B.g_zooms) dzoom += B.g_zooms[b];
        const size_t i = ((size_t)b * V + v) * 3;
        const float X = B.out[i], Y = B.out[i + 1], Z = B.out[i + 2] * zoom;
        float gX = B.g_out[i], gY = B.g_out[i + 1], gZ = B.g_out[i + 2] / zoom;
        if (v == vstar) {  // zoom = |Z*| / max(|X*|, |Y*|) * zoom_to
            const float dr = dzoom * B.f.zoom_to[b];
            const float ax = fabsf(X), ay = fabsf(Y);
            const float m = fmaxf(ax, ay);
            gZ += dr * (Z >= 0.f ? 1.f : -1.f) / m;
            const float dm = -dr * fabsf(Z) / (m * m);
            if (ax >= ay)
                gX += dm * (X >= 0.f ? 1.f : -1.f);
            else
                gY += dm * (Y >= 0.f ? 1.f : -1.f);
        }
        // shear: x = w0 - x0 / z0 * w2,  y = w1 - y0 / z0 * w2
        const float kx = p0[0] / p0[2], ky = p0[1] / p0[2];
        const float gw0 = gX, gw1 = gY, gw2 = (gZ - kx * gX) - ky * gY;
        part[3] = -gX * Z / p0[2];
        part[4] = -gY * Z / p0[2];
        part[5] = (gX * p0[0] + gY * p0[1]) * Z / (p0[2] * p0[2]);
        part[0] = gw0;
        part[1] = gw1;
        part[2] = gw2;
        const float* vin = B.f.verts + i;
        const float u0 = vin[0] * sc[0], u1 = vin[1] * sc[1], u2 = vin[2] * sc[2];
        part[6] = gw0 * u0;
        part[7] = gw0 * u1;
        part[8] = gw0 * u2;
        part[9] = gw1 * u0;
        part[10] = gw1 * u1;
        part[11] = gw1 * u2;
        part[12] = gw2 * u0;
        part[13] = gw2 * u1;
        part[14] = gw2 * u2;
        const float gu0 = (T[0] * gw0 + T[3] * gw1) + T[6] * gw2;
        const float gu1 = (T[1] * gw0 + T[4] * gw1) + T[7] * gw2;
        const float gu2 = (T[2] * gw0 + T[5] * gw1) + T[8] * gw2;
        part[15] = gu0 * vin[0];
        part[16] = gu1 * vin[1];
        part[17] = gu2 * vin[2];
        float* gv = B.g_verts + i;
        gv[0] = gu0 * sc[0];
        gv[1] = gu1 * sc[1];
        gv[2] = gu2 * sc[2];
    }
#pragma unroll
    for (int k = 0; k < 18; k++) {
        float s = part[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < 18)
        unsafeAtomicAdd(B.acc + 20 * b + threadIdx.x,
                        ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x]);
}

__global__ void k_ptf_bwd_c(const PtfBwdParams B)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B.f.n) return;
    const float* A = B.acc + 20 * b;
    const float* q = B.f.quat + 4 * b;
    const float a = q[0], bq = q[1], c = q[2], d = q[3];
    const float* g = A + 6;  // row-major d loss / d R
    B.g_quat[4 * b + 0] = 2 * a * ((g[0] + g[4]) + g[8]) + 2 * (((((-d * g[1] + c * g[2]) + d * g[3]) - bq * g[5]) - c * g[6]) + bq * g[7]);
    B.g_quat[4 * b + 1] = 2 * bq * ((g[0] - g[4]) - g[8]) + 2 * (((((c * g[1] + d * g[2]) + c * g[3]) - a * g[5]) + d * g[6]) + a * g[7]);
    B.g_quat[4 * b + 2] = 2 * c * ((-g[0] + g[4]) - g[8]) + 2 * (((((bq * g[1] + a * g[2]) + bq * g[3]) + d * g[5]) - a * g[6]) + d * g[7]);
    B.g_quat[4 * b + 3] = 2 * d * ((-g[0] - g[4]) + g[8]) + 2 * (((((-a * g[1] + bq * g[2]) + a * g[3]) + c * g[5]) + bq * g[6]) + c * g[7]);
    for (int k = 0; k < 3; k++) {
        if (B.g_persp == B.g_trans) {
            B.g_trans[3 * b + k] = A[k] + A[3 + k];   // one tensor passed for both arguments: its gradient is the sum (r06)
        } else {
            B.g_trans[3 * b + k] = A[k];
            B.g_persp[3 * b + k] = A[3 + k];
        }
        B.g_scales[3 * b + k] = A[15 + k];
    }
    const unsigned long long key = B.f.key[b];
    const float rmin = __uint_as_float((unsigned)(key >> 32));
    const float zoom = rmin * B.f.zoom_to[b];
    float dzoom = -ptf_S(B, b, B.s_parts) / zoom;
    if (B.g_zooms) dzoom += B.g_zooms[b];
    B.g_zoom_to[b] = dzoom * rmin;
}

}  // namespace sdn

using namespace sdn;

SDN_API int sdn_perspective_transform_scratch(int n, int V, size_t* key_bytes, size_t* acc_bytes)
{
    if (n <= 0 || V <= 0 || (!key_bytes && !acc_bytes))
        return fail(SDN_EINVAL, "sdn_perspective_transform_scratch: bad arguments");
    if (key_bytes) *key_bytes = (size_t)n * (1 + (size_t)cdiv(V, 256)) * sizeof(unsigned long long);
    if (acc_bytes) *acc_bytes = (size_t)n * 36 * sizeof(float);
    return SDN_OK;
}

SDN_API int sdn_perspective_transform(const float* verts, const float* scales, const float* quat, const float* trans,
                                      const float* persp, const float* zoom_to, const float* zoom_fixed, int n, int V,
                                      float* out, float* zooms, void* key, sdnStream stream)
{
    if (!verts || !scales || !quat || !trans || !persp || (!zoom_to && !zoom_fixed) || !out || !zooms || !key || n <= 0 ||
        V <= 0)
        return fail(SDN_EINVAL, "sdn_perspective_transform: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    PtfParams P{verts, scales, quat, trans, persp, zoom_to, zoom_fixed, out, zooms, (unsigned long long*)key, n, V};
    const dim3 grid(cdiv(V, 256), (unsigned)n);
    if (zoom_fixed) {
        hipLaunchKernelGGL(k_ptf_fwd_a, grid, dim3(256), 0, st, P);
        return check_launch("k_ptf_fwd (given zoom)");
    }
    hipLaunchKernelGGL(k_ptf_fwd_a, grid, dim3(256), 0, st, P);
    hipLaunchKernelGGL(k_ptf_fwd_b, grid, dim3(256), 0, st, P);
    return check_launch("k_ptf_fwd");
}

SDN_API int sdn_perspective_transform_bwd(const float* verts, const float* scales, const float* quat, const float* trans,
                                          const float* persp, const float* zoom_to, int n, int V, const float* out,
                                          const void* key, const float* g_out, const float* g_zooms, float* g_verts,
                                          float* g_scales, float* g_quat, float* g_trans, float* g_persp,
                                          float* g_zoom_to, float* acc, sdnStream stream)
{
    if (!verts || !scales || !quat || !trans || !persp || !zoom_to || !out || !key || !g_out || !g_verts || !g_scales ||
        !g_quat || !g_trans || !g_persp || !g_zoom_to || !acc || n <= 0 || V <= 0)
        return fail(SDN_EINVAL, "sdn_perspective_transform_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    PtfBwdParams B;
    B.f = PtfParams{verts, scales, quat, trans, persp, zoom_to, nullptr, nullptr, nullptr, (unsigned long long*)key, n, V};
    B.out = out;
    B.g_out = g_out;
    B.g_zooms = g_zooms;
    B.g_verts = g_verts;
    B.acc = acc;
    B.g_scales = g_scales;
    B.g_quat = g_quat;
    B.g_trans = g_trans;
    B.g_persp = g_persp;
    B.g_zoom_to = g_zoom_to;
    const unsigned vb = cdiv(V, 256);
    B.s_parts = (int)(vb < 16 ? vb : 16);
    hipLaunchKernelGGL(k_ptf_bwd_a, dim3((unsigned)B.s_parts, (unsigned)n), dim3(256), 0, st, B);
    hipLaunchKernelGGL(k_ptf_bwd_b, dim3(vb, (unsigned)n), dim3(256), 0, st, B);
    hipLaunchKernelGGL(k_ptf_bwd_c, dim3(cdiv(n, 64)), dim3(64), 0, st, B);
    return check_launch("k_ptf_bwd");
}

// ---- pose parameters of a frame's objects (Derenderer3d.render, /root/reference/geometric/derender3d/models/__init__.py:106-116):
//     rotations = (cos(theta / 2), 0, sin(theta / 2), 0),   scales = exp(log_scales)
// five torch launches forward (div, cos, sin, cat, exp) and as many again backward, for 16 objects; one launch each way here.
namespace sdn {

__global__ __launch_bounds__(64) void k_pose_params(const float* __restrict__ theta, const float* __restrict__ log_scales, int n,
                                                    float* __restrict__ quat, float* __restrict__ scales)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const float h = theta[i] / 2;
    quat[4 * i + 0] = cosf(h);
    quat[4 * i + 1] = 0.f;
    quat[4 * i + 2] = sinf(h);
    quat[4 * i + 3] = 0.f;
#pragma unroll
    for (int d = 0; d < 3; d++) scales[3 * i + d] = expf(log_scales[3 * i + d]);
}

// g_theta = (-sin(theta / 2) g_quat[0] + cos(theta / 2) g_quat[2]) / 2,   g_log_scales = g_scales * scales
__global__ __launch_bounds__(64) void k_pose_params_bwd(const float* __restrict__ theta, const float* __restrict__ scales,
                                                        const float* __restrict__ g_quat, const float* __restrict__ g_scales, int n,
                                                        float* __restrict__ g_theta, float* __restrict__ g_log_scales)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    if (g_theta) {
        const float h = theta[i] / 2;
        const float a = g_quat ? g_quat[4 * i + 0] : 0.f, c = g_quat ? g_quat[4 * i + 2] : 0.f;
        g_theta[i] = (a * (-sinf(h)) + c * cosf(h)) / 2;
    }
    if (g_log_scales) {
#pragma unroll
        for (int d = 0; d < 3; d++) g_log_scales[3 * i + d] = g_scales ? g_scales[3 * i + d] * scales[3 * i + d] : 0.f;
    }
}

// ---- the whole pose algebra of Derenderer3d.render (derender3d/models/__init__.py:95-158), one thread per object.
// The reference spells it as ~45 element-wise torch ops (atan2, exp, sqrt, stack, norm, atan, remainder, ...) and autograd
// runs as many again backward: ~100 launches of 16 elements per optimisation iteration, 0.7 ms of host time and 0.4 ms of GPU
// time around a 0.9 ms render.  Same formulas, fp32, libm's atan2f / atanf / expf / sqrtf / cosf / sinf.
struct PoseAlgebra {
    const float *centre, *extent, *focals, *delta, *log_scales, *log_depths, *t2;   // [n,2] [n,2] [n] [n,2] [n,3] [n] [n,2]
    float *theta, *alpha, *quat, *scales, *depth, *c2, *trans, *persp, *zoom;        // [n] [n] [n,4] [n,3] [n] [n,2] [n,3] [n,3] [n]
    int n, training;
    float image_size, render_size;
};

__device__ __forceinline__ void camera_ray(float row, float col, float r[3], float* norm)
{
    // the camera looks down -z with +y up: the pixel at normalised (row, column) lies along (column, -row, -1)  (:119-126)
    const float l = sqrtf((col * col + row * row) + 1.0f);
    r[0] = col / l;
    r[1] = -row / l;
    r[2] = -1.0f / l;
    *norm = l;
}

__global__ __launch_bounds__(64) void k_pose_algebra(const PoseAlgebra P)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= P.n) return;
    const float d0 = P.delta[2 * i], d1 = P.delta[2 * i + 1];
    const float theta = atan2f(d1, d0);
    P.theta[i] = theta;
    const float h = theta / 2;
    P.quat[4 * i + 0] = cosf(h);
    P.quat[4 * i + 1] = 0.f;
    P.quat[4 * i + 2] = sinf(h);
    P.quat[4 * i + 3] = 0.f;
#pragma unroll
    for (int d = 0; d < 3; d++) P.scales[3 * i + d] = expf(P.log_scales[3 * i + d]);
    const float e0 = P.extent[2 * i], e1 = P.extent[2 * i + 1], c0 = P.centre[2 * i], c1 = P.centre[2 * i + 1];
    const float depth = sqrtf(expf(P.log_depths[i]) / (e0 * e1));
    P.depth[i] = depth;
    const float q0 = c0 + P.t2[2 * i] * e0, q1 = c1 + P.t2[2 * i + 1] * e1;
    P.c2[2 * i] = q0;
    P.c2[2 * i + 1] = q1;
    float r[3], l;
    camera_ray(q0, q1, r, &l);
    const float tx = depth * r[0], ty = depth * r[1], tz = depth * r[2];
    P.trans[3 * i] = tx;
    P.trans[3 * i + 1] = ty;
    P.trans[3 * i + 2] = tz;
    // observation angle: yaw minus the bearing of the object centre, wrapped to [-pi, pi)  (:128-129); torch.remainder takes
    // the divisor's sign (fmod, then one step up for negative results -- ATen's remainder for floats)
    const float a = -(theta - atanf(tx / tz)) + 3.14159265358979323846f;
    const float two_pi = 6.28318530717958647692f;
    float m = fmodf(a, two_pi);
    if (m != 0.f && m < 0.f) m += two_pi;
    P.alpha[i] = m - 3.14159265358979323846f;
    const float f = P.focals[i];
    if (P.training) {   // crop-centred camera with a fixed zoom (:139-150)
        float rc[3], lc;
        camera_ray(c0, c1, rc, &lc);
        P.persp[3 * i] = depth * rc[0];
        P.persp[3 * i + 1] = depth * rc[1];
        P.persp[3 * i + 2] = depth * rc[2];
        P.zoom[i] = (P.image_size / f) / fmaxf(e0, e1);
    } else {            // object-centred camera, zoom-to-fit (:152-153)
        P.persp[3 * i] = tx;
        P.persp[3 * i + 1] = ty;
        P.persp[3 * i + 2] = tz;
        P.zoom[i] = P.render_size / (2.0f * f);
    }
}

struct PoseAlgebraBwd {
    PoseAlgebra f;          // inputs + the forward's outputs (theta, scales, depth, c2, trans are read)
    const float *g_theta, *g_alpha, *g_quat, *g_scales, *g_depth, *g_c2, *g_trans, *g_persp;   // each may be null
    float *g_delta, *g_log_scales, *g_log_depths, *g_t2;                                        // each may be null
};

__global__ __launch_bounds__(64) void k_pose_algebra_bwd(const PoseAlgebraBwd B)
{
    const PoseAlgebra& P = B.f;
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= P.n) return;
    const float theta = P.theta[i], depth = P.depth[i];
    const float tx = P.trans[3 * i], tz = P.trans[3 * i + 2];
    float gT[3] = {0.f, 0.f, 0.f};
    if (B.g_trans)
        for (int d = 0; d < 3; d++) gT[d] += B.g_trans[3 * i + d];
    if (B.g_persp && !P.training)
        for (int d = 0; d < 3; d++) gT[d] += B.g_persp[3 * i + d];
    float g_th = B.g_theta ? B.g_theta[i] : 0.f;
    if (B.g_alpha) {   // alpha = -(theta - atan(tx / tz)) (+ a constant step)
        const float ga = B.g_alpha[i];
        g_th -= ga;
        const float q = tx / tz, k = ga / (1.0f + q * q);
        gT[0] += k / tz;
        gT[2] -= k * q / tz;
    }
    if (B.g_quat) {
        const float h = theta / 2;
        g_th += (B.g_quat[4 * i + 0] * (-sinf(h)) + B.g_quat[4 * i + 2] * cosf(h)) / 2;
    }
    if (B.g_delta) {   // theta = atan2(d1, d0)
        const float d0 = P.delta[2 * i], d1 = P.delta[2 * i + 1], s = d0 * d0 + d1 * d1;
        B.g_delta[2 * i] = g_th * (-d1 / s);
        B.g_delta[2 * i + 1] = g_th * (d0 / s);
    }
    if (B.g_log_scales)
        for (int d = 0; d < 3; d++) B.g_log_scales[3 * i + d] = B.g_scales ? B.g_scales[3 * i + d] * P.scales[3 * i + d] : 0.f;
    // translations = depth * ray(c2)
    float r[3], l;
    camera_ray(P.c2[2 * i], P.c2[2 * i + 1], r, &l);
    float g_depth = B.g_depth ? B.g_depth[i] : 0.f;
    g_depth += (gT[0] * r[0] + gT[1] * r[1]) + gT[2] * r[2];
    if (B.g_persp && P.training) {
        float rc[3], lc;
        camera_ray(P.centre[2 * i], P.centre[2 * i + 1], rc, &lc);
        g_depth += (B.g_persp[3 * i] * rc[0] + B.g_persp[3 * i + 1] * rc[1]) + B.g_persp[3 * i + 2] * rc[2];
    }
    if (B.g_log_depths) B.g_log_depths[i] = g_depth * depth / 2;   // depth = sqrt(exp(ld) / area)
    if (B.g_t2) {
        // r = u / |u|, u = (c2_1, -c2_0, -1):  g_u = (g_r - r (r . g_r)) / |u|
        const float gr0 = depth * gT[0], gr1 = depth * gT[1], gr2 = depth * gT[2];
        const float dot = (r[0] * gr0 + r[1] * gr1) + r[2] * gr2;
        const float gu0 = (gr0 - r[0] * dot) / l, gu1 = (gr1 - r[1] * dot) / l;
        float gq0 = -gu1, gq1 = gu0;      // (u0 = c2_1, u1 = -c2_0)
        if (B.g_c2) {
            gq0 += B.g_c2[2 * i];
            gq1 += B.g_c2[2 * i + 1];
        }
        B.g_t2[2 * i] = gq0 * P.extent[2 * i];
        B.g_t2[2 * i + 1] = gq1 * P.extent[2 * i + 1];
    }
}

}  // namespace sdn

SDN_API int sdn_pose_algebra(const float* centre, const float* extent, const float* focals, const float* theta_deltas,
                             const float* log_scales, const float* log_depths, const float* translation2ds, int n, int training,
                             float image_size, float render_size, float* thetas, float* alphas, float* rotations, float* scales,
                             float* depths, float* center2ds, float* translations, float* persp, float* zooms, sdnStream stream)
{
    if (!centre || !extent || !focals || !theta_deltas || !log_scales || !log_depths || !translation2ds || !thetas || !alphas ||
        !rotations || !scales || !depths || !center2ds || !translations || !persp || !zooms || n < 1)
        return fail(SDN_EINVAL, "sdn_pose_algebra: bad arguments");
    sdn::PoseAlgebra P{centre, extent, focals, theta_deltas, log_scales, log_depths, translation2ds, thetas, alphas, rotations, scales,
                       depths, center2ds, translations, persp, zooms, n, training, image_size, render_size};
    hipLaunchKernelGGL(sdn::k_pose_algebra, dim3(cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream, P);
    return check_launch("k_pose_algebra");
}

SDN_API int sdn_pose_algebra_bwd(const float* centre, const float* extent, const float* theta_deltas, const float* thetas,
                                 const float* scales, const float* depths, const float* center2ds, const float* translations,
                                 int n, int training, const float* g_thetas, const float* g_alphas, const float* g_rotations,
                                 const float* g_scales, const float* g_depths, const float* g_center2ds,
                                 const float* g_translations, const float* g_persp, float* g_theta_deltas, float* g_log_scales,
                                 float* g_log_depths, float* g_translation2ds, sdnStream stream)
{
    if (!centre || !extent || !theta_deltas || !thetas || !scales || !depths || !center2ds || !translations || n < 1 ||
        (!g_theta_deltas && !g_log_scales && !g_log_depths && !g_translation2ds))
        return fail(SDN_EINVAL, "sdn_pose_algebra_bwd: bad arguments");
    sdn::PoseAlgebraBwd B;
    B.f = sdn::PoseAlgebra{centre, extent, nullptr, theta_deltas, nullptr, nullptr, nullptr, const_cast<float*>(thetas), nullptr, nullptr,
                           const_cast<float*>(scales), const_cast<float*>(depths), const_cast<float*>(center2ds),
                           const_cast<float*>(translations), nullptr, nullptr, n, training, 0.f, 0.f};
    B.g_theta = g_thetas; B.g_alpha = g_alphas; B.g_quat = g_rotations; B.g_scales = g_scales; B.g_depth = g_depths;
    B.g_c2 = g_center2ds; B.g_trans = g_translations; B.g_persp = g_persp;
    B.g_delta = g_theta_deltas; B.g_log_scales = g_log_scales; B.g_log_depths = g_log_depths; B.g_t2 = g_translation2ds;
    hipLaunchKernelGGL(sdn::k_pose_algebra_bwd, dim3(cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream, B);
    return check_launch("k_pose_algebra_bwd");
}

SDN_API int sdn_pose_params(const float* theta, const float* log_scales, int n, float* quat, float* scales, sdnStream stream)
{
    if (!theta || !log_scales || !quat || !scales || n < 1) return fail(SDN_EINVAL, "sdn_pose_params: bad arguments");
    hipLaunchKernelGGL(sdn::k_pose_params, dim3(cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream, theta, log_scales, n, quat, scales);
    return check_launch("k_pose_params");
}

SDN_API int sdn_pose_params_bwd(const float* theta, const float* scales, const float* g_quat, const float* g_scales, int n,
                                float* g_theta, float* g_log_scales, sdnStream stream)
{
    if (!theta || !scales || n < 1 || (!g_theta && !g_log_scales)) return fail(SDN_EINVAL, "sdn_pose_params_bwd: bad arguments");
    hipLaunchKernelGGL(sdn::k_pose_params_bwd, dim3(cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream, theta, scales, g_quat, g_scales,
                       n, g_theta, g_log_scales);
    return check_launch("k_pose_params_bwd");
}
