// Backward of the rasterizer for gfx950.
//
// Replaces Rasterize.backward_gpu (/root/reference/geometric/neural_renderer/rasterize.py:846-886):
//   k_bwd_edges    K5, rasterize.py:523-745 -- the hand-crafted silhouette / colour gradient wrt the x,y of
//                  every front face's vertices.  One thread per face with private accumulators, exactly the
//                  reference's summation order, so the result is deterministic and bit-comparable with the
//                  CPU oracle.
//   k_bwd_pixels   K6 + K7, rasterize.py:756-789 and 800-844 -- per covered pixel: scatter the colour
//                  gradient into the face's texture (or per-face colour) and the depth gradient into the
//                  winning face's 9 coordinates (hardware float atomics; the reference uses atomicAdd too).
// The upstream gradients arrive at output resolution (after vertical flip and 2x2 average pooling,
// rasterize.py:951-966); they are expanded on the fly: g_ss(y, x) = 0.25 * g_out(R-1-y/2 .., x/2).
#include "raster_math.h"
#include "sdn_common.h"

namespace sdn {

struct BwdParams {
    const float* faces;
    const float* textures;
    const float* face_inv;
    const int32_t* face_index_map;
    const float* weight_map;
    const float* depth_map;
    const float* rgb_map;
    const float* g_rgb_out;
    const float* g_alpha_out;
    const float* g_depth_out;
    float* grad_faces;
    float* grad_textures;
    double eps;
    int ts, bs, nf, S, flags;
};

struct MapReader {
    const BwdParams& P;
    int b, S, R;
    bool aa;
    __device__ MapReader(const BwdParams& p, int bn) : P(p), b(bn), S(p.S)
    {
        aa = (p.flags & SDN_AA) != 0;
        R = aa ? S / 2 : S;
    }
    __device__ __forceinline__ size_t q(int x, int y) const { return ((size_t)b * S + y) * S + x; }
    __device__ __forceinline__ int fidx(int x, int y) const { return P.face_index_map[q(x, y)]; }
    __device__ __forceinline__ float alpha(int x, int y) const { return fidx(x, y) >= 0 ? 1.0f : 0.0f; }
    __device__ __forceinline__ float rgb(int x, int y, int k) const { return P.rgb_map[q(x, y) * 3 + k]; }
    __device__ __forceinline__ size_t oidx(int x, int y) const
    {
        return aa ? ((size_t)((S - 1 - y) >> 1)) * R + (x >> 1) : ((size_t)(S - 1 - y)) * R + x;
    }
    __device__ __forceinline__ float g_alpha(int x, int y) const
    {
        if (!P.g_alpha_out) return 0.0f;
        const float g = P.g_alpha_out[(size_t)b * R * R + oidx(x, y)];
        return aa ? g * 0.25f : g;
    }
    __device__ __forceinline__ float g_depth(int x, int y) const
    {
        if (!P.g_depth_out) return 0.0f;
        const float g = P.g_depth_out[(size_t)b * R * R + oidx(x, y)];
        return aa ? g * 0.25f : g;
    }
    __device__ __forceinline__ float g_rgb(int x, int y, int k) const
    {
        if (!P.g_rgb_out) return 0.0f;
        const float g = P.g_rgb_out[((size_t)b * 3 + k) * R * R + oidx(x, y)];
        return aa ? g * 0.25f : g;
    }
};

// rasterize.py:646-647 -- (p1.d0 - p0.d0) / denom * (d1 - d1_cross) * 2. / is, then +/- eps in double
__device__ __forceinline__ float edge_dist(float pa, float pb, float denom, int d1, float d1_cross, float is_f,
                                           double eps)
{
    float t = (pb - pa) / denom;
    t = t * ((float)d1 - d1_cross);
    float dist = (2.0f * t) / is_f;  // == (float)((double)t * 2. / is): exact doubling, correctly rounded divide
    dist = (0.0f < dist) ? (float)((double)dist + eps) : (float)((double)dist - eps);
    return dist;
}

__global__ __launch_bounds__(256) void k_bwd_edges(const BwdParams P)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)P.bs * P.nf;
    if (i >= total) return;
    const int bn = (int)(i / P.nf);
    const int fn = (int)(i % P.nf);
    const int S = P.S;
    const float is_f = (float)S;
    const bool use_alpha = (P.flags & SDN_ALPHA) != 0;
    const bool use_rgb = (P.flags & SDN_RGB) != 0;
    const bool accumulate = (P.flags & SDN_ACCUMULATE) != 0;
    float face[9];
#pragma unroll
    for (int k = 0; k < 9; k++) face[k] = P.faces[i * 9 + k];
    float grad_face[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};

    if ((use_alpha || use_rgb) && !is_backface(face)) {
        const MapReader M(P, bn);
        for (int edge_num = 0; edge_num < 3; edge_num++) {
            int pi[3];
            float pp[3][2];
            for (int num = 0; num < 3; num++) pi[num] = (edge_num + num) % 3;
            for (int num = 0; num < 3; num++)
                for (int dim = 0; dim < 2; dim++) pp[num][dim] = ndc_to_pixel(face[3 * pi[num] + dim], is_f);

            for (int axis = 0; axis < 2; axis++) {
                float p[3][2];
                for (int num = 0; num < 3; num++)
                    for (int dim = 0; dim < 2; dim++) p[num][dim] = pp[num][(dim + axis) % 2];
                int direction;
                if (axis == 0)
                    direction = (p[0][0] < p[1][0]) ? -1 : 1;
                else
                    direction = (p[0][0] < p[1][0]) ? 1 : -1;

                const int d0_from = cvt_i32(fmaxf(ceilf(fminf(p[0][0], p[1][0])), 0.0f));
                const int d0_to = cvt_i32(fminf(fmaxf(p[0][0], p[1][0]), is_f - 1.0f));
                for (int d0 = d0_from; d0 <= d0_to; d0++) {
                    const float fd0 = (float)d0;
                    float d1_cross = (p[1][1] - p[0][1]) / (p[1][0] - p[0][0]);
                    d1_cross = d1_cross * (fd0 - p[0][0]);
                    d1_cross = d1_cross + p[0][1];
                    const int d1_in = (0 < direction) ? cvt_i32(floorf(d1_cross)) : cvt_i32(ceilf(d1_cross));
                    const int d1_out = d1_in + direction;
                    if (d1_in < 0 || S <= d1_in) continue;
                    if (d1_out < 0 || S <= d1_out) continue;

                    // pixel (d0, d1) in map coordinates: axis 0 -> x = d0, y = d1; axis 1 -> x = d1, y = d0
                    const int xin = axis == 0 ? d0 : d1_in, yin = axis == 0 ? d1_in : d0;
                    const int xout = axis == 0 ? d0 : d1_out, yout = axis == 0 ? d1_out : d0;
                    float alpha_in = 0.f, alpha_out = 0.f, rgb_in[3] = {0, 0, 0}, rgb_out[3] = {0, 0, 0};
                    const int f_in = M.fidx(xin, yin);
                    if (use_alpha) {
                        alpha_in = f_in >= 0 ? 1.0f : 0.0f;
                        alpha_out = M.alpha(xout, yout);
                    }
                    if (use_rgb) {
                        for (int k = 0; k < 3; k++) {
                            rgb_in[k] = M.rgb(xin, yin, k);
                            rgb_out[k] = M.rgb(xout, yout, k);
                        }
                    }
                    const bool nz1 = p[1][0] != fd0, nz0 = p[0][0] != fd0;
                    const float den1 = p[1][0] - fd0, den0 = fd0 - p[0][0];

                    // "out" pass: from the pixel just outside the edge to the image border
                    if (f_in == fn) {
                        const int d1_limit = (0 < direction) ? S - 1 : 0;
                        const int d1_from = max(min(d1_out, d1_limit), 0);
                        const int d1_to = min(max(d1_out, d1_limit), S - 1);
                        for (int d1 = d1_from; d1 <= d1_to; d1++) {
                            const int x = axis == 0 ? d0 : d1, y = axis == 0 ? d1 : d0;
                            float diff_grad = 0.0f;
                            if (use_alpha) diff_grad = diff_grad + (M.alpha(x, y) - alpha_in) * M.g_alpha(x, y);
                            if (use_rgb)
                                for (int k = 0; k < 3; k++)
                                    diff_grad = diff_grad + (M.rgb(x, y, k) - rgb_in[k]) * M.g_rgb(x, y, k);
                            if (diff_grad <= 0) continue;
                            if (nz1) {
                                const float dist = edge_dist(p[0][0], p[1][0], den1, d1, d1_cross, is_f, P.eps);
                                grad_face[pi[0] * 3 + (1 - axis)] -= diff_grad / dist;
                            }
                            if (nz0) {
                                const float dist = edge_dist(p[0][0], p[1][0], den0, d1, d1_cross, is_f, P.eps);
                                grad_face[pi[1] * 3 + (1 - axis)] -= diff_grad / dist;
                            }
                        }
                    }
                    // "in" pass: across the face to its opposite edge
                    {
                        float d0_cross2;
                        if ((fd0 - p[0][0]) * (fd0 - p[2][0]) < 0) {
                            d0_cross2 = (p[2][1] - p[0][1]) / (p[2][0] - p[0][0]);
                            d0_cross2 = d0_cross2 * (fd0 - p[0][0]);
                            d0_cross2 = d0_cross2 + p[0][1];
                        } else {
                            d0_cross2 = (p[1][1] - p[2][1]) / (p[1][0] - p[2][0]);
                            d0_cross2 = d0_cross2 * (fd0 - p[2][0]);
                            d0_cross2 = d0_cross2 + p[2][1];
                        }
                        const int d1_limit = (0 < direction) ? cvt_i32(ceilf(d0_cross2)) : cvt_i32(floorf(d0_cross2));
                        const int d1_from = max(min(d1_in, d1_limit), 0);
                        const int d1_to = min(max(d1_in, d1_limit), S - 1);
                        for (int d1 = d1_from; d1 <= d1_to; d1++) {
                            const int x = axis == 0 ? d0 : d1, y = axis == 0 ? d1 : d0;
                            if (M.fidx(x, y) != fn) continue;
                            float diff_grad = 0.0f;
                            if (use_alpha) diff_grad = diff_grad + (1.0f - alpha_out) * M.g_alpha(x, y);
                            if (use_rgb)
                                for (int k = 0; k < 3; k++)
                                    diff_grad = diff_grad + (M.rgb(x, y, k) - rgb_out[k]) * M.g_rgb(x, y, k);
                            if (diff_grad <= 0) continue;
                            if (nz1) {
                                const float dist = edge_dist(p[0][0], p[1][0], den1, d1, d1_cross, is_f, P.eps);
                                grad_face[pi[0] * 3 + (1 - axis)] -= diff_grad / dist;
                            }
                            if (nz0) {
                                const float dist = edge_dist(p[0][0], p[1][0], den0, d1, d1_cross, is_f, P.eps);
                                grad_face[pi[1] * 3 + (1 - axis)] -= diff_grad / dist;
                            }
                        }
                    }
                }
            }
        }
    }
    if (accumulate) {
#pragma unroll
        for (int k = 0; k < 9; k++) P.grad_faces[i * 9 + k] += grad_face[k];
    } else {
#pragma unroll
        for (int k = 0; k < 9; k++) P.grad_faces[i * 9 + k] = grad_face[k];
    }
}

__global__ __launch_bounds__(256) void k_bwd_pixels(const BwdParams P)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long npx = (long)P.bs * P.S * P.S;
    if (i >= npx) return;
    const int fn = P.face_index_map[i];
    if (fn < 0) return;
    const int S = P.S;
    const int bn = (int)(i / ((long)S * S));
    const int pn = (int)(i % ((long)S * S));
    const int yi = pn / S, xi = pn % S;
    const MapReader M(P, bn);
    const size_t fidx = (size_t)bn * P.nf + fn;
    float face[9];
#pragma unroll
    for (int k = 0; k < 9; k++) face[k] = P.faces[fidx * 9 + k];
    const float depth = P.depth_map[i];
    const float w[3] = {P.weight_map[i * 3 + 0], P.weight_map[i * 3 + 1], P.weight_map[i * 3 + 2]};

    if ((P.flags & SDN_DEPTH) && P.g_depth_out) {
        // K7, rasterize.py:807-834
        float inv[9];
#pragma unroll
        for (int k = 0; k < 9; k++) inv[k] = P.face_inv[fidx * 9 + k];
        const float g = M.g_depth(xi, yi);
        const float depth2 = depth * depth;
        float* gf = P.grad_faces + fidx * 9;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float zk = face[3 * k + 2];
            float t = g * w[k];
            t = t * depth2;
            t = t / (zk * zk);
            unsafeAtomicAdd(&gf[3 * k + 2], t);
        }
        float tmp[2] = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 2; k++)
#pragma unroll
            for (int l = 0; l < 3; l++) tmp[k] = tmp[k] + (-inv[3 * l + k] / face[3 * l + 2]);
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int l = 0; l < 2; l++) {
                float t = -g * tmp[l];
                t = t * w[k];
                t = t * depth2;
                t = t * (float)S;
                t = t / 2.0f;
                unsafeAtomicAdd(&gf[3 * k + l], t);
            }
    }
    if ((P.flags & SDN_RGB) && P.g_rgb_out && P.grad_textures) {
        // K6, rasterize.py:762-778, with the sampling weights recomputed instead of stored (64 B/pixel saved)
        const bool face_color = (P.flags & SDN_FACE_COLOR) != 0;
        const int ts = face_color ? 2 : P.ts;
        const TexCoord tc = texture_coord(w, depth, face[2], face[5], face[8], ts, P.eps);
        const float g[3] = {M.g_rgb(xi, yi, 0), M.g_rgb(xi, yi, 1), M.g_rgb(xi, yi, 2)};
        if (face_color) {
            float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 8; c++) {
                int isc;
                float wgt;
                texture_corner(tc, c, ts, isc, wgt);
#pragma unroll
                for (int k = 0; k < 3; k++) acc[k] = acc[k] + wgt * g[k];
            }
#pragma unroll
            for (int k = 0; k < 3; k++) unsafeAtomicAdd(&P.grad_textures[fidx * 3 + k], acc[k]);
        } else {
            float* gt = P.grad_textures + fidx * (size_t)ts * ts * ts * 3;
#pragma unroll
            for (int c = 0; c < 8; c++) {
                int isc;
                float wgt;
                texture_corner(tc, c, ts, isc, wgt);
#pragma unroll
                for (int k = 0; k < 3; k++) unsafeAtomicAdd(&gt[(long)isc * 3 + k], wgt * g[k]);
            }
        }
    }
}

}  // namespace sdn

using namespace sdn;

SDN_API int sdn_rasterize_bwd(const float* faces, const float* textures, int ts, int bs, int nf, int S, double eps,
                              int flags, const float* face_inv, const int32_t* face_index_map,
                              const float* weight_map, const float* depth_map, const float* rgb_map,
                              const float* g_rgb_out, const float* g_alpha_out, const float* g_depth_out,
                              float* grad_faces, float* grad_textures, sdnStream stream)
{
    if (!faces || !face_inv || !face_index_map || !weight_map || !depth_map || !grad_faces || bs <= 0 || nf <= 0 ||
        S <= 0)
        return fail(SDN_EINVAL, "sdn_rasterize_bwd: missing state (was the forward run with SDN_SAVE_MAPS?)");
    if ((flags & SDN_RGB) && (!rgb_map || !textures))
        return fail(SDN_EINVAL, "sdn_rasterize_bwd: rgb gradients need rgb_map and textures");
    if ((flags & SDN_AA) && (S & 1)) return fail(SDN_EINVAL, "sdn_rasterize_bwd: SDN_AA needs an even internal size");
    hipStream_t st = (hipStream_t)stream;
    BwdParams P;
    P.faces = faces;
    P.textures = textures;
    P.face_inv = face_inv;
    P.face_index_map = face_index_map;
    P.weight_map = weight_map;
    P.depth_map = depth_map;
    P.rgb_map = rgb_map;
    P.g_rgb_out = (flags & SDN_RGB) ? g_rgb_out : nullptr;
    P.g_alpha_out = (flags & SDN_ALPHA) ? g_alpha_out : nullptr;
    P.g_depth_out = (flags & SDN_DEPTH) ? g_depth_out : nullptr;
    P.grad_faces = grad_faces;
    P.grad_textures = grad_textures;
    P.eps = eps;
    P.ts = ts;
    P.bs = bs;
    P.nf = nf;
    P.S = S;
    P.flags = flags;
    // the reference zero-fills a missing upstream gradient (rasterize.py:855-875); a NULL g_* therefore simply
    // disables that term, but K5 still owns the store of grad_faces
    if (!P.g_rgb_out) P.flags &= ~SDN_RGB;
    if (!P.g_alpha_out) P.flags &= ~SDN_ALPHA;
    const long total = (long)bs * nf;
    hipLaunchKernelGGL(k_bwd_edges, dim3(cdiv(total, 256)), dim3(256), 0, st, P);
    int rc = check_launch("k_bwd_edges");
    if (rc) return rc;
    const bool need_tex = (P.flags & SDN_RGB) && grad_textures;
    if (need_tex && !(flags & SDN_ACCUMULATE)) {
        const size_t n = (flags & SDN_FACE_COLOR) ? (size_t)total * 3 : (size_t)total * ts * ts * ts * 3;
        hipError_t e = hipMemsetAsync(grad_textures, 0, n * sizeof(float), st);
        if (e != hipSuccess) return fail(SDN_ELAUNCH, "hipMemsetAsync(grad_textures): %s", hipGetErrorString(e));
    }
    if (((P.flags & SDN_DEPTH) && P.g_depth_out) || need_tex) {
        const long npx = (long)bs * S * S;
        hipLaunchKernelGGL(k_bwd_pixels, dim3(cdiv(npx, 256)), dim3(256), 0, st, P);
        rc = check_launch("k_bwd_pixels");
        if (rc) return rc;
    }
    return SDN_OK;
}
