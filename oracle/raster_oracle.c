/*
 * raster_oracle.c -- CPU restatement of the 3D-SDN / neural_renderer rasterizer kernels.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (3d-sdn_amd/) may call this file;
 * it exists so tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg have an
 * independent statement of what the reference computes.
 *
 * Reference: /root/reference/geometric/neural_renderer/rasterize.py (CUDA bodies embedded as
 * Python strings, JIT-compiled by CuPy).  Each function below cites the lines it follows.
 * The arithmetic is restated operation by operation, including the places where the CUDA
 * source silently computes in double (un-suffixed literals such as 0.5, 2., 1., 0.):
 * every intermediate that CUDA would round to float is stored in a `float` here, every
 * intermediate it would keep in double is a `double` here.  Build with
 * -ffp-contract=off (no FMA fusion) and without fast-math: see oracle/Makefile.
 *
 * Parity status: the reference ships no tests or golden vectors for this path
 * (SURVEY.md section 8c), and chainer/cupy are not installable here.  This restatement is
 * pinned instead against oracle/_ref (the reference's own kernel strings, extracted at
 * build time from where they lie and compiled for the CPU by oracle/build_ref.py) in
 * tests/test_oracle_vs_ref.py, and against analytic known-answer tests
 * (tests/test_oracle_kat.py).
 *
 * Conversions: CUDA float/double -> int conversion returns 0 for NaN and saturates;
 * x86 returns INT_MIN.  cuda_f2i() restates the CUDA rule so that pathological inputs
 * (vertical edges at integer pixel coordinates) take the same path.
 */
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

static inline int cuda_f2i(double v)
{
    if (v != v) return 0;
    if (v >= 2147483647.0) return INT_MAX;
    if (v <= -2147483648.0) return INT_MIN;
    return (int)v;
}

/* rasterize.py:120,252,307,537 -- back-face predicate, all float */
static inline int is_backface(const float *f)
{
    const float a = (f[7] - f[1]) * (f[3] - f[0]);
    const float b = (f[4] - f[1]) * (f[6] - f[0]);
    return a < b;
}

/* rasterize.py:138,258,546 -- NDC -> pixel coordinate, `0.5 * (v * is + is - 1)`:
 * the bracket is float (int operands converted to float), the product with 0.5 is double,
 * the store rounds back to float. */
static inline float ndc_to_pixel(float v, int is)
{
    float t = v * (float)is;
    t = t + (float)is;
    t = t - 1.0f;
    return (float)(0.5 * (double)t);
}

/* rasterize.py:147-155,261-269 -- inverse of [[x0,x1,x2],[y0,y1,y2],[1,1,1]] in pixel
 * coordinates, rows = vertices, columns = (a,b,c) with w_k = a_k*xi + b_k*yi + c_k. */
static inline void face_inverse(const float p[3][2], float inv[9])
{
    inv[0] = p[1][1] - p[2][1];
    inv[1] = p[2][0] - p[1][0];
    {
        const float m0 = p[1][0] * p[2][1];
        const float m1 = p[2][0] * p[1][1];
        inv[2] = m0 - m1;
    }
    inv[3] = p[2][1] - p[0][1];
    inv[4] = p[0][0] - p[2][0];
    {
        const float m0 = p[2][0] * p[0][1];
        const float m1 = p[0][0] * p[2][1];
        inv[5] = m0 - m1;
    }
    inv[6] = p[0][1] - p[1][1];
    inv[7] = p[1][0] - p[0][0];
    {
        const float m0 = p[0][0] * p[1][1];
        const float m1 = p[1][0] * p[0][1];
        inv[8] = m0 - m1;
    }
    float den;
    {
        const float t0 = p[2][0] * (p[0][1] - p[1][1]);
        const float t1 = p[0][0] * (p[1][1] - p[2][1]);
        const float t2 = p[1][0] * (p[2][1] - p[0][1]);
        den = t0 + t1;
        den = den + t2;
    }
    for (int k = 0; k < 9; k++) inv[k] = inv[k] / den;
}

/* rasterize.py:186-196,316-328 -- barycentric weights at integer pixel (xi, yi), clamp to
 * [0,1] through the double overloads of min/max (fmin/fmax: NaN loses), renormalise. */
static inline void bary_weights(const float inv[9], int xi, int yi, float w[3])
{
    const float fx = (float)xi, fy = (float)yi;
    float sum = 0.0f;
    for (int k = 0; k < 3; k++) {
        float t = inv[3 * k + 0] * fx;
        const float u = inv[3 * k + 1] * fy;
        t = t + u;
        t = t + inv[3 * k + 2];
        w[k] = (float)fmin(fmax((double)t, 0.), 1.);
        sum = sum + w[k];
    }
    for (int k = 0; k < 3; k++) w[k] = w[k] / sum;
}

/* rasterize.py:199,331 -- zp = 1. / (w0/z0 + w1/z1 + w2/z2); sum in float, reciprocal in double */
static inline float persp_depth(const float w[3], float z0, float z1, float z2)
{
    float s = w[0] / z0;
    s = s + w[1] / z1;
    s = s + w[2] / z2;
    return (float)(1. / (double)s);
}

/* --------------------------------------------------------------------------------------------
 * K2: per-face setup of the "safe" rasterizer -- rasterize.py:239-277.
 * faces [bs*nf,9]; faces_inv [bs*nf,9] must be zero-initialised by the caller (xp.zeros_like).
 */
ORC_API void orc_face_setup_safe(const float *faces, int n_faces_total, int is, float *faces_inv)
{
    for (int i = 0; i < n_faces_total; i++) {
        const float *f = faces + (size_t)i * 9;
        if (is_backface(f)) continue;
        float p[3][2];
        for (int num = 0; num < 3; num++)
            for (int dim = 0; dim < 2; dim++) p[num][dim] = ndc_to_pixel(f[3 * num + dim], is);
        float inv[9];
        face_inverse(p, inv);
        memcpy(faces_inv + (size_t)i * 9, inv, sizeof(inv));
    }
}

/* --------------------------------------------------------------------------------------------
 * K3: per-pixel z-buffered search over every face -- rasterize.py:280-360.
 * Maps must be pre-initialised by the caller exactly like forward_gpu (rasterize.py:475-493):
 * face_index_map = -1, weight_map = 0, depth_map = far, face_inv_map = 0.
 * near/far are doubles because the reference pastes the Python value as a C literal.
 */
ORC_API void orc_raster_safe(const float *faces, const float *faces_inv, int bs, int nf, int is,
                             double near, double far, int return_depth, int32_t *face_index_map,
                             float *weight_map, float *depth_map, float *face_inv_map)
{
    const long npx = (long)bs * is * is;
#pragma omp parallel for schedule(dynamic, 256)
    for (long i = 0; i < npx; i++) {
        const int bn = (int)(i / ((long)is * is));
        const int pn = (int)(i % ((long)is * is));
        const int yi = pn / is;
        const int xi = pn % is;
        /* rasterize.py:292-293: (2. * yi + 1 - is) / is in double, stored as float */
        const float yp = (float)((2. * yi + 1 - is) / is);
        const float xp = (float)((2. * xi + 1 - is) / is);

        const float *fbase = faces + (size_t)bn * nf * 9;
        const float *ibase = faces_inv + (size_t)bn * nf * 9;
        float depth_min = (float)far;
        int face_index_min = -1;
        float weight_min[3] = {0, 0, 0};
        float inv_min[9] = {0};
        for (int fn = 0; fn < nf; fn++) {
            const float *f = fbase + (size_t)fn * 9;
            const float *inv = ibase + (size_t)fn * 9;
            if (is_backface(f)) continue;
            /* rasterize.py:311-313 -- three edge tests in NDC, float */
            {
                const float l0 = (yp - f[1]) * (f[3] - f[0]);
                const float r0 = (xp - f[0]) * (f[4] - f[1]);
                if (l0 < r0) continue;
                const float l1 = (yp - f[4]) * (f[6] - f[3]);
                const float r1 = (xp - f[3]) * (f[7] - f[4]);
                if (l1 < r1) continue;
                const float l2 = (yp - f[7]) * (f[0] - f[6]);
                const float r2 = (xp - f[6]) * (f[1] - f[7]);
                if (l2 < r2) continue;
            }
            float w[3];
            bary_weights(inv, xi, yi, w);
            const float zp = persp_depth(w, f[2], f[5], f[8]);
            /* rasterize.py:332 -- comparisons against pasted literals happen in double */
            if ((double)zp <= near || far <= (double)zp) continue;
            if (zp < depth_min) {
                depth_min = zp;
                face_index_min = fn;
                for (int k = 0; k < 3; k++) weight_min[k] = w[k];
                if (return_depth) memcpy(inv_min, inv, sizeof(inv_min));
            }
        }
        if (0 <= face_index_min) {
            depth_map[i] = depth_min;
            face_index_map[i] = face_index_min;
            for (int k = 0; k < 3; k++) weight_map[3 * i + k] = weight_min[k];
            if (return_depth) memcpy(face_inv_map + 9 * i, inv_min, sizeof(inv_min));
        }
    }
}

/* --------------------------------------------------------------------------------------------
 * K1: the "unsafe" per-face scanline rasterizer -- rasterize.py:105-236, executed serially in
 * face order (the CUDA original resolves z-ties by scheduling order; serial order makes the
 * lowest face index win, which is also what K3 does).  Coverage rule differs from K3
 * (pixel-space scanline vs NDC edge functions); kept for cross-checking interior pixels.
 */
ORC_API void orc_raster_unsafe(const float *faces, int bs, int nf, int is, double near, double far,
                               int return_depth, int32_t *face_index_map, float *weight_map,
                               float *depth_map, float *face_inv_map)
{
    for (long i = 0; i < (long)bs * nf; i++) {
        const int bn = (int)(i / nf);
        const int fn = (int)(i % nf);
        const float *f = faces + (size_t)i * 9;
        if (is_backface(f)) continue;

        /* rasterize.py:123-131 -- order vertices by x */
        int pi[3] = {0, 0, 0};
        if (f[0] < f[3]) {
            pi[0] = (f[6] < f[0]) ? 2 : 0;
            pi[2] = (f[3] < f[6]) ? 2 : 1;
        } else {
            pi[0] = (f[6] < f[3]) ? 2 : 1;
            pi[2] = (f[0] < f[6]) ? 2 : 0;
        }
        for (int k = 0; k < 3; k++)
            if (pi[0] != k && pi[2] != k) pi[1] = k;

        float p[3][3];
        float p2[3][2];
        for (int num = 0; num < 3; num++) {
            p[num][0] = ndc_to_pixel(f[3 * pi[num] + 0], is);
            p[num][1] = ndc_to_pixel(f[3 * pi[num] + 1], is);
            p[num][2] = f[3 * pi[num] + 2];
            p2[num][0] = p[num][0];
            p2[num][1] = p[num][1];
        }
        if (p[0][0] == p[2][0]) continue; /* rasterize.py:144 */

        float inv[9];
        face_inverse(p2, inv);

        /* rasterize.py:158-159 */
        const int xi_min = cuda_f2i(fmax((double)ceilf(p[0][0]), 0.));
        const int xi_max = cuda_f2i(fmin((double)p[2][0], is - 1.));
        for (int xi = xi_min; xi <= xi_max; xi++) {
            float yi1, yi2;
            if ((float)xi <= p[1][0]) {
                if (p[1][0] - p[0][0] != 0) {
                    float s = (p[1][1] - p[0][1]) / (p[1][0] - p[0][0]);
                    s = s * ((float)xi - p[0][0]);
                    yi1 = s + p[0][1];
                } else {
                    yi1 = p[1][1];
                }
            } else {
                if (p[2][0] - p[1][0] != 0) {
                    float s = (p[2][1] - p[1][1]) / (p[2][0] - p[1][0]);
                    s = s * ((float)xi - p[1][0]);
                    yi1 = s + p[1][1];
                } else {
                    yi1 = p[1][1];
                }
            }
            {
                float s = (p[2][1] - p[0][1]) / (p[2][0] - p[0][0]);
                s = s * ((float)xi - p[0][0]);
                yi2 = s + p[0][1];
            }
            /* rasterize.py:179-180 */
            const int yi_min = cuda_f2i(fmax(0., (double)ceilf(fminf(yi1, yi2))));
            const int yi_max = cuda_f2i(fmin((double)fmaxf(yi1, yi2), is - 1.));
            for (int yi = yi_min; yi <= yi_max; yi++) {
                const long index = (long)bn * is * is + (long)yi * is + xi;
                float w[3];
                bary_weights(inv, xi, yi, w);
                const float zp = persp_depth(w, p[0][2], p[1][2], p[2][2]);
                if ((double)zp <= near || far <= (double)zp) continue;
                if (zp < depth_map[index]) {
                    depth_map[index] = zp;
                    face_index_map[index] = fn;
                    for (int k = 0; k < 3; k++) weight_map[3 * index + pi[k]] = w[k];
                    if (return_depth)
                        for (int k = 0; k < 3; k++)
                            for (int l = 0; l < 3; l++)
                                face_inv_map[9 * index + 3 * pi[l] + k] = inv[3 * l + k];
                }
            }
        }
    }
}

/* --------------------------------------------------------------------------------------------
 * K4: texture sampling -- rasterize.py:371-435.  Note rasterize.py:390 indexes `faces` without
 * the batch offset; restated as written (harmless for bs == 1, the only case 3D-SDN uses).
 */
ORC_API void orc_texture_sampling(const float *faces, const float *textures,
                                  const int32_t *face_index_map, const float *weight_map,
                                  const float *depth_map, int bs, int nf, int is, int ts, double eps,
                                  float *rgb_map, int32_t *sampling_index_map,
                                  float *sampling_weight_map)
{
    const long npx = (long)bs * is * is;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < npx; i++) {
        const int face_index = face_index_map[i];
        if (face_index < 0) continue;
        const int bn = (int)(i / ((long)is * is));
        const float *face = faces + (size_t)face_index * 9;
        const float *texture = textures + ((size_t)bn * nf + face_index) * ts * ts * ts * 3;
        const float *weight = weight_map + i * 3;
        const float depth = depth_map[i];

        float tif[3];
        for (int k = 0; k < 3; k++) {
            /* weight * (ts - 1 - eps) * (depth / z): double product of a float quotient */
            const float q = depth / face[3 * k + 2];
            const double c = (double)(ts - 1) - eps;
            tif[k] = (float)(((double)weight[k] * c) * (double)q);
        }
        float new_pixel[3] = {0, 0, 0};
        for (int pn = 0; pn < 8; pn++) {
            float w = 1;
            int tii[3];
            for (int k = 0; k < 3; k++) {
                const int base = cuda_f2i((double)tif[k]);
                const float frac = tif[k] - (float)base;
                if ((pn >> k) % 2 == 0) {
                    w = w * (1.0f - frac);
                    tii[k] = base;
                } else {
                    w = w * frac;
                    tii[k] = base + 1;
                }
            }
            const int isc = tii[0] * ts * ts + tii[1] * ts + tii[2];
            for (int k = 0; k < 3; k++) {
                const float t = w * texture[isc * 3 + k];
                new_pixel[k] = new_pixel[k] + t;
            }
            if (sampling_index_map) sampling_index_map[i * 8 + pn] = isc;
            if (sampling_weight_map) sampling_weight_map[i * 8 + pn] = w;
        }
        for (int k = 0; k < 3; k++) rgb_map[i * 3 + k] = new_pixel[k];
    }
}

/* rasterize.py:446 alpha map, :457-462 background blend (bg is [3] or [bs,3]) */
ORC_API void orc_alpha_background(const int32_t *face_index_map, int bs, int is, float *alpha_map,
                                  float *rgb_map, const float *bg, int bg_per_batch)
{
    const long npx = (long)bs * is * is;
    for (long i = 0; i < npx; i++) {
        const float mask = (0 <= face_index_map[i]) ? 1.0f : 0.0f;
        if (alpha_map && mask != 0.0f) alpha_map[i] = 1.0f;
        if (rgb_map) {
            const int bn = (int)(i / ((long)is * is));
            const float *c = bg + (bg_per_batch ? 3 * bn : 0);
            for (int k = 0; k < 3; k++) {
                const float a = rgb_map[3 * i + k] * mask;
                const float b = (1.0f - mask) * c[k];
                rgb_map[3 * i + k] = a + b;
            }
        }
    }
}

/* --------------------------------------------------------------------------------------------
 * K5: hand-crafted silhouette / colour gradient wrt x,y of each face's vertices --
 * rasterize.py:523-745.  grad_faces must be zero-initialised (rasterize.py:848); front faces
 * get a plain store (z components 0), back faces keep zeros.
 */
static inline float edge_dist(float pa, float pb, float denom_term, int d1, float d1_cross, int is,
                              double eps)
{
    /* (p1.d0 - p0.d0) / denom_term * (d1 - d1_cross) * 2. / is ; then +/- eps, in double */
    float t = (pb - pa) / denom_term;
    t = t * ((float)d1 - d1_cross);
    float dist = (float)(((double)t * 2.) / (double)is);
    dist = (0 < dist) ? (float)((double)dist + eps) : (float)((double)dist - eps);
    return dist;
}

ORC_API void orc_backward_pixel_map(const float *faces, const int32_t *face_index_map,
                                    const float *rgb_map, const float *alpha_map,
                                    const float *grad_rgb_map, const float *grad_alpha_map, int bs,
                                    int nf, int is, double eps, int return_rgb, int return_alpha,
                                    float *grad_faces)
{
    const long total = (long)bs * nf;
#pragma omp parallel for schedule(dynamic, 64)
    for (long i = 0; i < total; i++) {
        const int bn = (int)(i / nf);
        const int fn = (int)(i % nf);
        const float *face = faces + (size_t)i * 9;
        float grad_face[9] = {0};
        if (is_backface(face)) continue;

        for (int edge_num = 0; edge_num < 3; edge_num++) {
            int pi[3];
            float pp[3][2];
            for (int num = 0; num < 3; num++) pi[num] = (edge_num + num) % 3;
            for (int num = 0; num < 3; num++)
                for (int dim = 0; dim < 2; dim++) pp[num][dim] = ndc_to_pixel(face[3 * pi[num] + dim], is);

            for (int axis = 0; axis < 2; axis++) {
                float p[3][2];
                for (int num = 0; num < 3; num++)
                    for (int dim = 0; dim < 2; dim++) p[num][dim] = pp[num][(dim + axis) % 2];

                int direction;
                if (axis == 0)
                    direction = (p[0][0] < p[1][0]) ? -1 : 1;
                else
                    direction = (p[0][0] < p[1][0]) ? 1 : -1;

                /* rasterize.py:565-566 */
                const int d0_from = cuda_f2i(fmax((double)ceilf(fminf(p[0][0], p[1][0])), 0.));
                const int d0_to = cuda_f2i(fmin((double)fmaxf(p[0][0], p[1][0]), is - 1.));
                for (int d0 = d0_from; d0 <= d0_to; d0++) {
                    float d1_cross;
                    {
                        float s = (p[1][1] - p[0][1]) / (p[1][0] - p[0][0]);
                        s = s * ((float)d0 - p[0][0]);
                        d1_cross = s + p[0][1];
                    }
                    int d1_in;
                    if (0 < direction)
                        d1_in = cuda_f2i((double)floorf(d1_cross));
                    else
                        d1_in = cuda_f2i((double)ceilf(d1_cross));
                    const int d1_out = d1_in + direction;
                    if (d1_in < 0 || is <= d1_in) continue;
                    if (d1_out < 0 || is <= d1_out) continue;

                    long map_index_in, map_index_out;
                    const long base = (long)bn * is * is;
                    if (axis == 0) {
                        map_index_in = base + (long)d1_in * is + d0;
                        map_index_out = base + (long)d1_out * is + d0;
                    } else {
                        map_index_in = base + (long)d0 * is + d1_in;
                        map_index_out = base + (long)d0 * is + d1_out;
                    }
                    float alpha_in = 0, alpha_out = 0;
                    const float *rgb_in = 0, *rgb_out = 0;
                    if (return_alpha) {
                        alpha_in = alpha_map[map_index_in];
                        alpha_out = alpha_map[map_index_out];
                    }
                    if (return_rgb) {
                        rgb_in = rgb_map + map_index_in * 3;
                        rgb_out = rgb_map + map_index_out * 3;
                    }
                    const long map_offset = (axis == 0) ? is : 1;

                    /* "out" pass, rasterize.py:600-656 */
                    if (face_index_map[map_index_in] == fn) {
                        const int d1_limit = (0 < direction) ? is - 1 : 0;
                        int d1_from = d1_out < d1_limit ? d1_out : d1_limit;
                        if (d1_from < 0) d1_from = 0;
                        int d1_to = d1_out > d1_limit ? d1_out : d1_limit;
                        if (d1_to > is - 1) d1_to = is - 1;
                        long q = (axis == 0) ? base + (long)d1_from * is + d0 : base + (long)d0 * is + d1_from;
                        for (int d1 = d1_from; d1 <= d1_to; d1++, q += map_offset) {
                            float diff_grad = 0;
                            if (return_alpha) {
                                const float t = (alpha_map[q] - alpha_in) * grad_alpha_map[q];
                                diff_grad = diff_grad + t;
                            }
                            if (return_rgb)
                                for (int k = 0; k < 3; k++) {
                                    const float t = (rgb_map[q * 3 + k] - rgb_in[k]) * grad_rgb_map[q * 3 + k];
                                    diff_grad = diff_grad + t;
                                }
                            if (diff_grad <= 0) continue;
                            if (p[1][0] != (float)d0) {
                                const float dist = edge_dist(p[0][0], p[1][0], p[1][0] - (float)d0, d1, d1_cross, is, eps);
                                grad_face[pi[0] * 3 + (1 - axis)] -= diff_grad / dist;
                            }
                            if (p[0][0] != (float)d0) {
                                const float dist = edge_dist(p[0][0], p[1][0], (float)d0 - p[0][0], d1, d1_cross, is, eps);
                                grad_face[pi[1] * 3 + (1 - axis)] -= diff_grad / dist;
                            }
                        }
                    }

                    /* "in" pass, rasterize.py:658-727 */
                    {
                        float d0_cross2;
                        if (((float)d0 - p[0][0]) * ((float)d0 - p[2][0]) < 0) {
                            float s = (p[2][1] - p[0][1]) / (p[2][0] - p[0][0]);
                            s = s * ((float)d0 - p[0][0]);
                            d0_cross2 = s + p[0][1];
                        } else {
                            float s = (p[1][1] - p[2][1]) / (p[1][0] - p[2][0]);
                            s = s * ((float)d0 - p[2][0]);
                            d0_cross2 = s + p[2][1];
                        }
                        int d1_limit;
                        if (0 < direction)
                            d1_limit = cuda_f2i((double)ceilf(d0_cross2));
                        else
                            d1_limit = cuda_f2i((double)floorf(d0_cross2));
                        int d1_from = d1_in < d1_limit ? d1_in : d1_limit;
                        if (d1_from < 0) d1_from = 0;
                        int d1_to = d1_in > d1_limit ? d1_in : d1_limit;
                        if (d1_to > is - 1) d1_to = is - 1;
                        long q = (axis == 0) ? base + (long)d1_from * is + d0 : base + (long)d0 * is + d1_from;
                        for (int d1 = d1_from; d1 <= d1_to; d1++, q += map_offset) {
                            if (face_index_map[q] != fn) continue;
                            float diff_grad = 0;
                            if (return_alpha) {
                                const float t = (alpha_map[q] - alpha_out) * grad_alpha_map[q];
                                diff_grad = diff_grad + t;
                            }
                            if (return_rgb)
                                for (int k = 0; k < 3; k++) {
                                    const float t = (rgb_map[q * 3 + k] - rgb_out[k]) * grad_rgb_map[q * 3 + k];
                                    diff_grad = diff_grad + t;
                                }
                            if (diff_grad <= 0) continue;
                            if (p[1][0] != (float)d0) {
                                const float dist = edge_dist(p[0][0], p[1][0], p[1][0] - (float)d0, d1, d1_cross, is, eps);
                                grad_face[pi[0] * 3 + (1 - axis)] -= diff_grad / dist;
                            }
                            if (p[0][0] != (float)d0) {
                                const float dist = edge_dist(p[0][0], p[1][0], (float)d0 - p[0][0], d1, d1_cross, is, eps);
                                grad_face[pi[1] * 3 + (1 - axis)] -= diff_grad / dist;
                            }
                        }
                    }
                }
            }
        }
        for (int k = 0; k < 9; k++) grad_faces[i * 9 + k] = grad_face[k];
    }
}

/* --------------------------------------------------------------------------------------------
 * K6: texture gradient scatter -- rasterize.py:756-789 (serial => deterministic sum order).
 */
ORC_API void orc_backward_textures(const int32_t *face_index_map, const float *sampling_weight_map,
                                   const int32_t *sampling_index_map, const float *grad_rgb_map,
                                   int bs, int nf, int is, int ts, float *grad_textures)
{
    const long npx = (long)bs * is * is;
    for (long i = 0; i < npx; i++) {
        const int face_index = face_index_map[i];
        if (face_index < 0) continue;
        const int bn = (int)(i / ((long)is * is));
        float *gt = grad_textures + ((size_t)bn * nf + face_index) * ts * ts * ts * 3;
        for (int pn = 0; pn < 8; pn++) {
            const float w = sampling_weight_map[i * 8 + pn];
            const int isc = sampling_index_map[i * 8 + pn];
            for (int k = 0; k < 3; k++) {
                const float t = w * grad_rgb_map[i * 3 + k];
                gt[isc * 3 + k] = gt[isc * 3 + k] + t;
            }
        }
    }
}

/* --------------------------------------------------------------------------------------------
 * K7: depth gradient wrt x,y,z of the winning face -- rasterize.py:800-844.  Accumulates INTO
 * grad_faces (the reference atomically adds after K5's store).
 */
ORC_API void orc_backward_depth(const float *faces, const float *depth_map,
                                const int32_t *face_index_map, const float *face_inv_map,
                                const float *weight_map, const float *grad_depth_map, int bs, int nf,
                                int is, float *grad_faces)
{
    const long npx = (long)bs * is * is;
    for (long i = 0; i < npx; i++) {
        const int fn = face_index_map[i];
        if (fn < 0) continue;
        const int bn = (int)(i / ((long)is * is));
        const float *face = faces + ((size_t)bn * nf + fn) * 9;
        const float depth = depth_map[i];
        const float depth2 = depth * depth;
        const float *inv = face_inv_map + i * 9;
        const float *weight = weight_map + i * 3;
        const float g = grad_depth_map[i];
        float *gf = grad_faces + ((size_t)bn * nf + fn) * 9;

        for (int k = 0; k < 3; k++) {
            const float zk = face[3 * k + 2];
            float t = g * weight[k];
            t = t * depth2;
            t = t / (zk * zk);
            gf[3 * k + 2] = gf[3 * k + 2] + t;
        }
        float tmp[3] = {0, 0, 0};
        for (int k = 0; k < 3; k++)
            for (int l = 0; l < 3; l++) {
                const float t = -inv[3 * l + k] / face[3 * l + 2];
                tmp[k] = tmp[k] + t;
            }
        for (int k = 0; k < 3; k++)
            for (int l = 0; l < 2; l++) {
                float t = -g * tmp[l];
                t = t * weight[k];
                t = t * depth2;
                t = t * (float)is;
                t = t / 2.0f;
                gf[3 * k + l] = gf[3 * k + l] + t;
            }
    }
}

/* K8: cross product rows -- cross.py:25-38 */
ORC_API void orc_cross(const float *a, const float *b, long rows, float *c)
{
    for (long j = 0; j < rows; j++) {
        const float *ap = a + 3 * j, *bp = b + 3 * j;
        float *cp = c + 3 * j;
        cp[0] = ap[1] * bp[2] - ap[2] * bp[1];
        cp[1] = ap[2] * bp[0] - ap[0] * bp[2];
        cp[2] = ap[0] * bp[1] - ap[1] * bp[0];
    }
}

ORC_API int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

ORC_API void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}
