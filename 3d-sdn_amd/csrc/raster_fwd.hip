// Forward rasterizer for gfx950: per-face setup + LDS-tiled, deterministic z-resolve.
//
// Replaces Rasterize.forward_gpu and the post-processing of rasterize_rgbad
// (/root/reference/geometric/neural_renderer/rasterize.py:464-510, 942-972).  The reference's
// default "unsafe" kernel (rasterize.py:105-236) scatters per face under a per-pixel spinlock and its
// "safe" one (rasterize.py:238-360) tests every pixel against every face; this file keeps the *results*
// of the safe kernel (same coverage predicate, same barycentric / depth arithmetic, lowest face index
// wins exact depth ties) with a CDNA4 execution plan:
//
//   k_face_setup   one thread per face: back-face test, inverse matrix (rasterize.py:246-272),
//                  conservative pixel/tile bounding box, per-tile face counts (LDS-privatised histogram).
//   k_tile_offsets one workgroup per batch element: exclusive scan of the tile counts.
//   k_tile_fill    one thread per face: append the face index to the list of every tile it touches (a workgroup counts its
//                  entries per tile in LDS and reserves each tile's run with one global atomic).
//   k_raster_tiles one 256-thread workgroup per 32x32-pixel tile (framebuffer bin in LDS: 1024 x u64 =
//                  8 KiB).  The tile reads its own face list (coalesced 4-byte indices; if the lists of a
//                  batch element overflow their budget the tile streams the 4-byte tile-box of every face
//                  instead) in four runs, one per wave.  A wave rasterises 64 faces at a time: small faces (a few
//                  pixels: nearly all of an 85k-face mesh) stay with their lane(s), which only run the edge tests and
//                  queue the covered pixels in LDS; every 64 queued hits are shaded by all 64 lanes (barycentrics,
//                  perspective depth) and resolved with ds_min_u64 on the packed key ord(depth) << 32 | face_index;
//                  large faces are broadcast (v_readlane) and shared by the wave.  r02: 658 -> 399 us per 16-object frame.
//                  The epilogue recomputes the winner's barycentrics, samples colours
//                  (rasterize.py:398-423), blends the background, flips vertically and 2x2-averages
//                  (rasterize.py:951-966) straight from LDS to the output maps.
//
// Exactness: a (face, pixel) pair is evaluated with the same float operations wherever it is
// evaluated, so binning cannot change the result as long as no covering pair is skipped.  The
// bounding box is therefore dilated by a proven bound on where rounding can make the NDC edge
// tests of rasterize.py:311-313 pass (see face_margin_px); degenerate faces fall back to the whole
// screen band around the line through their longest edge, which is where the reference's unbounded edge
// tests can still accept pixels.
#include <mutex>
#include <type_traits>
#include <vector>

#include "camera_math.h"
#include "raster_math.h"
#include "sdn_common.h"

namespace sdn {

constexpr int TS = 32;      // tile side in internal pixels (16 with one wave per tile measured the same: 527 vs 506 us)
constexpr int NTHR = (TS / 2) * (TS / 2);  // threads per workgroup: one per 2x2 pixel quad of the tile
constexpr int NWAVE = NTHR / 64;
static_assert(NTHR % 64 == 0 && NTHR >= 2 * TS && TS <= 32, "tile / workgroup geometry (hit queue packs px, py in 5 bits)");
constexpr int QCAP = 512;   // queued faces per flush (<= 255 carried + 256 new)
constexpr int FREC = 13;        // floats per face record of a batch (odd: consecutive records start in different LDS banks)
#ifndef SDN_LAB_SMALL_AREA
#define SDN_LAB_SMALL_AREA 32   // (sweep r03, us per frame car_like / cad_like: 12: 208/303, 24: 191/297, 32: 185/293, 48: 193/306, 64: 200/318, 96: 214/323)
#endif
constexpr int SMALL_AREA = SDN_LAB_SMALL_AREA;  // clipped candidate boxes up to this many pixels are rasterised by ONE lane
#ifndef SDN_LAB_SPAN_AREA
#define SDN_LAB_SPAN_AREA 256   // (sweep r04, us per launch car_like / cad_like: 0: 219/312, 128: 203/291, 256: 200/290, 512: 200/289, off: 200/299;
                                // r06, with the depth cull, car_like / cad_like / the six templates / 3776e4d1: 64: 196/262/296/489, 128: 191/261/294/470,
                                // 256: 189/253/287/446, 512: 189/255/326/512, 1024: 187/260/487/-)
#endif
constexpr int SPAN_AREA = SDN_LAB_SPAN_AREA;    // wave-shared boxes above this many pixels are walked by row spans
#ifndef SDN_LAB_HIZ_MIN_LIST
#define SDN_LAB_HIZ_MIN_LIST 257   // = every tile on the long-list path (> 64 * NWAVE entries); us per launch cad_like / real templates / car_like:
                                   // off 289 / 490 / 196, always 291 / 325 / 214, >= 2048: 290 / 490 / 196, 1024: 288 / 343 / 195, 512: 275 / 317 / 195, 256: 266 / 313 / 196
#endif
constexpr int HIZ_MIN_LIST = SDN_LAB_HIZ_MIN_LIST;
#ifndef SDN_K1_HIZ
#define SDN_K1_HIZ 1   // (lab builds: 0 = the K1 tile kernel without the cull)
#endif   // adaptive hierarchical-z (k_raster_tiles<., 2>): tiles with at least this many list entries
constexpr uint32_t TB_CULLED = 0x000000FFu;  // tx0 = 255 > tx1 = 0: matches no tile

struct FwdParams {
    const float* faces;
    const float* textures;
    const float* bg;
    float* face_inv;
    int32_t* face_index_map;
    float* weight_map;
    float* depth_map;
    float* rgb_map;
    float* rgb_out;
    float* alpha_out;
    float* depth_out;
    const uint32_t* tilebox;
    const uint4* pixbox;        // [bs, nf] x0 | x1 << 16, y0 | y1 << 16, bits of the face's margin (face_margin_px), -
    const uint32_t* tile_off;   // [bs, ntiles + 1]
    const uint32_t* tile_list;  // [bs, list_cap]
    const uint32_t* overflow;   // [bs]
    const uint32_t* tile_order; // [bs * ntiles] launch rank -> b * ntiles + tile, heaviest lists first
    const uint32_t* thin_count; // [bs]
    const float4* thin_list;    // [bs, 2, nf]: {nx, ny, a . n, band} of every thin face, then {ax, ay, face index bits, -}
    unsigned long long* counters;  // [16] candidate pixel tests, tests passed, depth keys submitted, phase clocks (COUNT builds only)
    uint32_t list_cap;
    double eps;
    int ts, bs, nf, S, ntx, flags, bg_per_batch;
    int hiz;   // SDN_RASTER_HIZ (0 off, 1 every tile, 2 tiles with long lists): the template argument of k_raster_tiles; the K1 kernel reads it
    float near_le, far_f;
};

// Bound (in pixels) on how far outside its exact footprint a face can still pass the float edge
// tests.  Each test compares fl(fl(yp-ya)*fl(xb-xa)) with fl(fl(xp-xa)*fl(yb-ya)); the absolute error of
// the difference is <= 6u*D*L (u = 2^-24, D = max |pixel - vertex| component <= 1 + cmax, L = edge
// component), i.e. a slack of delta = 6u(1+cmax) in distance to the edge line.  Relaxing all three
// half-planes by delta scales the triangle about its incentre by (r+delta)/r, r = 2A/P, which moves a
// vertex by at most delta*P*Lmax/(2A).  A safety factor of ~2.7 is folded into 2^-20.
constexpr float THIN_MARGIN = 4.0f;  // faces whose margin exceeds this take the band path instead of a bounding box

__device__ __forceinline__ float face_margin_px(const float f[9], int S)
{
    const float ex0 = f[3] - f[0], ey0 = f[4] - f[1];
    const float ex1 = f[6] - f[0], ey1 = f[7] - f[1];
    const float ex2 = f[6] - f[3], ey2 = f[7] - f[4];
    const float l0 = sqrtf(ex0 * ex0 + ey0 * ey0);
    const float l1 = sqrtf(ex1 * ex1 + ey1 * ey1);
    const float l2 = sqrtf(ex2 * ex2 + ey2 * ey2);
    const float lmax = fmaxf(l0, fmaxf(l1, l2));
    const float perim = l0 + l1 + l2;
    float cmax = fmaxf(fmaxf(fabsf(f[0]), fabsf(f[1])), fmaxf(fabsf(f[3]), fabsf(f[4])));
    cmax = fmaxf(cmax, fmaxf(fabsf(f[6]), fabsf(f[7])));
    const float area2 = fabsf(ex0 * ey1 - ex1 * ey0) - 9.5367432e-7f * lmax * lmax;
    if (!(area2 > 0.0f)) return -1.0f;  // degenerate (or NaN): whole screen
    const float delta = 9.5367432e-7f * (1.0f + cmax);
    // base term: rounding of the NDC -> pixel mapping used for the box (a few ulp of S) + the slack itself
    const float base = 0.004f * fmaxf(1.0f, (float)S * (1.0f / 1024.0f)) + 0.5f * (float)S * delta;
    const float m = base + 0.5f * (float)S * (delta * perim * lmax / area2);
    if (!(m < (float)S)) return -1.0f;
    return m;
}

constexpr int HIST_MAX = 4096;  // tiles per image that fit the LDS histogram (S <= 2048)

// K1: the SDN_K1_COVERAGE build (a template, not a run-time branch: the extra live values cost the default build 8 of 27 us)
// GATHER (r06): the face is not read from a finished [bs, nf, 3, 3] tensor but built here -- its three vertices gathered through
// the index list (the fill_back twin in reverse order), x-flipped, run through camera + perspective (camera_math.h: k_project's
// operations), written to G.faces_out for the tile kernel and the backward pass, and its pre-camera normal to G.normals_out
// (k_face_normals_gather's operations).  What sdn_render_maps_fwd used to issue as k_face_normals_gather + k_project +
// k_gather_faces + k_face_setup (61 us of a 16-object frame, the projected vertices and the face tensor written and read back)
// is this one launch; the values are the same floats, so everything downstream is bit-identical.
template <bool K1, bool GATHER>
__global__ __launch_bounds__(256) void k_face_setup(const float* __restrict__ faces, int nf, int S, int ntx,
                                                     float* __restrict__ face_inv, uint32_t* __restrict__ tilebox,
                                                     uint4* __restrict__ pixbox, uint32_t* __restrict__ tile_count,
                                                     uint32_t* __restrict__ thin_count, float4* __restrict__ thin_list,
                                                     const FaceSource G)
{
    // grid = (ceil(nf / 256), bs): a workgroup never straddles two batch elements, so its histogram is private
    __shared__ uint32_t hist[HIST_MAX];
    const int ntiles = ntx * ntx;
    const bool use_lds = ntiles <= HIST_MAX;
    if (use_lds)
        for (int t = threadIdx.x; t < ntiles; t += 256) hist[t] = 0u;
    __syncthreads();
    const int b = blockIdx.y;
    const int t_local = blockIdx.x * 256 + threadIdx.x;
    uint32_t* gcount = tile_count + (size_t)b * ntiles;
    // GATHER with fill_back (r06): a thread builds face f AND its twin nf0 + f (the same three vertices in reverse order): gathered and
    // projected once (the launch has nf0 threads per object, see rasterize_fwd_core); exactly one of the two is front-facing.
    const bool pairs = GATHER && G.fill_back != 0;
    const int nthr = pairs ? G.nf0 : nf;
    float v[3][3], pf[9];
    if constexpr (GATHER) {
        if (t_local < nthr) {
            const int32_t* idx = G.faces_idx + (size_t)b * G.fstride + (size_t)t_local * 3;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float* p = G.verts + ((size_t)b * G.nv + idx[k]) * 3;
                v[k][0] = G.flip_x ? p[0] * -1.0f : p[0];
                v[k][1] = p[1];
                v[k][2] = p[2];
            }
            Basis B;
            if (G.mode != 0) B = camera_basis(G.mode, G.eye, G.dir, G.up, b);
            const float wv = G.width ? G.width[b] : 1.0f;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                float o[3];
                project_vertex(v[k], G.mode, B, G.width != nullptr, wv, o);
                pf[3 * k + 0] = o[0];
                pf[3 * k + 1] = o[1];
                pf[3 * k + 2] = o[2];
            }
        }
    }
    if (t_local < nthr)
    for (int rep = 0; rep < (pairs ? 2 : 1); rep++) {
    const int fn_local = t_local + (rep ? G.nf0 : 0);
    const long i = (long)b * nf + fn_local;
    float f[9];
    float nrm[3] = {0.f, 0.f, 0.f};   // GATHER: the pre-camera normal (the colour of the normal map)
    if constexpr (GATHER) {
        if (G.normals_out) {
            if (rep)
                face_normal(v[2], v[1], v[0], G.sx, nrm);
            else
                face_normal(v[0], v[1], v[2], G.sx, nrm);
        }
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int d = 0; d < 3; d++) f[3 * k + d] = rep ? pf[3 * (2 - k) + d] : pf[3 * k + d];
    } else {
#pragma unroll
        for (int k = 0; k < 9; k++) f[k] = faces[i * 9 + k];
    }
    float inv[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t tb = TB_CULLED;
    uint4 pb = make_uint4(0, 0, 0, 0);
    bool thin = false;   // recorded in the thin-face list: drawn through the band path although its tile box says culled
    if constexpr (K1) {
      if (!is_backface(f)) {
        // SDN_K1_COVERAGE (raster_math.h): the box is exactly the columns K1 walks and the vertices' rows (+- 1 for the
        // rounding of the edge interpolations); face_inv is computed on the x-SORTED vertices as K1 does and stored with its
        // rows back in the original vertex order -- what the reference keeps per pixel in face_inv_map (rasterize.py:206-210)
        bool finite = true;
#pragma unroll
        for (int k = 0; k < 9; k++) finite = finite && (f[k] - f[k] == 0.0f);
        const K1Face K = k1_setup(f, S);
        if (finite && !K.dead && K.xi_min <= K.xi_max) {
            float inv_s[9];
            face_inverse(K.px, K.py, inv_s);
#pragma unroll
            for (int l = 0; l < 3; l++)
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const float v = inv_s[3 * l + k];
                    if (K.pi[l] == 0) inv[k] = v;
                    if (K.pi[l] == 1) inv[3 + k] = v;
                    if (K.pi[l] == 2) inv[6 + k] = v;
                }
            const float ymin = fminf(K.py[0], fminf(K.py[1], K.py[2])), ymax = fmaxf(K.py[0], fmaxf(K.py[1], K.py[2]));
            int y0 = 0, y1 = S - 1;
            if (!(K.px[2] < 0.0f)) {   // (a face that ends in the pixel column left of the screen is EXTRAPOLATED to column 0)
                y0 = (int)fmaxf(floorf(ymin) - 1.0f, 0.0f);
                y1 = (int)fminf(ceilf(ymax) + 1.0f, (float)(S - 1));
            }
            if (y0 <= y1 && ymax + 1.0f >= 0.0f && ymin - 1.0f <= (float)(S - 1)) {
                const int x0 = K.xi_min, x1 = K.xi_max;
                tb = (uint32_t)(x0 / TS) | ((uint32_t)(x1 / TS) << 8) | ((uint32_t)(y0 / TS) << 16) | ((uint32_t)(y1 / TS) << 24);
                pb = make_uint4((uint32_t)x0 | ((uint32_t)x1 << 16), (uint32_t)y0 | ((uint32_t)y1 << 16), 0u, 0u);
                for (int ty = y0 / TS; ty <= y1 / TS; ty++)
                    for (int tx = x0 / TS; tx <= x1 / TS; tx++) {
                        if (use_lds)
                            atomicAdd(&hist[ty * ntx + tx], 1u);
                        else
                            atomicAdd(&gcount[ty * ntx + tx], 1u);
                    }
            }
        }
      }
    } else if (!is_backface(f)) {
        const float is_f = (float)S;
        float px[3], py[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            px[k] = ndc_to_pixel(f[3 * k + 0], is_f);
            py[k] = ndc_to_pixel(f[3 * k + 1], is_f);
        }
        face_inverse(px, py, inv);
        int x0 = 0, x1 = S - 1, y0 = 0, y1 = S - 1;
        const float m = face_margin_px(f, S);
        const float xmin = fminf(px[0], fminf(px[1], px[2])), xmax = fmaxf(px[0], fmaxf(px[1], px[2]));
        const float ymin = fminf(py[0], fminf(py[1], py[2])), ymax = fmaxf(py[0], fmaxf(py[1], py[2]));
        const bool finite = (xmin - xmin == 0.0f) && (xmax - xmax == 0.0f) && (ymin - ymin == 0.0f) &&
                            (ymax - ymax == 0.0f) && (px[0] == px[0]) && (px[1] == px[1]) && (px[2] == px[2]) &&
                            (py[0] == py[0]) && (py[1] == py[1]) && (py[2] == py[2]);
        bool visible = true;
        if (!finite) {
            // NaN / inf coordinates make every entry of face_inv NaN or zero: the barycentric weights clamp to 0, their
            // sum is 0 and the depth is NaN, which the reference never records (rasterize.py:335) -> nothing to draw
            visible = false;
        } else if (!(m > 0.0f) || m > THIN_MARGIN) {
            // Sliver / degenerate face: rounding can let pixels far beyond its extent pass the edge tests, but only
            // within a hair of the line through its longest edge.  Record that line; tiles test a band around it.
            visible = false;
            int ia = 0, ib = 1;
            float l2 = (px[1] - px[0]) * (px[1] - px[0]) + (py[1] - py[0]) * (py[1] - py[0]);
            const float l2b = (px[2] - px[0]) * (px[2] - px[0]) + (py[2] - py[0]) * (py[2] - py[0]);
            const float l2c = (px[2] - px[1]) * (px[2] - px[1]) + (py[2] - py[1]) * (py[2] - py[1]);
            if (l2b > l2) { l2 = l2b; ia = 0; ib = 2; }
            if (l2c > l2) { l2 = l2c; ia = 1; ib = 2; }
            if (l2 > 0.0f) {  // l2 == 0: the three vertices coincide, face_inv is all NaN, never recorded
                const float L = sqrtf(l2);
                const float nx = -(py[ib] - py[ia]) / L, ny = (px[ib] - px[ia]) / L;
                const int ic = 3 - ia - ib;
                const float h = fabsf((px[ic] - px[ia]) * nx + (py[ic] - py[ia]) * ny);
                const float pmax = fmaxf(fmaxf(fabsf(xmin), fabsf(xmax)), fmaxf(fabsf(ymin), fabsf(ymax)));
                // dilated triangle (see face_margin_px) lies within [-delta, h + 2 delta] of its longest edge's line
                float cmax = fmaxf(fmaxf(fabsf(f[0]), fabsf(f[1])), fmaxf(fabsf(f[3]), fabsf(f[4])));
                cmax = fmaxf(cmax, fmaxf(fabsf(f[6]), fabsf(f[7])));
                const float delta_px = 9.5367432e-7f * (1.0f + cmax) * 0.5f * (float)S;
                const float band = h + 0.05f + 2.5f * delta_px + 4e-6f * (pmax + (float)S);
                const uint32_t slot = atomicAdd(&thin_count[b], 1u);
                float4* e = thin_list + (size_t)b * nf * 2;      // [nf] cull records, then [nf] face records
                e[slot] = make_float4(nx, ny, px[ia] * nx + py[ia] * ny, band);
                e[nf + slot] = make_float4(px[ia], py[ia], __uint_as_float((uint32_t)fn_local), 0.f);
                thin = true;
            }
        } else {
            // pixel centres sit at integer pixel coordinates: candidates are the integers inside the dilated box
            const float fx0 = ceilf(xmin - m), fx1 = floorf(xmax + m);
            const float fy0 = ceilf(ymin - m), fy1 = floorf(ymax + m);
            if (fx1 < 0.0f || fy1 < 0.0f || fx0 > (float)(S - 1) || fy0 > (float)(S - 1) || fx0 > fx1 || fy0 > fy1) {
                visible = false;  // off screen, or no pixel centre inside: the face cannot win any pixel
            } else {
                x0 = (int)fmaxf(fx0, 0.0f);
                y0 = (int)fmaxf(fy0, 0.0f);
                x1 = (int)fminf(fx1, (float)(S - 1));
                y1 = (int)fminf(fy1, (float)(S - 1));
            }
        }
        if (visible) {
            tb = (uint32_t)(x0 / TS) | ((uint32_t)(x1 / TS) << 8) | ((uint32_t)(y0 / TS) << 16) |
                 ((uint32_t)(y1 / TS) << 24);
            pb = make_uint4((uint32_t)x0 | ((uint32_t)x1 << 16), (uint32_t)y0 | ((uint32_t)y1 << 16), __float_as_uint(m), 0u);
            for (int ty = y0 / TS; ty <= y1 / TS; ty++)
                for (int tx = x0 / TS; tx <= x1 / TS; tx++) {
                    if (use_lds)
                        atomicAdd(&hist[ty * ntx + tx], 1u);
                    else
                        atomicAdd(&gcount[ty * ntx + tx], 1u);
                }
        }
    }
    tilebox[i] = tb;
    // GATHER (sdn_render_maps_fwd's private state): a face that cannot win a pixel -- back-facing (one of every fill_back pair), off
    // screen, no pixel centre inside -- is never looked at again: the tile kernels take faces from the tile / thin lists, the
    // backward pass from the face-index map.  Its 100 bytes of face tensor, inverse matrix, colour and pixel box stay unwritten
    // (two thirds of the launch's 300 MB; the launch is bound by its stores).
    if (!GATHER || tb != TB_CULLED || thin) {
#pragma unroll
        for (int k = 0; k < 9; k++) face_inv[i * 9 + k] = inv[k];
        pixbox[i] = pb;
        if constexpr (GATHER) {
            float* fo = G.faces_out + i * 9;
#pragma unroll
            for (int k = 0; k < 9; k++) fo[k] = f[k];
            if (G.normals_out) {
                float* o = G.normals_out + i * 3;
                o[0] = nrm[0];
                o[1] = nrm[1];
                o[2] = nrm[2];
            }
        }
    }
    }
    __syncthreads();
    if (use_lds)
        for (int t = threadIdx.x; t < ntiles; t += 256) {
            const uint32_t c = hist[t];
            if (c) atomicAdd(&gcount[t], c);
        }
}

// exclusive scan of one batch element's tile counts; flags the element when its lists do not fit
// Launch order of the tiles (longest processing time first): a tile's weight class from the length of its list, heaviest
// class 0, empty tiles last.  Quarter-octave classes: the order inside a class does not matter for the schedule.
constexpr int ORDER_CLASSES = 64;
__device__ __forceinline__ int order_class(uint32_t entries, bool streams_all_faces)
{
    if (streams_all_faces) return 0;
    const int c = (int)(4.0f * __log2f((float)entries + 1.0f));
    return ORDER_CLASSES - 1 - min(ORDER_CLASSES - 1, c);
}

__global__ __launch_bounds__(256) void k_tile_offsets(const uint32_t* __restrict__ tile_count, int ntiles,
                                                       uint32_t list_cap, uint32_t* __restrict__ tile_off,
                                                       uint32_t* __restrict__ overflow, uint32_t* __restrict__ order_hist)
{
    __shared__ uint32_t wave_tot[4];
    __shared__ uint32_t carry_s;
    __shared__ uint32_t hist[ORDER_CLASSES];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid < ORDER_CLASSES) hist[tid] = 0u;
    const uint32_t* cnt = tile_count + (size_t)b * ntiles;
    uint32_t* off = tile_off + (size_t)b * (ntiles + 1);
    if (tid == 0) carry_s = 0u;
    __syncthreads();
    for (int base = 0; base < ntiles; base += 256) {
        const int t = base + tid;
        const uint32_t v = t < ntiles ? cnt[t] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d, 64);
            if ((tid & 63) >= d) incl += o;
        }
        if ((tid & 63) == 63) wave_tot[tid >> 6] = incl;
        __syncthreads();
        uint32_t wave_off = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const uint32_t x = wave_tot[w];
            if (w < (tid >> 6)) wave_off += x;
            total += x;
        }
        const uint32_t carry = carry_s;
        if (t < ntiles) off[t] = carry + wave_off + incl - v;
        __syncthreads();
        if (tid == 0) carry_s = carry + total;
        __syncthreads();
    }
    if (tid == 0) {
        off[ntiles] = carry_s;
        overflow[b] = carry_s > list_cap ? 1u : 0u;
    }
    const bool streams = carry_s > list_cap;
    for (int t = tid; t < ntiles; t += 256) atomicAdd(&hist[order_class(cnt[t], streams)], 1u);
    __syncthreads();
    if (tid < ORDER_CLASSES && hist[tid]) atomicAdd(&order_hist[tid], hist[tid]);
}

// tile_order[rank] = b * ntiles + tile, ranks grouped by weight class (class 0 first); one thread per tile
__global__ __launch_bounds__(256) void k_tile_order(const uint32_t* __restrict__ tile_count, int ntiles, int total,
                                                     const uint32_t* __restrict__ overflow,
                                                     const uint32_t* __restrict__ order_hist,
                                                     uint32_t* __restrict__ order_cursor, uint32_t* __restrict__ tile_order)
{
    __shared__ uint32_t base[ORDER_CLASSES];
    const int tid = threadIdx.x;
    if (tid < 64) {   // ORDER_CLASSES == 64: one wave scans the class sizes
        const uint32_t v = order_hist[tid];
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d, 64);
            if (tid >= d) incl += o;
        }
        base[tid] = incl - v;
    }
    __shared__ uint32_t lcnt[ORDER_CLASSES], lbase[ORDER_CLASSES];
    if (tid < ORDER_CLASSES) lcnt[tid] = 0u;
    __syncthreads();
    // two-level slot reservation (same-address global atomics serialise at ~5 ns: 60 % of the tiles are empty, one class):
    // the workgroup counts its tiles per class in LDS and reserves each class's run with ONE global atomic
    const int g = blockIdx.x * 256 + tid;
    int c = 0;
    uint32_t local = 0;
    if (g < total) {
        c = order_class(tile_count[g], overflow[g / ntiles] != 0u);
        local = atomicAdd(&lcnt[c], 1u);
    }
    __syncthreads();
    if (tid < ORDER_CLASSES && lcnt[tid]) lbase[tid] = base[tid] + atomicAdd(&order_cursor[tid], lcnt[tid]);
    __syncthreads();
    if (g < total) tile_order[lbase[c] + local] = (uint32_t)g;
}

// Appends every face to the list of each tile it touches.  Two-level slot reservation: the workgroup's 256 faces are
// first counted per tile in LDS, then ONE global atomic per (workgroup, touched tile) reserves a contiguous run of that
// tile's list for the whole workgroup, and the faces take their slots inside the run with LDS atomics.  Faces of an
// 85k-face mesh are a few pixels wide and consecutive face indices are neighbours on the surface, so a workgroup touches
// a few dozen tiles: ~30 global atomics per 256 faces instead of one per face x tile on a few hot counters (which was
// 10 % of the whole render step).  List order is irrelevant to the result (visibility is resolved by atomicMin on
// (depth, face index) keys).  Images with more tiles than the LDS tables hold fall back to per-entry global atomics.
__global__ __launch_bounds__(256) void k_tile_fill(const uint32_t* __restrict__ tilebox, int nf, int ntx,
                                                    const uint32_t* __restrict__ tile_off,
                                                    const uint32_t* __restrict__ overflow, uint32_t list_cap,
                                                    uint32_t* __restrict__ tile_cursor, uint32_t* __restrict__ tile_list)
{
    __shared__ uint32_t cnt[HIST_MAX];   // pass 1: entries of this workgroup per tile; pass 2: its running local cursor
    __shared__ uint32_t base[HIST_MAX];  // first slot of the workgroup's run in the tile's list
    const int b = blockIdx.y;
    if (overflow[b]) return;  // uniform over the block
    const int fn = blockIdx.x * 256 + threadIdx.x;
    const uint32_t v = fn < nf ? tilebox[(size_t)b * nf + fn] : TB_CULLED;
    const int tx0 = (int)(v & 255u), tx1 = (int)((v >> 8) & 255u), ty0 = (int)((v >> 16) & 255u), ty1 = (int)(v >> 24);
    const bool active = tx0 <= tx1;  // not culled
    const int ntiles = ntx * ntx;
    const uint32_t* off = tile_off + (size_t)b * (ntiles + 1);
    uint32_t* cur = tile_cursor + (size_t)b * ntiles;
    uint32_t* lst = tile_list + (size_t)b * list_cap;
    if (ntiles > HIST_MAX) {  // huge images: one global atomic per entry
        if (active)
            for (int ty = ty0; ty <= ty1; ty++)
                for (int tx = tx0; tx <= tx1; tx++) {
                    const int tt = ty * ntx + tx;
                    lst[off[tt] + atomicAdd(&cur[tt], 1u)] = (uint32_t)fn;
                }
        return;
    }
    for (int t = threadIdx.x; t < ntiles; t += 256) cnt[t] = 0u;
    __syncthreads();
    if (active)
        for (int ty = ty0; ty <= ty1; ty++)
            for (int tx = tx0; tx <= tx1; tx++) atomicAdd(&cnt[ty * ntx + tx], 1u);
    __syncthreads();
    for (int t = threadIdx.x; t < ntiles; t += 256) {
        const uint32_t c = cnt[t];
        if (c) {
            base[t] = off[t] + atomicAdd(&cur[t], c);
            cnt[t] = 0u;
        }
    }
    __syncthreads();
    if (active)
        for (int ty = ty0; ty <= ty1; ty++)
            for (int tx = tx0; tx <= tx1; tx++) {
                const int tt = ty * ntx + tx;
                lst[base[tt] + atomicAdd(&cnt[tt], 1u)] = (uint32_t)fn;
            }
}

struct PixelResult {
    int fn;
    float w[3];
    float zp;
    float rgb[3];
    float alpha;
};

// Evaluate everything the reference stores for one internal pixel from its resolved key.
__device__ __forceinline__ PixelResult shade_pixel(const FwdParams& P, int b, unsigned long long key, int gx, int gy)
{
    PixelResult r;
    const bool want_rgb = (P.flags & SDN_RGB) != 0;
    float bgc[3] = {0.f, 0.f, 0.f};
    if (want_rgb) {
        const float* c = P.bg + (P.bg_per_batch ? 3 * b : 0);
        bgc[0] = c[0];
        bgc[1] = c[1];
        bgc[2] = c[2];
    }
    if (key == ~0ull) {
        r.fn = -1;
        r.w[0] = r.w[1] = r.w[2] = 0.0f;
        r.zp = P.far_f;
        r.alpha = 0.0f;
        // rgb_map * mask + (1 - mask) * bg with rgb_map = 0, mask = 0
#pragma unroll
        for (int k = 0; k < 3; k++) r.rgb[k] = 0.0f * 0.0f + (1.0f - 0.0f) * bgc[k];
        return r;
    }
    const int fn = (int)(uint32_t)(key & 0xffffffffull);
    const size_t fidx = (size_t)b * P.nf + fn;
    float inv[9], f[9];
#pragma unroll
    for (int k = 0; k < 9; k++) inv[k] = P.face_inv[fidx * 9 + k];
#pragma unroll
    for (int k = 0; k < 9; k++) f[k] = P.faces[fidx * 9 + k];
    r.fn = fn;
    if (P.flags & SDN_K1_COVERAGE) {
        // K1 evaluates the weights on the x-sorted vertices (the sum of the clamped weights is taken in that order) and
        // scatters them back: weight_map[pi[k]] = w[k] (rasterize.py:185-205)
        int pi[3];
        k1_order(f, pi);
        float inv_s[9], ws[3];
#pragma unroll
        for (int l = 0; l < 3; l++)
#pragma unroll
            for (int k = 0; k < 3; k++) inv_s[3 * l + k] = pi[l] == 0 ? inv[k] : (pi[l] == 1 ? inv[3 + k] : inv[6 + k]);
        bary_weights(inv_s, gx, gy, ws);
#pragma unroll
        for (int l = 0; l < 3; l++) {
            if (pi[l] == 0) r.w[0] = ws[l];
            if (pi[l] == 1) r.w[1] = ws[l];
            if (pi[l] == 2) r.w[2] = ws[l];
        }
    } else {
        bary_weights(inv, gx, gy, r.w);
    }
    r.zp = ord_unbits((uint32_t)(key >> 32));
    r.alpha = 1.0f;
    r.rgb[0] = r.rgb[1] = r.rgb[2] = 0.0f;
    if (want_rgb) {
        const bool face_color = (P.flags & SDN_FACE_COLOR) != 0;
        const int ts = face_color ? 2 : P.ts;
        const TexCoord tc = texture_coord(r.w, r.zp, f[2], f[5], f[8], ts, P.eps);
        const float* tex = face_color ? (P.textures + fidx * 3) : (P.textures + fidx * (size_t)ts * ts * ts * 3);
        float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int pn = 0; pn < 8; pn++) {
            int isc;
            float wgt;
            texture_corner(tc, pn, ts, isc, wgt);
            const float* t = face_color ? tex : (tex + (long)isc * 3);
#pragma unroll
            for (int k = 0; k < 3; k++) acc[k] = acc[k] + wgt * t[k];
        }
#pragma unroll
        for (int k = 0; k < 3; k++) r.rgb[k] = acc[k] * 1.0f + (1.0f - 1.0f) * bgc[k];
    }
    return r;
}

// The tile epilogue shared by k_raster_tiles and k_raster_tiles_k1: one thread per 2x2 quad of internal pixels.
__device__ __forceinline__ void tile_epilogue(const FwdParams& P, const int b, const int X0, const int Y0,
                                              const unsigned long long* zbuf, const int tid)
{
    const int S = P.S;
    const bool aa = (P.flags & SDN_AA) != 0;
    const bool save = (P.flags & SDN_SAVE_MAPS) != 0;
    const bool lazy = (P.flags & SDN_LAZY_MAPS) != 0;
    const bool want_rgb = (P.flags & SDN_RGB) != 0;
    const bool want_alpha = (P.flags & SDN_ALPHA) != 0;
    const bool want_depth = (P.flags & SDN_DEPTH) != 0;
    const int qx = tid % (TS / 2), qy = tid / (TS / 2);
    const int R = aa ? S / 2 : S;
    float s_alpha = 0.f, s_depth = 0.f, s_rgb[3] = {0.f, 0.f, 0.f};
    bool any_valid = false;
    // order = the flipped image's pooling window: internal row 2qy+1 first (rasterize.py:953-966)
#pragma unroll
    for (int dyi = 0; dyi < 2; dyi++) {
#pragma unroll
        for (int dx = 0; dx < 2; dx++) {
            const int dy = 1 - dyi;
            const int px = 2 * qx + dx, py = 2 * qy + dy;
            const int gx = X0 + px, gy = Y0 + py;
            if (gx >= S || gy >= S) continue;
            any_valid = true;
            const PixelResult r = shade_pixel(P, b, zbuf[py * TS + px], gx, gy);
            const size_t q = ((size_t)b * S + gy) * S + gx;
            if (save) {
                P.face_index_map[q] = r.fn;
                P.depth_map[q] = r.zp;
            }
            if (save && !lazy) {
                P.weight_map[q * 3 + 0] = r.w[0];
                P.weight_map[q * 3 + 1] = r.w[1];
                P.weight_map[q * 3 + 2] = r.w[2];
                if (want_rgb) {
                    P.rgb_map[q * 3 + 0] = r.rgb[0];
                    P.rgb_map[q * 3 + 1] = r.rgb[1];
                    P.rgb_map[q * 3 + 2] = r.rgb[2];
                }
            }
            if (aa) {
                s_alpha = s_alpha + r.alpha;
                s_depth = s_depth + r.zp;
#pragma unroll
                for (int k = 0; k < 3; k++) s_rgb[k] = s_rgb[k] + r.rgb[k];
            } else {
                const size_t o = ((size_t)b * S + (S - 1 - gy)) * S + gx;
                if (want_alpha) P.alpha_out[o] = r.alpha;
                if (want_depth) P.depth_out[o] = r.zp;
                if (want_rgb) {
#pragma unroll
                    for (int k = 0; k < 3; k++)
                        P.rgb_out[(((size_t)b * 3 + k) * S + (S - 1 - gy)) * S + gx] = r.rgb[k];
                }
            }
        }
    }
    if (aa && any_valid) {
        const int oc = X0 / 2 + qx;
        const int orow = R - 1 - (Y0 / 2 + qy);
        const size_t o = ((size_t)b * R + orow) * R + oc;
        if (want_alpha) P.alpha_out[o] = s_alpha * 0.25f;
        if (want_depth) P.depth_out[o] = s_depth * 0.25f;
        if (want_rgb) {
#pragma unroll
            for (int k = 0; k < 3; k++) P.rgb_out[(((size_t)b * 3 + k) * R + orow) * R + oc] = s_rgb[k] * 0.25f;
        }
    }
}

// The hierarchical depth cull's two pieces (see k_raster_tiles), shared with the K1 tile kernel.
// lane l reads the pixels i * 64 + l (i < 16: row 2 i + (l >> 5), column l & 31): its column block is (l & 31) >> 3, its row block
// i >> 2; the 16 lanes of a column block ((l & 7) and (l >> 5) vary) then meet in four cross-lane maxima.
__device__ __forceinline__ void hiz_refresh(const uint32_t* zhi, uint32_t* blockmax, const int lane)
{
    uint32_t m[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 16; i++) m[i >> 2] = max(m[i >> 2], zhi[2 * (i * 64 + lane) + 1]);
#pragma unroll
    for (int rb = 0; rb < 4; rb++) {
        uint32_t v = m[rb];
        v = max(v, (uint32_t)__shfl_xor((int)v, 1, 64));
        v = max(v, (uint32_t)__shfl_xor((int)v, 2, 64));
        v = max(v, (uint32_t)__shfl_xor((int)v, 4, 64));
        v = max(v, (uint32_t)__shfl_xor((int)v, 32, 64));
        m[rb] = v;
    }
    if ((lane & 39) == 0) {   // lanes 0, 8, 16, 24: one per column block
#pragma unroll
        for (int rb = 0; rb < 4; rb++) blockmax[rb * 4 + (lane >> 3)] = m[rb];
    }
    __builtin_amdgcn_wave_barrier();
}
// a face whose conservative depth key zc lies behind the maxima of ALL 8 x 8 blocks its clipped box (tile pixel coordinates) touches
__device__ __forceinline__ bool hiz_hidden(const uint32_t* blockmax, const uint32_t zc, const int x0, const int x1, const int y0,
                                           const int y1)
{
    uint32_t zb = 0u;
    for (int by = y0 >> 3; by <= (y1 >> 3); by++)
        for (int bx = x0 >> 3; bx <= (x1 >> 3); bx++) zb = max(zb, blockmax[by * (TS / 8) + bx]);
    return zc > zb;
}

// COUNT: also tally the work (bench.py's ALU roofline): never used inside a timed region.
// HIZ (r05; r06: SDN_RASTER_HIZ = 0 | 1 | 2): hierarchical depth cull.  The tile keeps, per 8 x 8 pixel block, the LARGEST depth key among
// the block's current winners (0xffffffff while any pixel of the block is still uncovered).  A wave refreshes the 16 values before
// each batch it takes (16 LDS reads + a few cross-lane maxima per lane); winners only move nearer, so a value computed at any
// earlier time is still an upper bound -- stale or concurrently overwritten entries can only cull less.  A face whose conservative
// minimum depth (the `behind` test's zc) lies behind the maxima of ALL blocks its clipped box touches cannot win any of its pixels
// and is dropped before any candidate test.  Exact: the argument of `behind`, taken over a block -- maps stay bit-identical
// (every raster / renderer / CAD-golden test passes under either setting).
// r06: ON BY DEFAULT FOR THE TILES WITH LONG LISTS (HIZ == 2: the tiles that hand out batches of 64 as their waves become free; a short
// list is one batch per wave, all four started before anything is drawn -- the refresh can cull nothing there and was the whole
// regression of the always-on form): cad_like 289 -> 266 us, a frame of the reference's six templates 490 -> 313, the heaviest
// template alone 649 -> 529, car_like 196 -> 196 (tools/lab/hiz_ab.sh).  The r05 measurement of the always-on form
// (profiles/r05b_*, r05c_*: cad_like, 16 objects per launch): 79.4 M -> 64.3 M candidate tests, launch
// time 289.7 -> 288.8 us -- the refresh costs what the cull saves; on the six real ShapeNet meshes -13 % on the slowest one
// (544 -> 475 us) and +2...5 % on the other five.  Two extensions lost outright and were removed again: skipping single
// candidates of closed blocks in the wave-shared boxes (55.1 M candidates, +6 % time: one more LDS read per candidate), and
// ordering long tile lists near to far first (counting sort of <= 1024-entry chunks in LDS on the faces' depth keys, batches taken
// from the sorted chunk: 60.1 M candidates, 9.0 M instead of 10.3 M shaded hits, +5 % time on cad_like, +20...40 % on the real
// meshes).  Why the order buys nothing: four waves take the first four batches of a list at once, a median list IS two to eight
// batches, and a block only closes when all 64 of its pixels are covered -- near silhouettes and between parts they never are;
// the per-pixel `behind` test already catches 62 % of the passing candidates at the price of one LDS read.
template <bool COUNT, int HIZ>
__global__ __launch_bounds__(NTHR) void k_raster_tiles(const FwdParams P)
{
    unsigned n_cand = 0, n_in = 0, n_key = 0;
    // COUNT build: constant-clock ticks (wall_clock64: 100 MHz) this wave spent per phase (batch fetch, lane-private boxes, wave-shared boxes, thin
    // faces, epilogue) and the workgroup's total; summed / maximised over the launch into counters[3..]
    unsigned long long c_fetch = 0, c_small = 0, c_large = 0, c_thin = 0, c_epi = 0, t_mark = 0, t_begin = 0;
    auto tick = [&](unsigned long long& acc) {
        if constexpr (COUNT) {
            const unsigned long long now = wall_clock64();
            acc += now - t_mark;
            t_mark = now;
        }
    };
    if constexpr (COUNT) t_begin = t_mark = wall_clock64();
    __shared__ unsigned long long zbuf[TS * TS];
    __shared__ uint32_t q_fn[QCAP];
    __shared__ float xtab[TS], ytab[TS];
    __shared__ uint32_t q_count, next_batch, next_thin;
    __shared__ uint32_t hit_queue[NWAVE][128];       // per wave: (face slot | px << 6 | py << 11) of pixels that passed the edge tests
    __shared__ float face_rec[NWAVE][64 * FREC];     // per wave: the current batch's z0 z1 z2, inverse matrix, face index
    __shared__ uint32_t blockmax[(TS / 8) * (TS / 8)];   // HIZ: per 8 x 8 block an upper bound of its winners' depth keys
    static_assert(TS == 32, "the hi-z refresh maps 64 lanes x 16 reads onto a 32 x 32 tile");

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    // workgroups start in blockIdx order: the tiles with the longest lists first, so that the launch does not end with a
    // few heavy tiles running alone (a tile's cost spans two orders of magnitude; 60 % of a frame's tiles are empty)
    // (Measured and dropped, r04: rasterising the tiles with long lists as four 16 x 16 quadrants, one workgroup each -- the
    // counting build had shown the heaviest tile's waves alive for 212 us of a 300 us launch.  Every split made the launch
    // LONGER (lists >= 512 entries split: 243 / 360 us for car_like / cad_like against 197 / 299; profiles/r04_raster_split.log):
    // the launch is bound by the vector-ALU work summed over the tiles, not by its longest tile, and each quadrant walks the
    // tile's whole list again.  A side result worth keeping: of four workgroups per tile with three returning at once, the
    // one that works must sit at a HASHED position of its group -- at a fixed or slowly rotating position the dispatcher's
    // round-robin over XCDs and CUs put all working workgroups on a quarter of the CUs, 660 us.)
    const int ntiles_all = P.ntx * P.ntx;
    const uint32_t gid = P.tile_order[blockIdx.x];
    const int b = (int)(gid / (uint32_t)ntiles_all), tile = (int)(gid % (uint32_t)ntiles_all);
    const int tx = tile % P.ntx, ty = tile / P.ntx;
    const int X0 = tx * TS, Y0 = ty * TS;
    const int S = P.S, nf = P.nf;

    for (int i = tid; i < TS * TS; i += NTHR) zbuf[i] = ~0ull;
    if (tid < TS)
        xtab[tid] = pixel_to_ndc(X0 + tid, S);
    else if (tid < 2 * TS)
        ytab[tid - TS] = pixel_to_ndc(Y0 + tid - TS, S);
    if (tid == 0) q_count = next_batch = next_thin = 0;
    if (tid < (TS / 8) * (TS / 8)) blockmax[tid] = 0xffffffffu;
    __syncthreads();

    // the band path's first loads are issued now and consumed after the tile's own lists (two dependent round trips less at
    // the end of every tile); the list region holds 2 * nf records, so the speculative read of record `tid` stays inside it
    const float4* thin_cull = P.thin_list + (size_t)b * nf * 2;   // {nx, ny, a . n, band}
    const uint32_t n_thin = P.thin_count[b];
    const float4 thin_c0 = thin_cull[min(tid, 2 * nf - 1)];
    const uint32_t* tb = P.tilebox + (size_t)b * nf;
    const uint4* pbx = P.pixbox + (size_t)b * nf;
    const float* faces_b = P.faces + (size_t)b * nf * 9;
    const float* finv_b = P.face_inv + (size_t)b * nf * 9;

    // A wave takes 64 faces per batch: every lane first fetches ONE face of the batch (index, candidate box, 9 coordinates,
    // 9 inverse-matrix entries -- all loads of the batch are in flight together) and clips its box to the tile.
    //   * faces whose clipped box holds <= SMALL_AREA pixel centres (nearly all faces of an 85k-face mesh: a few pixels
    //     each) stay with their lane: 64 faces are rasterised at once, each lane walking its own box.  The previous
    //     one-face-per-wave scheme spent ~150 instructions per face to test ~20 pixels with a third of the lanes;
    //   * larger faces are broadcast with v_readlane and the 64 lanes share their pixels, as before.
    // A (face, pixel) pair is evaluated by the same float operations on either path and visibility is an atomicMin on
    // (depth, face) keys, so the split cannot change the result.  LDS traffic: 2 coordinate-table reads per candidate
    // pixel and the ds_min_u64 of pixels that are actually covered.
    auto shade_hit = [&](const float z0, const float z1, const float z2, const float (&inv)[9], const uint32_t qf,
                         const int px, const int py) {
        float bw[3];
        bary_weights(inv, X0 + px, Y0 + py, bw);
        const float zp = persp_depth(bw, z0, z1, z2);
        // rasterize.py:332,335 with the double comparisons folded into near_le / far_f on the host
        if (zp > P.near_le && zp < P.far_f) {
            if constexpr (COUNT) n_key++;
            const unsigned long long key = ((unsigned long long)ord_bits(zp) << 32) | qf;
            atomicMin(&zbuf[py * TS + px], key);
        }
    };
    // Covered pixels cost ~10x a rejected candidate (three barycentric divides, the perspective depth's reciprocals,
    // an LDS atomic) and only ~30 % of the candidates are covered, scattered over the lanes.  So the per-lane walk only
    // TESTS pixels and appends the hits to a per-wave LDS queue; whenever 64 hits are waiting all 64 lanes shade one
    // each, reading the hit's face record from the batch's LDS copy.
    uint32_t* hq = hit_queue[wave];
    float* frec = face_rec[wave];
    int hq_n = 0;  // wave-uniform
    auto drain = [&](const int count) {  // shade queue entries [0, count), count <= 64
        if (lane < count) {
            const uint32_t e = hq[lane];
            const int slot = (int)(e & 63u), px = (int)((e >> 6) & 31u), py = (int)((e >> 11) & 31u);
            const float* r = frec + slot * FREC;
            float inv[9];
#pragma unroll
            for (int k = 0; k < 9; k++) inv[k] = r[3 + k];
            shade_hit(r[0], r[1], r[2], inv, __float_as_uint(r[12]), px, py);
        }
    };
    // Conservative depth cull (exact): the perspective depth of a pixel is a weighted harmonic mean of the face's three
    // vertex depths, so it is >= their minimum up to a few ulp; a hit whose face minimum (lowered by 1e-5 relative) already
    // lies behind the pixel's current winner would lose the atomicMin anyway and is dropped before the seven divides of
    // shade_hit.  The winner only ever moves nearer, so a stale read can only cull less.  Meshes with interior geometry
    // (CAD files: depth complexity ~8) shade most covered pixels several times without it.
    const uint32_t* zhi = reinterpret_cast<const uint32_t*>(zbuf);
    auto behind = [&](const uint32_t zc, const int px, const int py) -> bool { return zc > zhi[2 * (py * TS + px) + 1]; };
    // HIZ == 2 (r06): the cull is switched per TILE -- on for tiles whose list is long (deep stacks of faces: where it pays), off for
    // the rest, which then only pay the test of this flag
    bool hiz_on = false;   // (workgroup-uniform)
    auto refresh_hiz = [&]() { hiz_refresh(zhi, blockmax, lane); };
    // append this iteration's hits (any lane subset) to the wave's queue; shade 64 as soon as 64 are waiting
    auto push_hits = [&](const bool hit, const uint32_t entry) {
        const unsigned long long hm = __ballot(hit);
        if (hm) {
            if (hit) hq[hq_n + (int)__popcll(hm & ((1ull << lane) - 1ull))] = entry;
            hq_n += (int)__popcll(hm);
            __builtin_amdgcn_wave_barrier();
            if (hq_n >= 64) {
                drain(64);
                __builtin_amdgcn_wave_barrier();
                const uint32_t rest = (lane < hq_n - 64) ? hq[64 + lane] : 0u;
                __builtin_amdgcn_wave_barrier();
                hq[lane] = rest;
                hq_n -= 64;
                __builtin_amdgcn_wave_barrier();
            }
        }
    };
    // ids[0..n), 1 <= n <= 64; called by a whole wave.  With fewer than 33 faces the wave gives each face k = 2^kshift
    // lanes (k * n <= 64) that interleave its pixels: a tile's ~100 list entries split over 4 waves leave ~25 faces per
    // wave, and the loop length is the largest box of the batch, not the number of faces.
    auto raster_batch = [&](const uint32_t* ids, const int n) {
        if constexpr (HIZ > 0) {
            if (HIZ == 1 || hiz_on) refresh_hiz();
        }
        const int kshift = 31 - __clz(64 / n);
        const int k = 1 << kshift;
        const int q = lane >> kshift, sub = lane & (k - 1);
        const bool mine = q < n;
        const uint32_t qf_l = mine ? ids[q] : 0u;
        uint4 pb_l = make_uint4(0u, 0u, 0u, 0u);
        float f_l[9], inv_l[9];
#pragma unroll
        for (int kk = 0; kk < 9; kk++) f_l[kk] = inv_l[kk] = 0.0f;
        if (mine) {
            pb_l = pbx[qf_l];
#pragma unroll
            for (int kk = 0; kk < 9; kk++) f_l[kk] = faces_b[(size_t)qf_l * 9 + kk];
#pragma unroll
            for (int kk = 0; kk < 9; kk++) inv_l[kk] = finv_b[(size_t)qf_l * 9 + kk];
        }
        const int lx0 = max((int)(pb_l.x & 0xffffu), X0), lx1 = min((int)(pb_l.x >> 16), X0 + TS - 1);
        const int ly0 = max((int)(pb_l.y & 0xffffu), Y0), ly1 = min((int)(pb_l.y >> 16), Y0 + TS - 1);
        const int lw = lx1 - lx0 + 1, lh = ly1 - ly0 + 1;
        int area_l = (mine && lw > 0 && lh > 0) ? lw * lh : 0;
        const float zmin_l = fminf(f_l[2], fminf(f_l[5], f_l[8]));
        const uint32_t zc_l = (zmin_l > 0.0f) ? ord_bits(zmin_l * 0.99999f) : 0u;   // 0: never culled
        if constexpr (HIZ > 0) {
            // hidden behind every block it touches: no candidate of this face can win
            if ((HIZ == 1 || hiz_on) && area_l > 0 && zc_l != 0u && hiz_hidden(blockmax, zc_l, lx0 - X0, lx1 - X0, ly0 - Y0, ly1 - Y0))
                area_l = 0;
        }
        // the batch's face records for the shading lanes (the previous batch's hits were drained before returning)
        if (mine && sub == 0) {
            float* r = frec + q * FREC;
            r[0] = f_l[2];
            r[1] = f_l[5];
            r[2] = f_l[8];
#pragma unroll
            for (int kk = 0; kk < 9; kk++) r[3 + kk] = inv_l[kk];
            r[12] = __uint_as_float(qf_l);
        }
        __builtin_amdgcn_wave_barrier();
        if constexpr (COUNT) {   // (the clock is read once the loaded values are in registers)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            tick(c_fetch);
        }
        // ---- small faces: k lanes each, test only
        {
            const bool small = area_l > 0 && area_l <= SMALL_AREA * k;
            const float rw = 1.0f / (float)(lw > 0 ? lw : 1);
            for (int i = sub; __ballot(small && i < area_l) != 0ull; i += k) {
                bool hit = false;
                int px = 0, py = 0;
                if (small && i < area_l) {
                    const int yy = (int)(((float)i + 0.5f) * rw);
                    const int xx = i - yy * lw;
                    px = lx0 - X0 + xx;
                    py = ly0 - Y0 + yy;
                    if constexpr (COUNT) n_cand++;
                    hit = inside_ndc(f_l, xtab[px], ytab[py]);
                    if constexpr (COUNT) n_in += hit ? 1u : 0u;
                    if (hit) hit = !behind(zc_l, px, py);
                }
                push_hits(hit, (uint32_t)q | ((uint32_t)px << 6) | ((uint32_t)py << 11));
            }
        }
        tick(c_small);
        // ---- large faces: the wave shares each one's box; hits join the same queue.  r04: the lanes walk ROW SPANS instead of
        // the whole clipped box.  The box's h <= 32 rows get 64 / 2^ceil(log2 h) lanes each; a row's lanes first bound the
        // pixels that can pass the edge tests -- per edge (a, b) the test (yp - ya)(xb - xa) >= (xp - xa)(yb - ya) is, for a
        // fixed row, a bound on xp (upper when yb > ya, lower when yb < ya); a pixel that passes the FLOAT tests lies within
        // the face's margin m of the exact triangle (face_margin_px), i.e. within m |ab| / |yb - ya| pixels of that bound
        // along the row; one more pixel covers the rounding of the bound itself (v_rcp, the NDC -> pixel map) -- and then
        // interleave over the span.  A superset of the passing pixels is tested with the exact predicate, as before; what
        // changes is how many fail: a long thin diagonal triangle fills a few percent of its box (CAD meshes: 27 % of all
        // candidate tests passed, 83 % of the candidates came from this path).
        unsigned long long big = __ballot(area_l > SMALL_AREA * k && sub == 0);
        const float fS = (float)S, fS1 = (float)(S - 1);
        while (big) {
            const int j = __builtin_amdgcn_readfirstlane(__ffsll((long long)big) - 1);
            big &= big - 1ull;
            const int x0 = __builtin_amdgcn_readlane(lx0, j), y0 = __builtin_amdgcn_readlane(ly0, j);
            const int w = __builtin_amdgcn_readlane(lw, j), h = __builtin_amdgcn_readlane(lh, j);
            const uint32_t zc = (uint32_t)__builtin_amdgcn_readlane((int)zc_l, j);
            const float mj = __int_as_float(__builtin_amdgcn_readlane((int)pb_l.z, j));
            const uint32_t slot = (uint32_t)(j >> kshift);
            float f[9];
#pragma unroll
            for (int kk = 0; kk < 9; kk++) f[kk] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f_l[kk]), j));
            if (w * h <= SPAN_AREA) {
                // a box of a few passes: the span set-up (~50 instructions) and its rows-to-lanes rounding cost more than
                // the failed tests it would save (measured: spans for every large face made the launch 4-10 % longer)
                const int area = w * h;
                // (Measured and dropped, r06: AXIS-ALIGNED passes -- a lane keeps one column (or row) of the box for the whole face,
                // so that its coordinate, one factor pair of each edge test, its zbuf address and queue entry are computed once per
                // face and a pass of 64 / 2^ceil(log2 w) rows only evaluates three subtract-multiply-compare triples: ~25 vector
                // instructions per pass in the ISA against ~45 of the linear walk below, taken when a box side fills >= 5/8 of its
                // power of two.  Bit-identical maps, and NOT faster: cad_like 288.6 vs 286.1 us, car_like 204.1 vs 197.5, the real
                // templates 520 vs 490 (mixed frame) and 699 vs 654 us (mesh 3776e4d1); fill thresholds 4/8 .. 7/8 within 2 % of each
                // other (profiles/r06c_raster_sweep2.log; the first cut with the orientation as a run-time select was 10 % slower,
                // r06c_raster_sweep.log).  The idle lanes of partly filled passes and the extra passes cost what the shorter
                // passes save: the walk is not bound by the instructions of the predicate.)
                const float rw = 1.0f / (float)w;
                for (int i0 = 0; i0 < area; i0 += 64) {
                    const int i = i0 + lane;
                    bool hit = false;
                    int px = 0, py = 0;
                    if (i < area) {
                        const int yy = (int)(((float)i + 0.5f) * rw);
                        const int xx = i - yy * w;
                        px = x0 - X0 + xx;
                        py = y0 - Y0 + yy;
                        if constexpr (COUNT) n_cand++;
                        hit = inside_ndc(f, xtab[px], ytab[py]);
                        if constexpr (COUNT) n_in += hit ? 1u : 0u;
                        if (hit) hit = !behind(zc, px, py);
                    }
                    push_hits(hit, slot | ((uint32_t)px << 6) | ((uint32_t)py << 11));
                }
                continue;
            }
            const int hs = h > 1 ? 32 - __clz(h - 1) : 0;          // ceil(log2 h) <= 5
            const int rshift = 6 - hs, per_row = 1 << rshift;      // lanes per row: 2 .. 64
            const int row = lane >> rshift, rsub = lane & (per_row - 1);
            const bool rvalid = row < h;
            const int py = y0 - Y0 + (rvalid ? row : 0);
            const float yp = ytab[py];
            const int bx0 = x0 - X0;
            float lo = (float)bx0, hi = (float)(bx0 + w - 1);
#pragma unroll
            for (int e = 0; e < 3; e++) {
                const float xa = f[3 * e], ya = f[3 * e + 1], xb = f[(3 * e + 3) % 9], yb = f[(3 * e + 4) % 9];
                const float D = xb - xa, E = yb - ya;
                if (E != 0.0f) {                                   // (wave-uniform; a horizontal edge bounds rows, not x)
                    const float rE = __builtin_amdgcn_rcpf(E);
                    const float xe = xa + (yp - ya) * D * rE;                      // NDC x where the row meets the edge's line
                    const float pe = (xe * fS + fS1) * 0.5f - (float)X0;           // ... as a pixel coordinate of this tile
                    const float sl = mj * __builtin_amdgcn_sqrtf(D * D + E * E) * fabsf(rE) + 1.0f;
                    // (fminf / fmaxf drop a NaN operand: 0 * inf from a denormal E leaves the bound where it was)
                    if (E > 0.0f)
                        hi = fminf(hi, pe + sl);
                    else
                        lo = fmaxf(lo, pe - sl);
                }
            }
            const int xl = max(bx0, (int)floorf(fminf(lo, 64.0f)));
            const int xr = min(bx0 + w - 1, (int)ceilf(fmaxf(hi, -64.0f)));
            const int len = rvalid ? xr - xl + 1 : 0;
            for (int i = rsub; __ballot(i < len) != 0ull; i += per_row) {
                bool hit = false;
                int px = 0;
                if (i < len) {
                    px = xl + i;
                    if constexpr (COUNT) n_cand++;
                    hit = inside_ndc(f, xtab[px], yp);
                    if constexpr (COUNT) n_in += hit ? 1u : 0u;
                    if (hit) hit = !behind(zc, px, py);
                }
                push_hits(hit, slot | ((uint32_t)px << 6) | ((uint32_t)py << 11));
            }
        }
        if (hq_n) {  // the face records change with the next batch
            drain(hq_n);
            hq_n = 0;
            __builtin_amdgcn_wave_barrier();
        }
        tick(c_large);
    };

    // ---- slivers / degenerate faces: band test around their line instead of a bounding box ---------------------
    // Meshes carry them (every pole triangle of a UV sphere has two coincident vertices; ~0.4 % of a CAD file's faces): such a
    // face can win pixels anywhere along the line through its longest edge.  thin_cull: the 256 threads look at different thin
    // faces each and queue the few whose band comes near this tile; thin_items: only those are tested, pixel by pixel.
    const float4* thin_recs = thin_cull + nf;                            // {ax, ay, face index bits, -}
    auto thin_cull_pass = [&]() {   // q_count == 0 on entry; barrier needed after
        const float cx = (float)X0 + 0.5f * (float)(TS - 1), cy = (float)Y0 + 0.5f * (float)(TS - 1);
        const float reach = 0.7072f * (float)TS + 0.4f;  // half diagonal of the tile (+ 1 pixel of slack)
#ifndef SDN_LAB_NO_THIN_CULL   // (lab: what testing every thin face of the object against every tile costs)
        for (uint32_t t = tid; t < n_thin; t += NTHR) {
            const float4 c = t < (uint32_t)NTHR ? thin_c0 : thin_cull[t];
            if (fabsf(cx * c.x + cy * c.y - c.z) > c.w + reach) continue;  // tile too far from the face's line
            const uint32_t slot = atomicAdd(&q_count, 1u);
            if (slot < (uint32_t)QCAP) q_fn[slot] = t;
        }
#endif
    };
    // work item = (queued face, group of ROWS_PER_ITEM tile rows); the waves claim 64 items at a time as they become free (no
    // barrier between a wave's last batch of the tile's list and its first items: visibility is an atomicMin either way).
    // A thread solves each row for the few pixels inside the band and tests only those.
    auto thin_items = [&]() {
#ifndef SDN_LAB_NO_THIN
        const uint32_t queued = q_count;
        const bool all = queued > (uint32_t)QCAP;  // queue overflow: walk the whole list (duplicates are harmless)
        const uint32_t n_loop = all ? n_thin : queued;
        constexpr int ROWS_PER_ITEM = 8, GROUPS = TS / ROWS_PER_ITEM;
        for (;;) {
            uint32_t first = 0;
            if (lane == 0) first = atomicAdd(&next_thin, 64u);
            first = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
            if (first >= n_loop * GROUPS) break;
            const uint32_t item = first + (uint32_t)lane;
            if (item >= n_loop * GROUPS) continue;
            const uint32_t c = item / GROUPS;
            const int g = (int)(item % GROUPS);
            const uint32_t t = all ? c : q_fn[c];
            const float4 e0 = thin_cull[t], e1 = thin_recs[t];
            const float nx = e0.x, ny = e0.y, band = e0.w, ax = e1.x, ay = e1.y;
            const uint32_t qf = __float_as_uint(e1.z);
            float f[9];
            bool loaded = false;
            for (int py = g * ROWS_PER_ITEM; py < (g + 1) * ROWS_PER_ITEM; py++) {
                // pixels of this row with |(x - ax) nx + (y - ay) ny| <= band: an interval in x (one pixel of slack either side;
                // the exact predicate below decides), the whole row when the line runs (nearly) along it
                const float off = ((float)(Y0 + py) - ay) * ny;
                int lo = 0, hi = TS - 1;
                if (fabsf(nx) > 1e-3f) {
                    const float r = 1.0f / nx;
                    const float xa = ax + (-band - off) * r, xb = ax + (band - off) * r;
                    const float x_lo = fminf(xa, xb) - (float)X0 - 1.0f, x_hi = fmaxf(xa, xb) - (float)X0 + 1.0f;
                    if (!(x_lo <= (float)(TS - 1)) || !(x_hi >= 0.0f)) continue;       // (also skips NaN)
                    lo = max(0, (int)fmaxf(x_lo, 0.0f));
                    hi = min(TS - 1, (int)fminf(x_hi, (float)(TS - 1)));
                } else if (fabsf(off + ((float)X0 + 0.5f * (float)(TS - 1) - ax) * nx) > band + 0.5f * (float)TS * fabsf(nx) + 1.0f) {
                    continue;
                }
                for (int px = lo; px <= hi; px++) {
                    const float dist = fabsf(((float)(X0 + px) - ax) * nx + ((float)(Y0 + py) - ay) * ny);
                    if (!(dist <= band)) continue;
                    if (!loaded) {
#pragma unroll
                        for (int k = 0; k < 9; k++) f[k] = faces_b[(size_t)qf * 9 + k];
                        loaded = true;
                    }
                    if (inside_ndc(f, xtab[px], ytab[py])) {
                        float inv[9];
#pragma unroll
                        for (int k = 0; k < 9; k++) inv[k] = finv_b[(size_t)qf * 9 + k];
                        float w[3];
                        bary_weights(inv, X0 + px, Y0 + py, w);
                        const float zp = persp_depth(w, f[2], f[5], f[8]);
                        if (zp > P.near_le && zp < P.far_f) {
                            const unsigned long long key = ((unsigned long long)ord_bits(zp) << 32) | qf;
                            atomicMin(&zbuf[py * TS + px], key);
                        }
                    }
                }
            }
        }
#endif
    };

    if (P.overflow[b] == 0u) {
        // ---- normal path: this tile's own face list, then the thin faces queued for it ---------------------------------
        const int ntiles = P.ntx * P.ntx;
        const uint32_t* off = P.tile_off + (size_t)b * (ntiles + 1);
        const uint32_t lo = off[tile], hi = off[tile + 1];
        thin_cull_pass();
        __syncthreads();
        tick(c_fetch);
        const uint32_t* lst = P.tile_list + (size_t)b * P.list_cap + lo;
        const int n_list = (int)(hi - lo);
        hiz_on = n_list >= HIZ_MIN_LIST;
        if (n_list <= 64 * NWAVE) {
            // a short list is cut into four equal runs, one batch per wave (fewer faces per batch = more lanes per face)
            const int per_wave = (n_list + NWAVE - 1) / NWAVE;
            const int run_lo = wave * per_wave, run_hi = min(n_list, run_lo + per_wave);
            if (run_lo < run_hi) raster_batch(lst + run_lo, run_hi - run_lo);
        } else {
            // a long one is handed out in batches of 64 as the waves become free: equal runs left the waves of the heavy
            // tiles -- the ones that decide when the launch ends -- waiting for the run that held the large faces
            for (;;) {
                int base = 0;
                if (lane == 0) base = (int)atomicAdd(&next_batch, 64u);
                base = __builtin_amdgcn_readfirstlane(base);
                if (base >= n_list) break;
                raster_batch(lst + base, min(64, n_list - base));
            }
        }
        thin_items();
        tick(c_thin);
    } else {
        // ---- fallback: stream every face's tile box, queue the hits in LDS, rasterise the queue --------------------
        for (int base = 0; base < nf; base += NTHR) {
            const int fn = base + tid;
            if (fn < nf) {
                const uint32_t v = tb[fn];
                const bool hit = (uint32_t)tx >= (v & 255u) && (uint32_t)tx <= ((v >> 8) & 255u) &&
                                 (uint32_t)ty >= ((v >> 16) & 255u) && (uint32_t)ty <= (v >> 24);
                if (hit) {
                    const uint32_t slot = atomicAdd(&q_count, 1u);
                    q_fn[slot] = (uint32_t)fn;
                }
            }
            __syncthreads();
            const int cnt = (int)q_count;
            __syncthreads();  // every thread has read q_count before the next chunk's atomicAdd can move it
            const bool last = base + NTHR >= nf;
            if (cnt >= NTHR || (last && cnt > 0)) {
                const int per_wave = (cnt + NWAVE - 1) / NWAVE;
                const int run_lo = wave * per_wave, run_hi = min(cnt, run_lo + per_wave);
                for (int qb = run_lo; qb < run_hi; qb += 64) raster_batch(q_fn + qb, min(64, run_hi - qb));
                __syncthreads();
                if (tid == 0) q_count = 0;
                __syncthreads();
            }
        }
        // (the streaming loop owns q_fn / q_count until here and leaves the count at zero behind a barrier)
        thin_cull_pass();
        __syncthreads();
        thin_items();
    }
    __syncthreads();
    tick(c_fetch);   // (waiting for the tile's slowest wave counts as fetch / idle time)


    tile_epilogue(P, b, X0, Y0, zbuf, tid);
    if constexpr (COUNT) {
        tick(c_epi);
        // (few atomics: same-address atomics serialise in L2 at ~5 ns each and would slow the launch they measure)
        __shared__ unsigned long long cnt[3];
        if (tid < 3) cnt[tid] = 0ull;
        __syncthreads();
        unsigned long long c[3] = {n_cand, n_in, n_key};
#pragma unroll
        for (int k = 0; k < 3; k++) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) c[k] += __shfl_xor(c[k], o, 64);
            if (lane == 0 && c[k]) atomicAdd(&cnt[k], c[k]);
        }
        __syncthreads();
        if (tid < 3 && cnt[tid]) atomicAdd(P.counters + tid, cnt[tid]);
        if (lane == 0 && (gid & 15u) == 0u) {   // phase clocks: a 1/16 sample of the tiles
            atomicAdd(P.counters + 3, c_fetch);
            atomicAdd(P.counters + 4, c_small);
            atomicAdd(P.counters + 5, c_large);
            atomicAdd(P.counters + 6, c_thin);
            atomicAdd(P.counters + 7, c_epi);
            atomicAdd(P.counters + 8, t_mark - t_begin);
            atomicMax(P.counters + 9, t_mark - t_begin);
            atomicAdd(P.counters + 10, 1ull);
        }
    }
}

// ---- SDN_K1_COVERAGE: the same tile organisation with the reference's default kernel's coverage rule (raster_math.h) -- what a
// `scripts/env.sh` user of the reference renders (NEURAL_RENDERER_UNSAFE=1 -> rasterize.py:102-236).
// r05: K1's walk only FINDS the covered pixels; they join the per-wave hit queue of k_raster_tiles and are shaded 64 at a time
// from the batch's LDS face records (sorted-vertex inverse matrix and depths: K1's barycentric order).  K1's rule needs no
// per-pixel test at all -- a column's covered rows are an interval (k1_column) -- so a lane's work per hit is the conservative
// depth cull and the queue append; the seven IEEE divides of a covered pixel, which the r04 kernel ran with whatever lanes
// happened to hold one, now always run 64 wide.  Same float operations per (face, pixel) pair, same ds_min_u64 resolve: the maps
// stay bit-equal to the oracle's K1 (tests/test_gpu_k1_coverage.py).
__global__ __launch_bounds__(NTHR) void k_raster_tiles_k1(const FwdParams P)
{
    __shared__ unsigned long long zbuf[TS * TS];
    __shared__ uint32_t next_batch;
    __shared__ uint32_t hit_queue[NWAVE][128];       // per wave: (face slot | px << 6 | py << 11) of covered pixels
    __shared__ float face_rec[NWAVE][64 * FREC];     // per wave: the batch's sorted z0 z1 z2, sorted inverse matrix, face index
    __shared__ uint32_t blockmax[(TS / 8) * (TS / 8)];   // r06: the hierarchical depth cull of k_raster_tiles, tiles with long lists
    bool hiz_on = false;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int ntiles = P.ntx * P.ntx;
    const uint32_t gid = P.tile_order[blockIdx.x];
    const int b = (int)(gid / (uint32_t)ntiles), tile = (int)(gid % (uint32_t)ntiles);
    const int tx = tile % P.ntx, ty = tile / P.ntx;
    const int X0 = tx * TS, Y0 = ty * TS;
    const int S = P.S, nf = P.nf;
    for (int i = tid; i < TS * TS; i += NTHR) zbuf[i] = ~0ull;
    if (tid == 0) next_batch = 0;
    if (tid < (TS / 8) * (TS / 8)) blockmax[tid] = 0xffffffffu;
    __syncthreads();
    const uint4* pbx = P.pixbox + (size_t)b * nf;
    const uint32_t* tb = P.tilebox + (size_t)b * nf;
    const float* faces_b = P.faces + (size_t)b * nf * 9;
    const float* finv_b = P.face_inv + (size_t)b * nf * 9;
    const uint32_t* zhi = reinterpret_cast<const uint32_t*>(zbuf);
    uint32_t* hq = hit_queue[wave];
    float* frec = face_rec[wave];
    int hq_n = 0;  // wave-uniform
    // one covered (face, pixel) pair: barycentrics on the sorted vertices, depth window, z-resolve (rasterize.py:185-210)
    auto drain = [&](const int count) {
        if (lane < count) {
            const uint32_t e = hq[lane];
            const int slot = (int)(e & 63u), px = (int)((e >> 6) & 31u), py = (int)((e >> 11) & 31u);
            const float* r = frec + slot * FREC;
            float inv_s[9], w[3];
#pragma unroll
            for (int k = 0; k < 9; k++) inv_s[k] = r[3 + k];
            bary_weights(inv_s, X0 + px, Y0 + py, w);
            const float zp = persp_depth(w, r[0], r[1], r[2]);
            if (zp > P.near_le && zp < P.far_f) {   // (rasterize.py:196 with the double comparisons folded, as k_raster_tiles)
                const unsigned long long key = ((unsigned long long)ord_bits(zp) << 32) | __float_as_uint(r[12]);
                atomicMin(&zbuf[py * TS + px], key);
            }
        }
    };
    // (the conservative depth cull of k_raster_tiles: a pixel's perspective depth is >= the face's nearest vertex depth up to
    // a few ulp, so a face whose minimum, lowered by 1e-5, lies behind the pixel's current winner cannot win it)
    auto behind = [&](const uint32_t zc, const int px, const int py) -> bool { return zc > zhi[2 * (py * TS + px) + 1]; };
    auto push_hits = [&](const bool hit, const uint32_t entry) {
        const unsigned long long hm = __ballot(hit);
        if (hm) {
            if (hit) hq[hq_n + (int)__popcll(hm & ((1ull << lane) - 1ull))] = entry;
            hq_n += (int)__popcll(hm);
            __builtin_amdgcn_wave_barrier();
            if (hq_n >= 64) {
                drain(64);
                __builtin_amdgcn_wave_barrier();
                const uint32_t rest = (lane < hq_n - 64) ? hq[64 + lane] : 0u;
                __builtin_amdgcn_wave_barrier();
                hq[lane] = rest;
                hq_n -= 64;
                __builtin_amdgcn_wave_barrier();
            }
        }
    };
    // a batch: lane l holds face ids[l] (have = l < n).  Boxes of up to K1_SMALL pixels are walked by their own lane; larger
    // ones are broadcast (v_readlane) and shared by the 64 lanes, two lanes per pixel column.
    constexpr int K1_SMALL = 32;
    auto raster_batch = [&](const bool have, const uint32_t fn) {
        if (hiz_on && SDN_K1_HIZ) hiz_refresh(zhi, blockmax, lane);
        float f[9], inv[9];
#pragma unroll
        for (int k = 0; k < 9; k++) f[k] = inv[k] = 0.0f;
        uint4 pb = make_uint4(0u, 0u, 0u, 0u);
        if (have) {
#pragma unroll
            for (int k = 0; k < 9; k++) f[k] = faces_b[(size_t)fn * 9 + k];
#pragma unroll
            for (int k = 0; k < 9; k++) inv[k] = finv_b[(size_t)fn * 9 + k];
            pb = pbx[fn];
        }
        const int lx0 = max((int)(pb.x & 0xffffu), X0), lx1 = min((int)(pb.x >> 16), X0 + TS - 1);
        const int ly0 = max((int)(pb.y & 0xffffu), Y0), ly1 = min((int)(pb.y >> 16), Y0 + TS - 1);
        const int lw = lx1 - lx0 + 1, lh = ly1 - ly0 + 1;
        int area = (have && lw > 0 && lh > 0) ? lw * lh : 0;
        const K1Face K = k1_setup(f, S);
        const float zmin = fminf(f[2], fminf(f[5], f[8]));
        const uint32_t zc_l = (zmin > 0.0f) ? ord_bits(zmin * 0.99999f) : 0u;   // 0: never culled
        // (the whole face behind every 8 x 8 block its box touches: none of its covered pixels can pass `behind` below)
        if (hiz_on && SDN_K1_HIZ && area > 0 && zc_l != 0u && hiz_hidden(blockmax, zc_l, lx0 - X0, lx1 - X0, ly0 - Y0, ly1 - Y0)) area = 0;
        if (area > 0) {   // the face's record for the shading lanes: rows in K1's sorted vertex order
            float* r = frec + lane * FREC;
#pragma unroll
            for (int l = 0; l < 3; l++) {
                const int k = K.pi[l];
                r[l] = k == 0 ? f[2] : (k == 1 ? f[5] : f[8]);
#pragma unroll
                for (int c = 0; c < 3; c++) r[3 + 3 * l + c] = k == 0 ? inv[c] : (k == 1 ? inv[3 + c] : inv[6 + c]);
            }
            r[12] = __uint_as_float(fn);
        }
        __builtin_amdgcn_wave_barrier();
        // ---- small boxes: K1's own loops clipped to the tile, one covered pixel per lane and iteration (columns whose
        // covered interval misses the box cost an iteration without a hit)
        {
            bool act = area > 0 && area <= K1_SMALL;
            int xi = lx0 - 1, yi = 0, ye = -1;
            while (__ballot(act) != 0ull) {
                bool hit = false;
                uint32_t entry = 0u;
                if (act) {
                    if (yi > ye) {
                        xi++;
                        if (xi > lx1) {
                            act = false;
                        } else {
                            int ya, yb;
                            k1_column(K, xi, S, ya, yb);
                            yi = max(ya, ly0);
                            ye = min(yb, ly1);
                        }
                    }
                    if (act && yi <= ye) {
                        const int px = xi - X0, py = yi - Y0;
                        hit = !behind(zc_l, px, py);
                        entry = (uint32_t)lane | ((uint32_t)px << 6) | ((uint32_t)py << 11);
                        yi++;
                    }
                }
                push_hits(hit, entry);
            }
        }
        // ---- large boxes: two lanes per column (w <= 32): even / odd rows of the column's covered interval
        unsigned long long big = __ballot(area > K1_SMALL);
        while (big) {
            const int j = __builtin_amdgcn_readfirstlane(__ffsll((long long)big) - 1);
            big &= big - 1ull;
            const int x0 = __builtin_amdgcn_readlane(lx0, j), y0 = __builtin_amdgcn_readlane(ly0, j);
            const int w = __builtin_amdgcn_readlane(lw, j), h = __builtin_amdgcn_readlane(lh, j);
            const uint32_t zc = (uint32_t)__builtin_amdgcn_readlane((int)zc_l, j);
            K1Face Kj;   // wave-uniform copy of lane j's set-up (the same values k1_setup would compute again)
#pragma unroll
            for (int l = 0; l < 3; l++) {
                Kj.px[l] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(K.px[l]), j));
                Kj.py[l] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(K.py[l]), j));
                Kj.pi[l] = 0;   // (not used by k1_column)
            }
            Kj.sa = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(K.sa), j));
            Kj.sb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(K.sb), j));
            Kj.sc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(K.sc), j));
            Kj.xi_min = __builtin_amdgcn_readlane(K.xi_min, j);
            Kj.xi_max = __builtin_amdgcn_readlane(K.xi_max, j);
            Kj.a_ok = __builtin_amdgcn_readlane((int)K.a_ok, j) != 0;
            Kj.b_ok = __builtin_amdgcn_readlane((int)K.b_ok, j) != 0;
            Kj.dead = false;
            const int col = lane >> 1;
            int ya = 0, yb = -1;
            if (col < w) {
                k1_column(Kj, x0 + col, S, ya, yb);
                ya = max(ya, y0);
                yb = min(yb, y0 + h - 1);
            }
            const int px = x0 + col - X0;
            for (int yi = ya + (lane & 1); __ballot(yi <= yb) != 0ull; yi += 2) {
                const bool in = yi <= yb;
                const int py = in ? yi - Y0 : 0;
                const bool hit = in && !behind(zc, in ? px : 0, py);
                push_hits(hit, (uint32_t)j | ((uint32_t)px << 6) | ((uint32_t)py << 11));
            }
        }
        if (hq_n) {  // the face records change with the next batch
            drain(hq_n);
            hq_n = 0;
            __builtin_amdgcn_wave_barrier();
        }
    };
    if (P.overflow[b] == 0u) {
        const uint32_t* off = P.tile_off + (size_t)b * (ntiles + 1);
        const uint32_t lo = off[tile], hi = off[tile + 1];
        const uint32_t* lst = P.tile_list + (size_t)b * P.list_cap + lo;
        const int n_list = (int)(hi - lo);
        hiz_on = P.hiz != 0 && n_list >= HIZ_MIN_LIST;   // (the K1 kernel knows only the per-tile form)
        for (;;) {   // batches of 64 faces, claimed by the waves as they become free
            int base = 0;
            if (lane == 0) base = (int)atomicAdd(&next_batch, 64u);
            base = __builtin_amdgcn_readfirstlane(base);
            if (base >= n_list) break;
            const bool have = base + lane < n_list;
            raster_batch(have, have ? lst[base + lane] : 0u);
        }
    } else {
        // the lists of this image overflowed: every tile looks at every face's tile box
        for (int base = 0; base < nf; base += NTHR) {
            const int fn = base + tid;
            bool have = false;
            if (fn < nf) {
                const uint32_t v = tb[fn];
                have = (uint32_t)tx >= (v & 255u) && (uint32_t)tx <= ((v >> 8) & 255u) && (uint32_t)ty >= ((v >> 16) & 255u) &&
                       (uint32_t)ty <= (v >> 24);
            }
            raster_batch(have, (uint32_t)fn);
        }
    }
    __syncthreads();
    tile_epilogue(P, b, X0, Y0, zbuf, tid);
}

// ---- SDN_K1_COVERAGE, the r04 kernel (kept selectable: SDN_K1_PLAIN=1): K1's own column / row loops clipped to the tile, covered
// pixels shaded where they are found and resolved straight into the LDS bin -- no hit queue, so the seven IEEE divides of a
// covered pixel run with whatever lanes happen to hold one (648 / 694 us per 16-object frame on car_like / cad_like against
// 199 / 291 us of k_raster_tiles).
__global__ __launch_bounds__(NTHR) void k_raster_tiles_k1_plain(const FwdParams P)
{
    __shared__ unsigned long long zbuf[TS * TS];
    __shared__ uint32_t next_batch;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int ntiles = P.ntx * P.ntx;
    const uint32_t gid = P.tile_order[blockIdx.x];
    const int b = (int)(gid / (uint32_t)ntiles), tile = (int)(gid % (uint32_t)ntiles);
    const int tx = tile % P.ntx, ty = tile / P.ntx;
    const int X0 = tx * TS, Y0 = ty * TS;
    const int S = P.S, nf = P.nf;
    for (int i = tid; i < TS * TS; i += NTHR) zbuf[i] = ~0ull;
    if (tid == 0) next_batch = 0;
    __syncthreads();
    const uint4* pbx = P.pixbox + (size_t)b * nf;
    const uint32_t* tb = P.tilebox + (size_t)b * nf;
    const float* faces_b = P.faces + (size_t)b * nf * 9;
    const float* finv_b = P.face_inv + (size_t)b * nf * 9;
    // one covered (face, pixel) pair: barycentrics on the sorted vertices, depth window, z-resolve
    // (the conservative depth cull of k_raster_tiles: a pixel's perspective depth is >= the face's nearest vertex depth up to
    // a few ulp, so a face whose minimum, lowered by 1e-5, lies behind the pixel's current winner cannot win it)
    const uint32_t* zhi = reinterpret_cast<const uint32_t*>(zbuf);
    auto k1_pixel = [&](const float (&inv_s)[9], const float (&zs)[3], const uint32_t fn, const int xi, const int yi) {
        const float zmin = fminf(zs[0], fminf(zs[1], zs[2]));
        if (zmin > 0.0f && ord_bits(zmin * 0.99999f) > zhi[2 * ((yi - Y0) * TS + (xi - X0)) + 1]) return;
        float w[3];
        bary_weights(inv_s, xi, yi, w);
        const float zp = persp_depth(w, zs[0], zs[1], zs[2]);
        if (zp > P.near_le && zp < P.far_f) {   // (rasterize.py:196 with the double comparisons folded, as k_raster_tiles)
            const unsigned long long key = ((unsigned long long)ord_bits(zp) << 32) | fn;
            atomicMin(&zbuf[(yi - Y0) * TS + (xi - X0)], key);
        }
    };
    auto sorted_rows = [&](const K1Face& K, const float (&f)[9], const float (&inv)[9], float (&inv_s)[9], float (&zs)[3]) {
#pragma unroll
        for (int l = 0; l < 3; l++) {
            const int k = K.pi[l];
            zs[l] = k == 0 ? f[2] : (k == 1 ? f[5] : f[8]);
#pragma unroll
            for (int c = 0; c < 3; c++) inv_s[3 * l + c] = k == 0 ? inv[c] : (k == 1 ? inv[3 + c] : inv[6 + c]);
        }
    };
    // a batch: lane l holds face ids[l] (have = l < n).  Boxes of up to K1_SMALL pixels are walked by their own lane; larger
    // ones are broadcast (v_readlane) and shared by the 64 lanes -- the organisation of k_raster_tiles, without its hit queue
    constexpr int K1_SMALL = 32;
    auto raster_batch = [&](const bool have, const uint32_t fn) {
        float f[9], inv[9];
#pragma unroll
        for (int k = 0; k < 9; k++) f[k] = inv[k] = 0.0f;
        uint4 pb = make_uint4(0u, 0u, 0u, 0u);
        if (have) {
#pragma unroll
            for (int k = 0; k < 9; k++) f[k] = faces_b[(size_t)fn * 9 + k];
#pragma unroll
            for (int k = 0; k < 9; k++) inv[k] = finv_b[(size_t)fn * 9 + k];
            pb = pbx[fn];
        }
        const int lx0 = max((int)(pb.x & 0xffffu), X0), lx1 = min((int)(pb.x >> 16), X0 + TS - 1);
        const int ly0 = max((int)(pb.y & 0xffffu), Y0), ly1 = min((int)(pb.y >> 16), Y0 + TS - 1);
        const int lw = lx1 - lx0 + 1, lh = ly1 - ly0 + 1;
        const int area = (have && lw > 0 && lh > 0) ? lw * lh : 0;
        if (area > 0 && area <= K1_SMALL) {
            const K1Face K = k1_setup(f, S);
            float inv_s[9], zs[3];
            sorted_rows(K, f, inv, inv_s, zs);
            for (int xi = lx0; xi <= lx1; xi++) {   // K1's own loops, clipped to the tile: only covered pixels are visited
                int ya, yb;
                k1_column(K, xi, S, ya, yb);
                for (int yi = max(ya, ly0); yi <= min(yb, ly1); yi++) k1_pixel(inv_s, zs, fn, xi, yi);
            }
        }
        unsigned long long big = __ballot(area > K1_SMALL);
        while (big) {
            const int j = __builtin_amdgcn_readfirstlane(__ffsll((long long)big) - 1);
            big &= big - 1ull;
            float g[9], ginv[9];
#pragma unroll
            for (int k = 0; k < 9; k++) {
                g[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f[k]), j));
                ginv[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(inv[k]), j));
            }
            const int x0 = __builtin_amdgcn_readlane(lx0, j), y0 = __builtin_amdgcn_readlane(ly0, j);
            const int w = __builtin_amdgcn_readlane(lw, j), h = __builtin_amdgcn_readlane(lh, j);
            const uint32_t gfn = (uint32_t)__builtin_amdgcn_readlane((int)fn, j);
            const K1Face K = k1_setup(g, S);
            float inv_s[9], zs[3];
            sorted_rows(K, g, ginv, inv_s, zs);
            // two lanes per column (w <= 32): even / odd rows of the column's covered range
            const int col = lane >> 1;
            if (col < w) {
                int ya, yb;
                k1_column(K, x0 + col, S, ya, yb);
                ya = max(ya, y0);
                yb = min(yb, y0 + h - 1);
                for (int yi = ya + (lane & 1); yi <= yb; yi += 2) k1_pixel(inv_s, zs, gfn, x0 + col, yi);
            }
        }
    };
    if (P.overflow[b] == 0u) {
        const uint32_t* off = P.tile_off + (size_t)b * (ntiles + 1);
        const uint32_t lo = off[tile], hi = off[tile + 1];
        const uint32_t* lst = P.tile_list + (size_t)b * P.list_cap + lo;
        const int n_list = (int)(hi - lo);
        for (;;) {   // batches of 64 faces, claimed by the waves as they become free
            int base = 0;
            if (lane == 0) base = (int)atomicAdd(&next_batch, 64u);
            base = __builtin_amdgcn_readfirstlane(base);
            if (base >= n_list) break;
            const bool have = base + lane < n_list;
            raster_batch(have, have ? lst[base + lane] : 0u);
        }
    } else {
        // the lists of this image overflowed: every tile looks at every face's tile box
        for (int base = 0; base < nf; base += NTHR) {
            const int fn = base + tid;
            bool have = false;
            if (fn < nf) {
                const uint32_t v = tb[fn];
                have = (uint32_t)tx >= (v & 255u) && (uint32_t)tx <= ((v >> 8) & 255u) && (uint32_t)ty >= ((v >> 16) & 255u) &&
                       (uint32_t)ty <= (v >> 24);
            }
            raster_batch(have, (uint32_t)fn);
        }
    }
    __syncthreads();
    tile_epilogue(P, b, X0, Y0, zbuf, tid);
}

}  // namespace sdn

using namespace sdn;

struct FwdWorkspace {
    size_t tilebox, pixbox, zeroed, tile_count, tile_cursor, overflow, thin_count, counters, order_hist, zeroed_bytes,
        tile_order, tile_off, tile_list, thin_list, total;
    uint32_t list_cap;
    int ntx, ntiles;
};

static size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

static FwdWorkspace workspace_layout(int bs, int nf, int S)
{
    FwdWorkspace w;
    const size_t n = (size_t)bs * nf;
    w.ntx = (S + TS - 1) / TS;
    w.ntiles = w.ntx * w.ntx;
    w.list_cap = (uint32_t)(8 * (size_t)nf + 4 * (size_t)w.ntiles);  // per batch element
    size_t o = 0;
    w.tilebox = o;
    o += align256(n * sizeof(uint32_t));
    w.pixbox = o;
    o += align256(n * sizeof(uint4));
    w.zeroed = o;  // tile_count | tile_cursor | overflow are cleared by one memset
    w.tile_count = o;
    o += align256((size_t)bs * w.ntiles * sizeof(uint32_t));
    w.tile_cursor = o;
    o += align256((size_t)bs * w.ntiles * sizeof(uint32_t));
    w.overflow = o;
    o += align256((size_t)bs * sizeof(uint32_t));
    w.thin_count = o;
    o += align256((size_t)bs * sizeof(uint32_t));
    w.counters = o;
    o += align256(16 * sizeof(unsigned long long));
    w.order_hist = o;   // [ORDER_CLASSES] class sizes, then [ORDER_CLASSES] cursors
    o += align256(2 * ORDER_CLASSES * sizeof(uint32_t));
    w.zeroed_bytes = o - w.zeroed;
    w.tile_order = o;
    o += align256((size_t)bs * w.ntiles * sizeof(uint32_t));
    w.tile_off = o;
    o += align256((size_t)bs * (w.ntiles + 1) * sizeof(uint32_t));
    w.tile_list = o;
    o += align256((size_t)bs * w.list_cap * sizeof(uint32_t));
    w.thin_list = o;
    o += align256(n * 2 * sizeof(float4));
    w.total = o;
    return w;
}

SDN_API int sdn_raster_workspace_bytes(int bs, int nf, int S, size_t* out)
{
    if (bs <= 0 || nf <= 0 || S <= 0 || !out) return fail(SDN_EINVAL, "sdn_raster_workspace_bytes: bad sizes");
    *out = workspace_layout(bs, nf, S).total;
    return SDN_OK;
}

namespace sdn {
// the body of sdn_rasterize_fwd; `src` (optional): build the faces from vertices inside k_face_setup (GATHER), `faces` is then
// src->faces_out
int rasterize_fwd_core(const FaceSource* src, const float* faces, const float* textures, int ts, int bs, int nf, int S, double near,
                       double far, double eps, const float* bg, int bg_per_batch, int flags, float* face_inv,
                       int32_t* face_index_map, float* weight_map, float* depth_map, float* rgb_map,
                       float* rgb_out, float* alpha_out, float* depth_out, void* workspace,
                       size_t workspace_bytes, sdnStream stream)
{
    if (!(flags & (SDN_RGB | SDN_ALPHA | SDN_DEPTH)))
        return fail(SDN_EINVAL, "sdn_rasterize_fwd: nothing to draw (rasterize.py:25-27)");
    if (!faces || !face_inv || bs <= 0 || nf <= 0 || S <= 0) return fail(SDN_EINVAL, "sdn_rasterize_fwd: bad faces/sizes");
    if (S > TS * 256) return fail(SDN_EINVAL, "sdn_rasterize_fwd: internal size %d > %d", S, TS * 256);
    if ((flags & SDN_AA) && (S & 1)) return fail(SDN_EINVAL, "sdn_rasterize_fwd: SDN_AA needs an even internal size");
    if (flags & SDN_RGB) {
        if (!textures || !bg || !rgb_out) return fail(SDN_EINVAL, "sdn_rasterize_fwd: rgb needs textures, bg, rgb_out");
        if (!(flags & SDN_FACE_COLOR) && ts < 2) return fail(SDN_EINVAL, "sdn_rasterize_fwd: texture size < 2");
    }
    if ((flags & SDN_ALPHA) && !alpha_out) return fail(SDN_EINVAL, "sdn_rasterize_fwd: alpha_out is NULL");
    if ((flags & SDN_DEPTH) && !depth_out) return fail(SDN_EINVAL, "sdn_rasterize_fwd: depth_out is NULL");
    if ((flags & SDN_SAVE_MAPS) && (!face_index_map || !depth_map))
        return fail(SDN_EINVAL, "sdn_rasterize_fwd: SDN_SAVE_MAPS needs the state maps");
    if ((flags & SDN_SAVE_MAPS) && !(flags & SDN_LAZY_MAPS) && (!weight_map || ((flags & SDN_RGB) && !rgb_map)))
        return fail(SDN_EINVAL, "sdn_rasterize_fwd: SDN_SAVE_MAPS needs the state maps");
    const FwdWorkspace W = workspace_layout(bs, nf, S);
    if (!workspace || workspace_bytes < W.total)
        return fail(SDN_ENOMEM, "sdn_rasterize_fwd: workspace %zu < %zu bytes", workspace_bytes, W.total);

    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    uint32_t* tilebox = (uint32_t*)(ws + W.tilebox);
    uint4* pixbox = (uint4*)(ws + W.pixbox);
    uint32_t* tile_count = (uint32_t*)(ws + W.tile_count);
    uint32_t* tile_cursor = (uint32_t*)(ws + W.tile_cursor);
    uint32_t* overflow = (uint32_t*)(ws + W.overflow);
    uint32_t* tile_off = (uint32_t*)(ws + W.tile_off);
    uint32_t* tile_list = (uint32_t*)(ws + W.tile_list);
    uint32_t* thin_count = (uint32_t*)(ws + W.thin_count);
    float4* thin_list = (float4*)(ws + W.thin_list);
    const int ntx = W.ntx;
    hipError_t me = hipMemsetAsync(ws + W.zeroed, 0, W.zeroed_bytes, st);
    if (me != hipSuccess) return fail(SDN_ELAUNCH, "hipMemsetAsync(tile counters): %s", hipGetErrorString(me));
    const dim3 face_grid(cdiv(nf, 256), bs);
    if (src && nf != (src->fill_back ? 2 : 1) * src->nf0) return fail(SDN_EINVAL, "rasterize_fwd_core: nf %d does not match the face source (%d faces, fill_back %d)", nf, src->nf0, src->fill_back);
    const dim3 setup_grid(cdiv((src && src->fill_back) ? src->nf0 : nf, 256), bs);   // (the fused build takes a face and its twin per thread)
    const int k1 = (flags & SDN_K1_COVERAGE) ? 1 : 0;
    FaceSource G = FaceSource();   // (value-initialised: null pointers, zeros)
    if (src) G = *src;
#define SDN_FACE_SETUP(K1_, GATHER_)                                                                                          \
    hipLaunchKernelGGL((k_face_setup<K1_, GATHER_>), setup_grid, dim3(256), 0, st, faces, nf, S, ntx, face_inv, tilebox, pixbox, \
                       tile_count, thin_count, thin_list, G)
    if (k1 && src) SDN_FACE_SETUP(true, true);
    else if (k1) SDN_FACE_SETUP(true, false);
    else if (src) SDN_FACE_SETUP(false, true);
    else SDN_FACE_SETUP(false, false);
#undef SDN_FACE_SETUP
    int rc = check_launch("k_face_setup");
    if (rc) return rc;
    const uint32_t list_cap = (flags & SDN_STREAM_FACES) ? 0u : W.list_cap;
    uint32_t* order_hist = (uint32_t*)(ws + W.order_hist);
    uint32_t* tile_order = (uint32_t*)(ws + W.tile_order);
    hipLaunchKernelGGL(k_tile_offsets, dim3(bs), dim3(256), 0, st, tile_count, W.ntiles, list_cap, tile_off, overflow,
                       order_hist);
    if ((rc = check_launch("k_tile_offsets"))) return rc;
    hipLaunchKernelGGL(k_tile_order, dim3(cdiv(bs * W.ntiles, 256)), dim3(256), 0, st, tile_count, W.ntiles, bs * W.ntiles,
                       overflow, order_hist, order_hist + ORDER_CLASSES, tile_order);
    if ((rc = check_launch("k_tile_order"))) return rc;
    hipLaunchKernelGGL(k_tile_fill, face_grid, dim3(256), 0, st, tilebox, nf, ntx, tile_off, overflow, W.list_cap,
                       tile_cursor, tile_list);
    if ((rc = check_launch("k_tile_fill"))) return rc;

    FwdParams P;
    P.faces = faces;
    P.textures = textures;
    P.bg = bg;
    P.face_inv = face_inv;
    P.face_index_map = face_index_map;
    P.weight_map = weight_map;
    P.depth_map = depth_map;
    P.rgb_map = rgb_map;
    P.rgb_out = rgb_out;
    P.alpha_out = alpha_out;
    P.depth_out = depth_out;
    P.tilebox = tilebox;
    P.pixbox = pixbox;
    P.tile_off = tile_off;
    P.tile_list = tile_list;
    P.overflow = overflow;
    P.tile_order = tile_order;
    P.thin_count = thin_count;
    P.thin_list = thin_list;
    P.counters = (unsigned long long*)(ws + W.counters);
    P.list_cap = W.list_cap;
    P.eps = eps;
    P.ts = ts;
    P.bs = bs;
    P.nf = nf;
    P.S = S;
    P.ntx = ntx;
    P.flags = flags;
    P.bg_per_batch = bg_per_batch;
    // rasterize.py:332: `zp <= near || far <= zp` compares in double against the pasted literals, and
    // :297,335 keep zp < (float)far.  near_le = largest float <= near, so (double)zp <= near <=> zp <= near_le.
    float near_le = (float)near;
    if ((double)near_le > near) near_le = nextafterf(near_le, -INFINITY);
    P.near_le = near_le;
    P.far_f = (float)far;
    // SDN_RASTER_HIZ: whole faces culled against the tile's 8 x 8 block maxima (see k_raster_tiles): 0 off, 1 in every tile, 2 [default since
    // r06] in the tiles with long lists; read once per process
    static const int hiz = [] { const char* e = getenv("SDN_RASTER_HIZ"); return e && e[0] >= '0' && e[0] <= '2' ? e[0] - '0' : 2; }();
    P.hiz = hiz;
    if (k1) {
        if (flags & SDN_COUNT_WORK) return fail(SDN_EINVAL, "sdn_rasterize_fwd: SDN_COUNT_WORK is not built for SDN_K1_COVERAGE");
        TimedLaunch timed(TIME_RASTER_TILES_K1, st, 0.0);
        static const bool plain = [] { const char* e = getenv("SDN_K1_PLAIN"); return e && e[0] == '1'; }();
        if (plain)   // the r04 kernel (no hit queue), kept for A/B measurements
            hipLaunchKernelGGL(k_raster_tiles_k1_plain, dim3(ntx * ntx * bs), dim3(NTHR), 0, st, P);
        else
            hipLaunchKernelGGL(k_raster_tiles_k1, dim3(ntx * ntx * bs), dim3(NTHR), 0, st, P);
        return check_launch("k_raster_tiles_k1");
    }
    const dim3 grid(ntx * ntx * bs);
    if (flags & SDN_COUNT_WORK) {
        if (hiz == 0)
            hipLaunchKernelGGL((k_raster_tiles<true, 0>), grid, dim3(NTHR), 0, st, P);
        else if (hiz == 1)
            hipLaunchKernelGGL((k_raster_tiles<true, 1>), grid, dim3(NTHR), 0, st, P);
        else
            hipLaunchKernelGGL((k_raster_tiles<true, 2>), grid, dim3(NTHR), 0, st, P);
    } else {
        TimedLaunch timed(TIME_RASTER_TILES, st, 0.0);
        if (hiz == 0)
            hipLaunchKernelGGL((k_raster_tiles<false, 0>), grid, dim3(NTHR), 0, st, P);
        else if (hiz == 1)
            hipLaunchKernelGGL((k_raster_tiles<false, 1>), grid, dim3(NTHR), 0, st, P);
        else
            hipLaunchKernelGGL((k_raster_tiles<false, 2>), grid, dim3(NTHR), 0, st, P);
    }
    return check_launch("k_raster_tiles");
}
}  // namespace sdn

SDN_API int sdn_rasterize_fwd(const float* faces, const float* textures, int ts, int bs, int nf, int S, double near,
                              double far, double eps, const float* bg, int bg_per_batch, int flags, float* face_inv,
                              int32_t* face_index_map, float* weight_map, float* depth_map, float* rgb_map,
                              float* rgb_out, float* alpha_out, float* depth_out, void* workspace,
                              size_t workspace_bytes, sdnStream stream)
{
    return rasterize_fwd_core(nullptr, faces, textures, ts, bs, nf, S, near, far, eps, bg, bg_per_batch, flags, face_inv,
                              face_index_map, weight_map, depth_map, rgb_map, rgb_out, alpha_out, depth_out, workspace,
                              workspace_bytes, stream);
}

// Re-derive the weight (and colour) maps an SDN_LAZY_MAPS forward left out: per internal pixel the forward's own shade_pixel
// on (face index, stored depth) -- the z-buffer key the forward shaded from is exactly (ord(depth) << 32 | face), so the
// results are the forward's, bit for bit.
__global__ __launch_bounds__(256) void k_reshade_maps(const FwdParams P, long npx)
{
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= npx) return;
    const int S = P.S;
    const int gx = (int)(q % S), gy = (int)((q / S) % S), b = (int)(q / ((long)S * S));
    const int fn = P.face_index_map[q];
    const unsigned long long key = fn < 0 ? ~0ull : (((unsigned long long)ord_bits(P.depth_map[q]) << 32) | (uint32_t)fn);
    const PixelResult r = shade_pixel(P, b, key, gx, gy);
    P.weight_map[q * 3 + 0] = r.w[0];
    P.weight_map[q * 3 + 1] = r.w[1];
    P.weight_map[q * 3 + 2] = r.w[2];
    if ((P.flags & SDN_RGB) && P.rgb_map) {
        P.rgb_map[q * 3 + 0] = r.rgb[0];
        P.rgb_map[q * 3 + 1] = r.rgb[1];
        P.rgb_map[q * 3 + 2] = r.rgb[2];
    }
}

namespace sdn {
int launch_reshade_maps(const float* faces, const float* textures, int ts, int bs, int nf, int S, double far, double eps,
                        const float* bg, int bg_per_batch, int flags, const float* face_inv, const int32_t* face_index_map,
                        const float* depth_map, float* weight_map, float* rgb_map, hipStream_t st)
{
    if (!faces || !face_inv || !face_index_map || !depth_map || !weight_map) return fail(SDN_EINVAL, "reshade: null pointer");
    if ((flags & SDN_RGB) && (!textures || !bg || !rgb_map)) return fail(SDN_EINVAL, "reshade: colour maps need textures, bg, rgb_map");
    FwdParams P = {};
    P.faces = faces;
    P.textures = textures;
    P.bg = bg;
    P.face_inv = const_cast<float*>(face_inv);
    P.face_index_map = const_cast<int32_t*>(face_index_map);
    P.depth_map = const_cast<float*>(depth_map);
    P.weight_map = weight_map;
    P.rgb_map = rgb_map;
    P.eps = eps;
    P.ts = ts;
    P.bs = bs;
    P.nf = nf;
    P.S = S;
    P.flags = flags;
    P.bg_per_batch = bg_per_batch;
    P.far_f = (float)far;
    const long npx = (long)bs * S * S;
    hipLaunchKernelGGL(k_reshade_maps, dim3(cdiv(npx, 256)), dim3(256), 0, st, P, npx);
    return check_launch("k_reshade_maps");
}
}  // namespace sdn

SDN_API int sdn_raster_work_counters(const void* workspace, int bs, int nf, int S, unsigned long long* out3, sdnStream stream)
{
    if (!workspace || !out3 || bs <= 0 || nf <= 0 || S <= 0) return fail(SDN_EINVAL, "sdn_raster_work_counters: bad arguments");
    const FwdWorkspace W = workspace_layout(bs, nf, S);
    hipStream_t st = (hipStream_t)stream;
    if (hipMemcpyAsync(out3, (const char*)workspace + W.counters, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st) !=
            hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return fail(SDN_ELAUNCH, "sdn_raster_work_counters: copy failed");
    return SDN_OK;
}

SDN_API int sdn_raster_phase_clocks(const void* workspace, int bs, int nf, int S, unsigned long long* out8, sdnStream stream)
{
    if (!workspace || !out8 || bs <= 0 || nf <= 0 || S <= 0) return fail(SDN_EINVAL, "sdn_raster_phase_clocks: bad arguments");
    const FwdWorkspace W = workspace_layout(bs, nf, S);
    hipStream_t st = (hipStream_t)stream;
    if (hipMemcpyAsync(out8, (const char*)workspace + W.counters + 3 * sizeof(unsigned long long), 8 * sizeof(unsigned long long),
                       hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return fail(SDN_ELAUNCH, "sdn_raster_phase_clocks: copy failed");
    return SDN_OK;
}
