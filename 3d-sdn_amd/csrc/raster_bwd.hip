// Backward of the rasterizer for gfx950.
//
// Replaces Rasterize.backward_gpu (/root/reference/geometric/neural_renderer/rasterize.py:846-886):
//
//   K5 (rasterize.py:523-745), the hand-crafted silhouette / colour gradient wrt the x,y of every front face's
//   vertices.  The reference runs one thread per face that walks each edge pixel by pixel and, for every edge
//   pixel, scans a whole image row or column serially -- O(edge length x image size) per thread, so a single
//   large triangle stalls the launch (measured here: 31 ms per 85k-face object with the literal port).
//   CDNA4 plan: only faces that own at least one pixel can contribute, so
//     k_mark_visible  flags them from face_index_map;
//     k_edge_plan     (1 thread / face) cuts each (edge, axis) walk into chunks of <= 8 edge pixels and
//                     allocates them with one atomic per face;
//     k_edge_scan     persistent waves, one chunk at a time: the 64 lanes stride over the row / column
//                     segment of every edge pixel (coalesced along rows), keep private partial sums and
//                     butterfly-reduce once per chunk;
//     k_edge_reduce   (1 thread / face) adds the chunk results in edge / axis / pixel order.
//   Every partial sum has a fixed order, so the result is deterministic run to run (the reference's is too);
//   it differs from the reference's strictly serial sum by float re-association only (~1e-7 relative).
//   Faces whose chunks do not fit the workspace fall back to the literal serial walk inside k_edge_reduce.
//
//   K6 + K7 (rasterize.py:756-789, 800-844) in k_bwd_pixels: per covered pixel, scatter the colour gradient into
//   the face's texture (or per-face colour) and the depth gradient into the winning face's 9 coordinates with
//   hardware float atomics (the reference uses atomicAdd as well).
//
// Upstream gradients arrive at output resolution (after vertical flip and 2x2 average pooling,
// rasterize.py:951-966); they are expanded on the fly: g_ss(y, x) = 0.25 * g_out(R-1-y/2, x/2).
#include "raster_math.h"
#include "sdn_common.h"

namespace sdn {

#ifndef SDN_EDGE_CHUNK
#define SDN_EDGE_CHUNK 8
#endif
constexpr int CHUNK = SDN_EDGE_CHUNK;  // edge pixels per scan chunk (4 or 8; a lab build may set it: r06 measured 4 -- edge kernels -5 ... -23 us,
                                       // k_edge_plan / k_edge_reduce + that and more: frame step +23 us on cad_like, 0 on car_like)
static_assert(CHUNK == 4 || CHUNK == 8, "k_edge_scan_sil / k_chunk_sum are written for 4 or 8");

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
    // K5 plan
    uint32_t* visible;      // [bs*nf]
    int32_t* chunk_base;    // [bs*nf]  >= 0 first chunk, -1 serial fallback, -2 contributes nothing
    uint32_t* counter;      // [1] chunks allocated so far
    uint4* chunk_desc;      // [cap] {global face, edge*2+axis, d0_start, count}
    float2* chunk_out;      // [cap]
    const float* hmap;      // [bs,S,S]  silhouette-only fast path: max(-g_alpha, 0) on background pixels, else 0
    const float* hmapT;     // [bs,S,S]  the same, transposed (column scans become coalesced)
    // sparse form of hmap / hmapT rows (k_compact_rows): h is zero on every covered pixel and wherever the upstream
    // gradient is >= 0, and K5's "out" scans run from an edge all the way to the image border, so most of what they
    // read is zero.  row r = (b * S + line) of the row-major map, rows [bs*S, 2*bs*S) the transposed one.
    const uint16_t* nz_cnt;  // [2*bs*S, S+1]  number of non-zeros in [0, x)
    const uint16_t* nz_pos;  // [2*bs*S, S]    their positions, ascending
    const float* nz_val;     // [2*bs*S, S]    their values
    // "out" scans grouped by the row they run along (k_edge_scan_sil files them, k_edge_rows evaluates them)
    uint32_t* row_cnt;       // [2*bs*S, 2]     owners filed per row: scans towards larger / smaller positions (zeroed per call)
    struct OwnerRec* own_rec;  // [2*bs*S, 3*S]  at most 3 owners per pixel of the row (one per edge of the pixel's face)
    float2* own_out;         // [cap*8]         result of the owner in slot chunk*8 + lane
    uint8_t* chunk_mask;     // [cap]           which of a chunk's 8 edge pixels own an "out" scan
    uint32_t cap;
    double eps;
    int ts, bs, nf, S, flags;
    VertexSink sink;         // sink.grad_verts != null: k_edge_reduce adds to the vertices instead of writing grad_faces rows
};

// One "out" scan of K5 (rasterize.py:600-656) reduced to what its terms need: sum over the row's non-zero list [k0, k1) of
// val / (t * (pos - cross) +- eps) for the edge's two end points.
struct OwnerRec {
    float t1, t0, cross;
    uint32_t kk;    // k0 | k1 << 16
    uint32_t slot;  // chunk * CHUNK + lane | (nz1 | nz0 << 1) << 30
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

// Geometry of one (edge, axis) walk of K5 (rasterize.py:540-566).
struct EdgeWalk {
    float p[3][2];  // (d0, d1) of the edge's two end points and of the opposite vertex
    int pi0, pi1;   // vertex numbers of the end points
    int direction;  // +1 / -1: where "outside" lies along d1
    int d0_from, d0_to;
};

__device__ __forceinline__ EdgeWalk edge_walk(const float face[9], int edge_num, int axis, float is_f)
{
    EdgeWalk w;
    int pi[3];
    float pp[3][2];
#pragma unroll
    for (int num = 0; num < 3; num++) {
        pi[num] = (edge_num + num) % 3;
        pp[num][0] = ndc_to_pixel(face[3 * pi[num] + 0], is_f);
        pp[num][1] = ndc_to_pixel(face[3 * pi[num] + 1], is_f);
    }
#pragma unroll
    for (int num = 0; num < 3; num++) {
        w.p[num][0] = pp[num][axis];
        w.p[num][1] = pp[num][1 - axis];
    }
    w.pi0 = pi[0];
    w.pi1 = pi[1];
    if (axis == 0)
        w.direction = (w.p[0][0] < w.p[1][0]) ? -1 : 1;
    else
        w.direction = (w.p[0][0] < w.p[1][0]) ? 1 : -1;
    w.d0_from = cvt_i32(fmaxf(ceilf(fminf(w.p[0][0], w.p[1][0])), 0.0f));
    w.d0_to = cvt_i32(fminf(fmaxf(w.p[0][0], w.p[1][0]), is_f - 1.0f));
    return w;
}

// One edge pixel d0 of a walk: both passes, with the d1 loops strided by (lane, nlanes).  With lane = 0,
// nlanes = 1 this is exactly the reference's serial loop (rasterize.py:567-728).
__device__ __forceinline__ void edge_pixel(const BwdParams& P, const MapReader& M, const EdgeWalk& w, int fn, int axis,
                                           int d0, bool use_alpha, bool use_rgb, int lane, int nlanes, float& acc0,
                                           float& acc1)
{
    const int S = P.S;
    const float is_f = (float)S;
    const float fd0 = (float)d0;
    float d1_cross = (w.p[1][1] - w.p[0][1]) / (w.p[1][0] - w.p[0][0]);
    d1_cross = d1_cross * (fd0 - w.p[0][0]);
    d1_cross = d1_cross + w.p[0][1];
    const int d1_in = (0 < w.direction) ? cvt_i32(floorf(d1_cross)) : cvt_i32(ceilf(d1_cross));
    const int d1_out = d1_in + w.direction;
    if (d1_in < 0 || S <= d1_in) return;
    if (d1_out < 0 || S <= d1_out) return;

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
#pragma unroll
        for (int k = 0; k < 3; k++) {
            rgb_in[k] = M.rgb(xin, yin, k);
            rgb_out[k] = M.rgb(xout, yout, k);
        }
    }
    const bool nz1 = w.p[1][0] != fd0, nz0 = w.p[0][0] != fd0;
    const float den1 = w.p[1][0] - fd0, den0 = fd0 - w.p[0][0];
    const float* hline = nullptr;
    if (P.hmap && use_alpha && !use_rgb) hline = (axis == 0 ? P.hmapT : P.hmap) + ((size_t)M.b * S + d0) * S;

    // "out" pass (rasterize.py:600-656): from the pixel just outside the edge to the image border
    if (f_in == fn && hline) {
        // silhouette-only: alpha_in = 1, so diff_grad = (alpha(p) - 1) * g(p) is positive only on background pixels
        // with g < 0, where it equals -g: exactly what k_hmap stored (0 elsewhere).  Only the non-zero entries of the
        // segment are visited (coalesced reads of the row's compact list); the skipped terms are exactly the ones the
        // reference drops with `if (diff_grad <= 0) continue` (rasterize.py:644,715).
        const int d1_limit = (0 < w.direction) ? S - 1 : 0;
        const int d1_from = max(min(d1_out, d1_limit), 0);
        const int d1_to = min(max(d1_out, d1_limit), S - 1);
        const size_t row = (size_t)(axis == 0 ? P.bs : 0) * S + (size_t)M.b * S + d0;
        const uint16_t* cnt = P.nz_cnt + row * (S + 1);
        const int k1 = cnt[d1_to + 1];
        for (int k = cnt[d1_from] + lane; k < k1; k += nlanes) {
            const int d1 = P.nz_pos[row * S + k];
            const float diff_grad = P.nz_val[row * S + k];
            if (nz1) acc0 -= diff_grad / edge_dist(w.p[0][0], w.p[1][0], den1, d1, d1_cross, is_f, P.eps);
            if (nz0) acc1 -= diff_grad / edge_dist(w.p[0][0], w.p[1][0], den0, d1, d1_cross, is_f, P.eps);
        }
    } else if (f_in == fn) {
        const int d1_limit = (0 < w.direction) ? S - 1 : 0;
        const int d1_from = max(min(d1_out, d1_limit), 0);
        const int d1_to = min(max(d1_out, d1_limit), S - 1);
        for (int d1 = d1_from + lane; d1 <= d1_to; d1 += nlanes) {
            const int x = axis == 0 ? d0 : d1, y = axis == 0 ? d1 : d0;
            float diff_grad = 0.0f;
            if (use_alpha) diff_grad = diff_grad + (M.alpha(x, y) - alpha_in) * M.g_alpha(x, y);
            if (use_rgb) {
#pragma unroll
                for (int k = 0; k < 3; k++) diff_grad = diff_grad + (M.rgb(x, y, k) - rgb_in[k]) * M.g_rgb(x, y, k);
            }
            if (diff_grad <= 0) continue;
            if (nz1) acc0 -= diff_grad / edge_dist(w.p[0][0], w.p[1][0], den1, d1, d1_cross, is_f, P.eps);
            if (nz0) acc1 -= diff_grad / edge_dist(w.p[0][0], w.p[1][0], den0, d1, d1_cross, is_f, P.eps);
        }
    }
    // "in" pass (rasterize.py:658-727): across the face to its opposite edge
    {
        float d0_cross2;
        if ((fd0 - w.p[0][0]) * (fd0 - w.p[2][0]) < 0) {
            d0_cross2 = (w.p[2][1] - w.p[0][1]) / (w.p[2][0] - w.p[0][0]);
            d0_cross2 = d0_cross2 * (fd0 - w.p[0][0]);
            d0_cross2 = d0_cross2 + w.p[0][1];
        } else {
            d0_cross2 = (w.p[1][1] - w.p[2][1]) / (w.p[1][0] - w.p[2][0]);
            d0_cross2 = d0_cross2 * (fd0 - w.p[2][0]);
            d0_cross2 = d0_cross2 + w.p[2][1];
        }
        const int d1_limit = (0 < w.direction) ? cvt_i32(ceilf(d0_cross2)) : cvt_i32(floorf(d0_cross2));
        const int d1_from = max(min(d1_in, d1_limit), 0);
        const int d1_to = min(max(d1_in, d1_limit), S - 1);
        for (int d1 = d1_from + lane; d1 <= d1_to; d1 += nlanes) {
            const int x = axis == 0 ? d0 : d1, y = axis == 0 ? d1 : d0;
            if (M.fidx(x, y) != fn) continue;
            float diff_grad = 0.0f;
            if (use_alpha) diff_grad = diff_grad + (1.0f - alpha_out) * M.g_alpha(x, y);
            if (use_rgb) {
#pragma unroll
                for (int k = 0; k < 3; k++) diff_grad = diff_grad + (M.rgb(x, y, k) - rgb_out[k]) * M.g_rgb(x, y, k);
            }
            if (diff_grad <= 0) continue;
            if (nz1) acc0 -= diff_grad / edge_dist(w.p[0][0], w.p[1][0], den1, d1, d1_cross, is_f, P.eps);
            if (nz0) acc1 -= diff_grad / edge_dist(w.p[0][0], w.p[1][0], den0, d1, d1_cross, is_f, P.eps);
        }
    }
}

__global__ __launch_bounds__(256) void k_mark_visible(const int32_t* __restrict__ face_index_map, long npx, int S,
                                                       int nf, uint32_t* __restrict__ visible)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= npx) return;
    const int fn = face_index_map[i];
    if (fn >= 0) visible[(i / ((long)S * S)) * nf + fn] = 1u;
}

// h(x,y) = max(-g_alpha(x,y), 0) on background pixels, 0 on covered ones; written TRANSPOSED only (r06: the image rows are formed
// by k_compact_rows itself from the face-index map -- the row-major h map was 38 MB written and read back per frame)
// (the same pass over the face-index map sets the visible flags k_mark_visible would: one launch less in the silhouette path)
__global__ __launch_bounds__(256) void k_hmap(const BwdParams P, float* __restrict__ hmapT)
{
    __shared__ float tile[32][33];
    const int S = P.S, b = blockIdx.z;
    const MapReader M(P, b);
    const int x = blockIdx.x * 32 + (threadIdx.x & 31);
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int ly = (threadIdx.x >> 5) + 8 * r, y = blockIdx.y * 32 + ly;
        float h = 0.0f;
        if (x < S && y < S) {
            const int fn = M.fidx(x, y);
            if (fn < 0) {
                const float t = (0.0f - 1.0f) * M.g_alpha(x, y);
                h = t > 0.0f ? t : 0.0f;
            } else {
                P.visible[(size_t)b * P.nf + fn] = 1u;
            }
        }
        tile[ly][threadIdx.x & 31] = h;
    }
    __syncthreads();
    const int yt = blockIdx.y * 32 + (threadIdx.x & 31);
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int lx = (threadIdx.x >> 5) + 8 * r, xt = blockIdx.x * 32 + lx;
        if (xt < S && yt < S) hmapT[((size_t)b * S + xt) * S + yt] = tile[threadIdx.x & 31][lx];
    }
}

// One WAVE per map row: prefix counts of the non-zero entries and their compact (position, value) list.
// (rows [0, rows_per_map) come from `maps`, the next rows_per_map from `mapsT`: both orientations in one launch)
// r06: was one 256-thread workgroup per row with four barriers per 256 positions (the wave totals met in LDS); a wave carries the
// running count in a register and needs neither LDS nor barriers: 34 -> 2x us.  Two positions per lane and round (8-byte loads).
__global__ __launch_bounds__(256) void k_compact_rows(const BwdParams P, const float* __restrict__ mapsT,
                                                      size_t rows_per_map, int S, uint16_t* __restrict__ cnt,
                                                      uint16_t* __restrict__ pos, float* __restrict__ val)
{
    const int lane = threadIdx.x & 63;
    const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= 2 * rows_per_map) return;   // (wave-uniform)
    const bool image_row = row < rows_per_map;   // h from the face-index map and the upstream gradient (k_hmap's expression)
    const float* src = mapsT + (image_row ? 0 : row - rows_per_map) * S;
    const int yb = (int)(row / (size_t)S), yy = (int)(row % (size_t)S);
    const MapReader M(P, image_row ? yb : 0);
    auto h_at = [&](const int x) {
        float h = 0.0f;
        if (M.fidx(x, yy) < 0) {
            const float t = (0.0f - 1.0f) * M.g_alpha(x, yy);
            h = t > 0.0f ? t : 0.0f;
        }
        return h;
    };
    uint16_t* c = cnt + row * (S + 1);
    int base = 0;
    for (int x0 = 0; x0 < S; x0 += 128) {
        const int x = x0 + 2 * lane;
        float h0 = 0.0f, h1 = 0.0f;
        if (image_row) {
            if (x < S) h0 = h_at(x);
            if (x + 1 < S) h1 = h_at(x + 1);
        } else if (x + 1 < S && ((S & 1) == 0)) {   // (rows of an even S are 8-byte aligned)
            const float2 v = *reinterpret_cast<const float2*>(src + x);
            h0 = v.x;
            h1 = v.y;
        } else {
            if (x < S) h0 = src[x];
            if (x + 1 < S) h1 = src[x + 1];
        }
        const bool nz0 = h0 > 0.0f, nz1 = h1 > 0.0f;
        const unsigned long long m0 = __ballot(nz0), m1 = __ballot(nz1);
        const unsigned long long below = (1ull << lane) - 1ull;
        // entries in position order: lane l's two positions come after both positions of every lane below it
        const int i0 = base + __popcll(m0 & below) + __popcll(m1 & below);
        const int i1 = i0 + (nz0 ? 1 : 0);
        if (x < S) {
            c[x] = (uint16_t)i0;
            if (nz0) {
                pos[row * S + i0] = (uint16_t)x;
                val[row * S + i0] = h0;
            }
        }
        if (x + 1 < S) {
            c[x + 1] = (uint16_t)i1;
            if (nz1) {
                pos[row * S + i1] = (uint16_t)(x + 1);
                val[row * S + i1] = h1;
            }
        }
        base += __popcll(m0) + __popcll(m1);
    }
    if (lane == 0) c[S] = (uint16_t)base;
}

// (Measured and dropped, r05: k_hmap + k_compact_rows as ONE pass over the face-index map without the dense h maps -- nothing but
// k_compact_rows reads them.  Image rows are easy (h on the fly, the same compaction); image COLUMNS were taken as bands of 32
// columns, a 256-row chunk parked transposed in LDS, 32 rows of a column per thread (rotated LDS reads, counts written out
// coalesced through a second LDS tile, running base carried over the chunks).  Bit-identical lists, all tests green -- and 161 us
// (first cut: per-thread 2-byte count stores) / 99 us (second cut) against the pair's 66 us: 384 long-running band workgroups with
// three dependent chunk phases each cannot compete with two streaming passes at ~3.5 TB/s that the transposing tile kernel and
// 24 576 row workgroups keep fully parallel.  profiles/r05f_*, r05g_*.)
// (Measured and dropped, r06: the visible faces of a workgroup's run compacted through LDS in front of the heavy path of k_edge_plan and
// k_edge_reduce -- nine of ten faces are hidden, so every wave runs it for ~6 lanes.  Identical results, frame step +30 us on cad_like:
// the per-face work is a chain of dependent loads that dense waves do not shorten, and a 1024-thread workgroup whose work sits in its
// first two waves holds its CU slot until they finish.)
constexpr int PLAN_THREADS = 1024;   // one counter atomic per workgroup: 1340 of them per frame instead of 5359 (~5 ns each)
__global__ __launch_bounds__(PLAN_THREADS) void k_edge_plan(const BwdParams P)
{
    const long i = (long)blockIdx.x * PLAN_THREADS + threadIdx.x;
    const long total = (long)P.bs * P.nf;
    if (P.sink.grad_verts) {   // what k_edge_reduce's atomics add onto (one launch ahead of them on the stream)
        const long n = (long)P.bs * P.sink.nv * 3;
        for (long k = i; k < n; k += (long)gridDim.x * PLAN_THREADS) P.sink.grad_verts[k] = 0.0f;
    }
    const int lane = threadIdx.x & 63;
    int from[6], cnt[6];
    uint32_t nchunks = 0;
    bool contributes = false;
    if (i < total) {
        const bool active = (P.flags & (SDN_ALPHA | SDN_RGB)) != 0;
        if (active && P.visible[i] != 0u) {   // (nine of ten faces are hidden: their coordinates are never loaded)
            float face[9];
#pragma unroll
            for (int k = 0; k < 9; k++) face[k] = P.faces[i * 9 + k];
            if (!is_backface(face)) {
                const float is_f = (float)P.S;
#pragma unroll
                for (int e = 0; e < 6; e++) {
                    const EdgeWalk w = edge_walk(face, e >> 1, e & 1, is_f);
                    from[e] = w.d0_from;
                    cnt[e] = max(w.d0_to - w.d0_from + 1, 0);
                    nchunks += (uint32_t)((cnt[e] + CHUNK - 1) / CHUNK);
                }
                contributes = nchunks != 0;
            }
        }
    }
    // ONE atomic on the (single) chunk counter per WORKGROUP: inclusive prefix sum of the lanes' chunk counts per wave, the
    // wave totals meet in LDS, thread 0 reserves the workgroup's run.  Atomics on one address serialise in L2 at ~5 ns
    // each: one per wave (what the compiler's atomic optimizer makes of a per-face atomicAdd) was 21k per frame = the
    // kernel's whole 100 us.
    __shared__ uint32_t wave_sum[PLAN_THREADS / 64];
    __shared__ uint32_t block_base;
    uint32_t incl = nchunks;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    const int wave = threadIdx.x >> 6;
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int w = 0; w < PLAN_THREADS / 64; w++) tot += wave_sum[w];
        block_base = tot ? atomicAdd(P.counter, tot) : 0u;
    }
    __syncthreads();
    uint32_t wave_base = block_base;
    for (int w = 0; w < wave; w++) wave_base += wave_sum[w];
    if (i >= total) return;
    if (!contributes) {
        P.chunk_base[i] = -2;
        return;
    }
    const uint32_t base = wave_base + incl - nchunks;
    if (base + nchunks > P.cap) {
        P.chunk_base[i] = -1;  // no room: k_edge_reduce walks this face serially
        // the slots it reserved below the cap must not look like valid work to the scan kernels (face id out of range)
        for (uint32_t c = base; c < P.cap; c++) P.chunk_desc[c] = make_uint4(0xffffffffu, 0xffffffffu, 0u, 0u);
        return;
    }
    P.chunk_base[i] = (int32_t)base;
    uint32_t c = base;
#pragma unroll
    for (int e = 0; e < 6; e++) {
        for (int s = 0; s < cnt[e]; s += CHUNK)
            P.chunk_desc[c++] = make_uint4((uint32_t)i, (uint32_t)e, (uint32_t)(from[e] + s),
                                           (uint32_t)min(CHUNK, cnt[e] - s));
    }
}

__global__ __launch_bounds__(256) void k_edge_scan(const BwdParams P)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * 256u) >> 6;
    const uint32_t nchunks = min(*P.counter, P.cap);
    const bool use_alpha = (P.flags & SDN_ALPHA) != 0;
    const bool use_rgb = (P.flags & SDN_RGB) != 0;
    const float is_f = (float)P.S;
    for (uint32_t c = wave; c < nchunks; c += nwaves) {
        const uint4 d = P.chunk_desc[c];
        if (d.x >= (uint32_t)((long)P.bs * P.nf)) {  // slot reserved by a face that then fell back to serial
            if (lane == 0) P.chunk_out[c] = make_float2(0.f, 0.f);
            continue;
        }
        const long i = d.x;
        const int bn = (int)(i / P.nf), fn = (int)(i % P.nf);
        float face[9];
#pragma unroll
        for (int k = 0; k < 9; k++) face[k] = P.faces[i * 9 + k];
        const int axis = (int)(d.y & 1u);
        const EdgeWalk w = edge_walk(face, (int)(d.y >> 1), axis, is_f);
        const MapReader M(P, bn);
        float acc0 = 0.f, acc1 = 0.f;
        for (int s = 0; s < (int)d.w; s++)
            edge_pixel(P, M, w, fn, axis, (int)d.z + s, use_alpha, use_rgb, lane, 64, acc0, acc1);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            acc0 += __shfl_xor(acc0, o, 64);
            acc1 += __shfl_xor(acc1, o, 64);
        }
        if (lane == 0) P.chunk_out[c] = make_float2(acc0, acc1);
    }
}

// Silhouette-only scan (the training / test-time-optimisation case: only masks are differentiated,
// geometric/scripts/main.py:142-151,447): 64 EDGE PIXELS per wave instead of one.
//   phase A  lane = one edge pixel of one of 8 chunks: walk geometry, the pixel just inside / outside the edge, the
//            whole "in" pass (a few pixels across the face) and the range [k0, k1) of the row's non-zero list that the
//            "out" pass has to visit -- all lane-private, nothing is redundantly computed wave-wide;
//   phase B  the 64 lanes share the "out" terms: for every lane that owns a non-empty range the owner's constants are
//            broadcast (v_readlane -> scalar registers) and the lanes stride over its range; partial sums are kept per
//            chunk and butterfly-reduced once per chunk.
// Terms and their values are exactly those of edge_pixel(); only the association of the float sums differs.
#ifndef SDN_LAB_IN_LANE_MAX
#define SDN_LAB_IN_LANE_MAX 32   // (4 ... 32 measured: within 3 % of each other, 32 best on the reference's own templates)
#endif
constexpr int IN_LANE_MAX = SDN_LAB_IN_LANE_MAX;   // "in" walks of more pixels than this are shared by the wave ...
#ifndef SDN_LAB_COOP_MAX_WALKS
#define SDN_LAB_COOP_MAX_WALKS 32   // (edge kernels, us, cad_like / six templates / 7905d83a / 3776e4d1: 2-8: 184 / 242 / 294 / 237, 16: 182 / 187 / 287 / 217, 24-32: 180 / 188 / 293 / 170, 63: 186 / 187 / 332 / 169, always: 187 / 190 / 480 / 172)
#endif
constexpr int COOP_MAX_WALKS = SDN_LAB_COOP_MAX_WALKS;   // ... when at most this many lanes of the wave hold one

__global__ __launch_bounds__(256) void k_edge_scan_sil(const BwdParams P)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * 256u) >> 6;
    const uint32_t nchunks = min(*P.counter, P.cap);
    const int S = P.S;
    const float is_f = (float)S;
    const float two_over_is = 2.0f / is_f, eps_f = (float)P.eps;
    const long total_faces = (long)P.bs * P.nf;
    // (A software pipeline over the wave's iterations -- record of iteration k + 2 and face of k + 1 requested at the top of k --
    // was measured and dropped: 72 registers instead of 54, 138 -> 154 us.)
    constexpr uint32_t CPW = 64u / CHUNK;   // chunks per wave and round
    for (uint32_t c0 = wave * CPW; c0 < nchunks; c0 += nwaves * CPW) {
        const uint32_t c = c0 + (uint32_t)(lane / CHUNK);
        const int s_in = lane % CHUNK;
        // ---------------- phase A
        float in0 = 0.f, in1 = 0.f;  // this edge pixel's "in" pass
        int k0 = 0, k1 = 0;           // its "out" range in the row's non-zero list
        bool up = false;              // the scan runs towards larger positions
        size_t row = 0;
        float pa = 0.f, pb = 0.f, den1 = 1.f, den0 = 1.f, d1_cross = 0.f;
        int nzflags = 0;
        bool chunk_ok = false;
        // a long "in" walk is handed to the whole wave after phase A (IN_LANE_MAX pixels stay with the lane)
        int w_from = 0, w_to = -1, w_key = 0, w_d0 = 0;   // w_key: axis | fn << 1;  w_d0: d0 | bn << 16
        float w_ta1 = 0.f, w_ta0 = 0.f;
        if (c < nchunks) {
            const uint4 d = P.chunk_desc[c];
            chunk_ok = d.x < (uint32_t)total_faces;
            if (chunk_ok && s_in < (int)d.w) {
                const long i = d.x;
                const int bn = (int)(i / P.nf), fn = (int)(i % P.nf);
                float face[9];
#pragma unroll
                for (int k = 0; k < 9; k++) face[k] = P.faces[i * 9 + k];
                const int axis = (int)(d.y & 1u);
                const EdgeWalk w = edge_walk(face, (int)(d.y >> 1), axis, is_f);
                const MapReader M(P, bn);
                const int d0 = (int)d.z + s_in;
                const float fd0 = (float)d0;
                d1_cross = (w.p[1][1] - w.p[0][1]) / (w.p[1][0] - w.p[0][0]);
                d1_cross = d1_cross * (fd0 - w.p[0][0]);
                d1_cross = d1_cross + w.p[0][1];
                const int d1_in = (0 < w.direction) ? cvt_i32(floorf(d1_cross)) : cvt_i32(ceilf(d1_cross));
                const int d1_out = d1_in + w.direction;
                if (!(d1_in < 0 || S <= d1_in || d1_out < 0 || S <= d1_out)) {
                    const int xin = axis == 0 ? d0 : d1_in, yin = axis == 0 ? d1_in : d0;
                    const int xout = axis == 0 ? d0 : d1_out, yout = axis == 0 ? d1_out : d0;
                    const int f_in = M.fidx(xin, yin);
                    const float alpha_out = M.alpha(xout, yout);
                    const bool nz1 = w.p[1][0] != fd0, nz0 = w.p[0][0] != fd0;
                    pa = w.p[0][0];
                    pb = w.p[1][0];
                    den1 = w.p[1][0] - fd0;
                    den0 = fd0 - w.p[0][0];
                    nzflags = (nz1 ? 1 : 0) | (nz0 ? 2 : 0);
                    if (f_in == fn) {  // "out" pass range (rasterize.py:600-656)
                        const int d1_limit = (0 < w.direction) ? S - 1 : 0;
                        const int d1_from = max(min(d1_out, d1_limit), 0);
                        const int d1_to = min(max(d1_out, d1_limit), S - 1);
                        row = (size_t)(axis == 0 ? P.bs : 0) * S + (size_t)bn * S + d0;
                        const uint16_t* cnt = P.nz_cnt + row * (S + 1);
                        // the scan runs to the image border: one end of [k0, k1) is the start of the row's list or its
                        // end (cnt[0] = 0, cnt[S] = the row's total): ONE scattered 2-byte load per owner besides the total
                        up = 0 < w.direction;
                        if (up) {
                            k0 = cnt[d1_from];
                            k1 = cnt[S];
                        } else {
                            k0 = 0;
                            k1 = cnt[d1_to + 1];
                        }
                    }
                    // "in" pass (rasterize.py:658-727), lane-serial: a few pixels across the face.  Its terms are
                    // diff = (1 - alpha_out) * g: with the pixel just outside the edge covered (alpha_out = 1) every one of
                    // them is dropped by `if (diff_grad <= 0) continue` (rasterize.py:715), so only edge pixels on the
                    // SILHOUETTE walk at all -- a few per cent of them; the rest used to load 8 map values per round for
                    // nothing (the kernel is bound by the rate of scattered 4-byte loads, one cache line per lane)
#ifdef SDN_LAB_SCAN_NOIN     // (lab: what the "in" walks cost)
                    if (false) {
#else
                    if (alpha_out == 0.0f) {
#endif
                    float d0_cross2;
                    if ((fd0 - w.p[0][0]) * (fd0 - w.p[2][0]) < 0) {
                        d0_cross2 = (w.p[2][1] - w.p[0][1]) / (w.p[2][0] - w.p[0][0]);
                        d0_cross2 = d0_cross2 * (fd0 - w.p[0][0]);
                        d0_cross2 = d0_cross2 + w.p[0][1];
                    } else {
                        d0_cross2 = (w.p[1][1] - w.p[2][1]) / (w.p[1][0] - w.p[2][0]);
                        d0_cross2 = d0_cross2 * (fd0 - w.p[2][0]);
                        d0_cross2 = d0_cross2 + w.p[2][1];
                    }
                    const int d1_limit = (0 < w.direction) ? cvt_i32(ceilf(d0_cross2)) : cvt_i32(floorf(d0_cross2));
                    const int d1_from = max(min(d1_in, d1_limit), 0);
                    const int d1_to = min(max(d1_in, d1_limit), S - 1);
                    const float ta1 = (pb - pa) / den1 * two_over_is, ta0 = (pb - pa) / den0 * two_over_is;
                    // (the walk itself runs after phase A, by this lane or by the whole wave: see there)
                    w_from = d1_from;
                    w_to = d1_to;
                    w_key = axis | (fn << 1);
                    w_d0 = d0 | (bn << 16);
                    w_ta1 = ta1;
                    w_ta0 = ta0;
                    }  // alpha_out == 0
                }
            }
        }
        // ---------------- the "in" walks.  Most are a few pixels long and stay with their lane.  A face many pixels across (CAD panels:
        // hundreds) made its lane the wave's tail -- 52 of the kernel's 117 us on cad_like (lab build without the walks,
        // tools/lab/scan_parts.sh): when FEW lanes of the wave hold a walk longer than IN_LANE_MAX those walks go to the whole wave,
        // one at a time, the 64 lanes striding over the pixels (same terms; the sum is a tree then).  When MANY do (the edge pixels of
        // one giant face fill the wave: template 7905d83a) 64 walks side by side beat 64 broadcast passes: 254 vs 400 us.
        {
            const bool has = w_to >= w_from;
            unsigned long long lm = __ballot(has && w_to - w_from >= IN_LANE_MAX);
            const bool coop = __popcll(lm) <= COOP_MAX_WALKS;
            if (!coop) lm = 0ull;
            const bool own = has && (coop ? w_to - w_from < IN_LANE_MAX : true);
            while (lm) {
                const int j = __builtin_amdgcn_readfirstlane(__ffsll((long long)lm) - 1);
                lm &= lm - 1ull;
                const int from = __builtin_amdgcn_readlane(w_from, j), to = __builtin_amdgcn_readlane(w_to, j);
                const int key = __builtin_amdgcn_readlane(w_key, j), d0b = __builtin_amdgcn_readlane(w_d0, j);
                const float ta1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w_ta1), j));
                const float ta0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w_ta0), j));
                const float cross = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d1_cross), j));
                const int nzj = __builtin_amdgcn_readlane(nzflags, j);
                const int axis = key & 1, fn = key >> 1, d0 = d0b & 0xffff;
                const MapReader M(P, d0b >> 16);
                float p0 = 0.f, p1 = 0.f;
                for (int d1 = from + lane; d1 <= to; d1 += 64) {
                    const int x = axis == 0 ? d0 : d1, y = axis == 0 ? d1 : d0;
                    if (M.fidx(x, y) != fn) continue;
                    float diff_grad = 0.0f;
                    diff_grad = diff_grad + (1.0f - 0.0f) * M.g_alpha(x, y);   // (alpha_out = 0: the condition of every walk)
                    if (diff_grad <= 0) continue;
                    const float dd = (float)d1 - cross;
                    if (nzj & 1) {
                        const float dist = ta1 * dd;
                        p0 -= diff_grad * __builtin_amdgcn_rcpf(0.0f < dist ? dist + eps_f : dist - eps_f);
                    }
                    if (nzj & 2) {
                        const float dist = ta0 * dd;
                        p1 -= diff_grad * __builtin_amdgcn_rcpf(0.0f < dist ? dist + eps_f : dist - eps_f);
                    }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    p0 += __shfl_xor(p0, o, 64);
                    p1 += __shfl_xor(p1, o, 64);
                }
                if (lane == j) {
                    in0 += p0;
                    in1 += p1;
                }
            }
            if (own) {
                // four pixels per round: their eight loads (owner index, upstream gradient) are issued together -- the walk was
                // bound by the latency of one dependent load pair per pixel
                const int axis = w_key & 1, fn = w_key >> 1, d0 = w_d0 & 0xffff;
                const MapReader M(P, w_d0 >> 16);
                const bool nz1 = (nzflags & 1) != 0, nz0 = (nzflags & 2) != 0;
                for (int d1b = w_from; d1b <= w_to; d1b += 4) {
                    int fi[4];
                    float ga[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int d1 = min(d1b + u, w_to);
                        const int x = axis == 0 ? d0 : d1, y = axis == 0 ? d1 : d0;
                        fi[u] = M.fidx(x, y);
                        ga[u] = M.g_alpha(x, y);
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int d1 = d1b + u;
                        if (d1 > w_to || fi[u] != fn) continue;
                        float diff_grad = 0.0f;
                        diff_grad = diff_grad + (1.0f - 0.0f) * ga[u];   // (alpha_out = 0)
                        if (diff_grad <= 0) continue;
                        // same evaluation as the "out" terms of k_edge_rows (hoisted factor, v_rcp_f32)
                        const float dd = (float)d1 - d1_cross;
                        if (nz1) {
                            const float dist = w_ta1 * dd;
                            in0 -= diff_grad * __builtin_amdgcn_rcpf(0.0f < dist ? dist + eps_f : dist - eps_f);
                        }
                        if (nz0) {
                            const float dist = w_ta0 * dd;
                            in1 -= diff_grad * __builtin_amdgcn_rcpf(0.0f < dist ? dist + eps_f : dist - eps_f);
                        }
                    }
                }
            }
        }
        // ---------------- the "out" scan is filed under its row; k_edge_rows evaluates all scans of a row together
#ifdef SDN_LAB_SCAN_NOFILE   // (lab: what the filing of the owners costs -- draws wrong gradients)
        const bool owner = false;
#else
        const bool owner = k1 > k0;
#endif
        if (owner) {
            // scans towards the far border ([k0, row total)) are filed from the front of the row's region, scans towards position 0
            // ([0, k1)) from its back: the lanes of a k_edge_rows batch then walk ranges that end (start) together
            uint32_t at = atomicAdd(P.row_cnt + 2 * row + (up ? 0 : 1), 1u);
            if (!up) at = (uint32_t)(3 * S) - 1u - at;
            OwnerRec r;
            r.t1 = (pb - pa) / den1 * two_over_is;   // the owner-constant factor of dist (rasterize.py:646-652), as before
            r.t0 = (pb - pa) / den0 * two_over_is;
            r.cross = d1_cross;
            r.kk = (uint32_t)k0 | ((uint32_t)k1 << 16);
            r.slot = (c * (uint32_t)CHUNK + (uint32_t)s_in) | ((uint32_t)nzflags << 30);
            if (at < (uint32_t)(3 * S)) P.own_rec[row * (size_t)(3 * S) + at] = r;   // (always: <= 3 owners per pixel of the row)
        }
        const unsigned long long owners = __ballot(owner);
        // add the "in" sums of each chunk (lanes g*CHUNK .. g*CHUNK + CHUNK-1) into lane g*CHUNK
        float sum0 = in0, sum1 = in1;
#pragma unroll
        for (int o = CHUNK / 2; o > 0; o >>= 1) {
            const float t0 = __shfl_down(sum0, o, 64), t1 = __shfl_down(sum1, o, 64);
            if (s_in + o < CHUNK) {
                sum0 += t0;
                sum1 += t1;
            }
        }
        if (s_in == 0 && c < nchunks) {
            P.chunk_out[c] = chunk_ok ? make_float2(sum0, sum1) : make_float2(0.f, 0.f);
            P.chunk_mask[c] = chunk_ok ? (uint8_t)((owners >> (lane & ~(CHUNK - 1))) & ((1ull << CHUNK) - 1ull)) : (uint8_t)0;
        }
    }
}

// All "out" scans that run along one row / column of one object: one workgroup per row.  The row's non-zero list (<= S
// entries of (position, value)) is staged in LDS once; then every lane takes one scan (owner) and walks its range [k0, k1) of
// the list out of LDS -- lane-private sums, no cross-lane traffic, no re-read of the list per owner (the former cooperative
// scan re-fetched it from L2 for each of the ~130 owners of a row: 840 MB of traffic for 240 MB of maps).  The batches of 64
// owners go round-robin to the four waves (a row has up to ~300 owners and 600 list entries: one wave walking them all was the
// kernel's tail), and the walk is unrolled four entries deep so that the LDS reads of a lane overlap.
// Terms as in edge_pixel(): val / (t * (pos - cross) +- eps), the divide as v_rcp_f32; summed in list order, four interleaved
// partial sums per owner.
__global__ __launch_bounds__(256) void k_edge_rows(const BwdParams P, int nrows)
{
    extern __shared__ float2 lds_list[];   // [S]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int S = P.S;
    const int row = blockIdx.x;
    const float eps_f = (float)P.eps;
    const uint32_t n_up = min(P.row_cnt[2 * row], (uint32_t)(3 * S));
    const uint32_t n_down = min(P.row_cnt[2 * row + 1], (uint32_t)(3 * S) - n_up);
    if (n_up + n_down == 0) return;   // (uniform over the workgroup)
    const int total = P.nz_cnt[(size_t)row * (S + 1) + S];
    for (int k = threadIdx.x; k < total; k += 256)
        lds_list[k] = make_float2((float)P.nz_pos[(size_t)row * S + k], P.nz_val[(size_t)row * S + k]);
    __syncthreads();
    const OwnerRec* recs = P.own_rec + (size_t)row * (3 * S);
    const uint32_t b_up = (n_up + 63) >> 6, b_all = b_up + ((n_down + 63) >> 6);
    for (uint32_t b = wave; b < b_all; b += 4) {
        const bool down = b >= b_up;
        const uint32_t idx = (down ? b - b_up : b) * 64u + lane;
        if (idx >= (down ? n_down : n_up)) continue;
        const OwnerRec r = recs[down ? (uint32_t)(3 * S) - 1u - idx : idx];
        const int k0 = (int)(r.kk & 0xffffu), k1 = (int)(r.kk >> 16);
        const bool nz1 = (r.slot >> 30) & 1u, nz0 = (r.slot >> 31) & 1u;
        // the two sums of an owner (one per end point of its edge) as the halves of one packed register: the compiler turns the products
        // and sums into v_pk_mul_f32 / v_pk_add_f32, which leaves the two compare / select pairs and the two v_rcp_f32 per entry (r06:
        // 86 -> 6x us; the same operations on the same values as before, so the same bits).  A sum whose end point lies on the pixel column
        // (nz flag clear: t = +-inf or NaN) is computed and thrown away below instead of being masked entry by entry.
        typedef float v2f __attribute__((ext_vector_type(2)));
        const v2f t = {r.t1, r.t0};
        // +- eps (rasterize.py:647, 652: `0 < dist ? dist + eps : dist - eps`) does not depend on the entry: an owner filed from the front
        // scans towards larger positions, all of them beyond the edge crossing (pos >= floor(cross) + 1: pos - cross > 0 strictly), one
        // from the back towards smaller ones (pos <= ceil(cross) - 1); t * (pos - cross) cannot underflow (|t| >= 1e-10, |pos - cross| >=
        // 1e-5), so its sign is the sign of t times the direction -- and t = 0 or NaN gives `0 < dist` false either way.  The same values
        // as the per-entry test, two compares and two selects per entry less.
        const float sdd = down ? -1.0f : 1.0f;
        const v2f es = {0.0f < t.x * sdd ? eps_f : -eps_f, 0.0f < t.y * sdd ? eps_f : -eps_f};
        v2f o[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
        auto term = [&](const float2 e, v2f& a) {
            // the value sits in the HIGH half of the (position, value) pair the LDS read returns, and the compiler would multiply with
            // `v_pk_mul_f32 ... op_sel:[1,0]` (low result lane reading the high half): the operand form that returned wrong low
            // results in conv_narrow.hip beside an MFMA wave on the same SIMD (see there).  An opaque copy makes it a low-half splat.
            float val = e.y;
#ifndef SDN_LAB_ROWS_HI_SPLAT   // (lab build: what the compiler does on its own, for tools/lab/pk_race.py)
            asm volatile("" : "+v"(val));
#endif
            const float dd = e.x - r.cross;
            v2f dist = t * dd;
#ifdef SDN_LAB_ROWS_EPS_PER_ENTRY   // (lab: the per-entry test, for a bit-for-bit comparison of the gradients)
            const v2f s = {0.0f < dist.x ? eps_f : -eps_f, 0.0f < dist.y ? eps_f : -eps_f};
            dist = dist + s;
#else
            dist = dist + es;
#endif
            const v2f rc = {__builtin_amdgcn_rcpf(dist.x), __builtin_amdgcn_rcpf(dist.y)};
            a = a - val * rc;
        };
        int k = k0;
        for (; k + 4 <= k1; k += 4) {
            const float2 e0 = lds_list[k], e1 = lds_list[k + 1], e2 = lds_list[k + 2], e3 = lds_list[k + 3];
            term(e0, o[0]);
            term(e1, o[1]);
            term(e2, o[2]);
            term(e3, o[3]);
        }
        if (k < k1) term(lds_list[k], o[0]);
        if (k + 1 < k1) term(lds_list[k + 1], o[1]);
        if (k + 2 < k1) term(lds_list[k + 2], o[2]);
        const v2f sum = (o[0] + o[1]) + (o[2] + o[3]);
        P.own_out[r.slot & 0x3fffffffu] = make_float2(nz1 ? sum.x : 0.0f, nz0 ? sum.y : 0.0f);
    }
}

// chunk_out[c] += the "out" scans of chunk c's edge pixels, in lane order: one thread per chunk, its 8 result slots are one
// 64-byte line (k_edge_reduce, one thread per FACE, would chase them one dependent load at a time: 32 -> 123 us).
__global__ __launch_bounds__(256) void k_chunk_sum(const BwdParams P)
{
    const uint32_t nchunks = min(*P.counter, P.cap);
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= nchunks) return;
    uint32_t m = P.chunk_mask[c];
    if (!m) return;
    float2 a = P.chunk_out[c];
    const float4* q = reinterpret_cast<const float4*>(P.own_out + (size_t)c * CHUNK);
#pragma unroll
    for (int h = 0; h < CHUNK / 2; h++) {
        if (!(m & (3u << (2 * h)))) continue;
        const float4 v = q[h];
        if (m & (1u << (2 * h))) {
            a.x += v.x;
            a.y += v.y;
        }
        if (m & (2u << (2 * h))) {
            a.x += v.z;
            a.y += v.w;
        }
    }
    P.chunk_out[c] = a;
}

__global__ __launch_bounds__(256) void k_edge_reduce(const BwdParams P)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)P.bs * P.nf;
    if (i >= total) return;
    const bool accumulate = (P.flags & SDN_ACCUMULATE) != 0;
    // SDN_SPARSE_GRAD: a face that owns no pixel has a zero gradient from every term (K5's scans start at pixels the face
    // covers, K6 / K7 run over its pixels); its row is left unwritten and the consumer skips it by the same flag
    if ((P.flags & SDN_SPARSE_GRAD) && P.visible[i] == 0u) return;
    const int32_t base = P.chunk_base[i];
    float grad_face[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (base != -2) {
        float face[9];
#pragma unroll
        for (int k = 0; k < 9; k++) face[k] = P.faces[i * 9 + k];
        const float is_f = (float)P.S;
        const bool use_alpha = (P.flags & SDN_ALPHA) != 0;
        const bool use_rgb = (P.flags & SDN_RGB) != 0;
        const int bn = (int)(i / P.nf), fn = (int)(i % P.nf);
        uint32_t c = (uint32_t)base;
        for (int e = 0; e < 6; e++) {
            const int axis = e & 1;
            const EdgeWalk w = edge_walk(face, e >> 1, axis, is_f);
            float a0 = 0.f, a1 = 0.f;
            if (base >= 0) {
                const int cnt = max(w.d0_to - w.d0_from + 1, 0);
                // eight loads in flight, added in chunk order (r04: one dependent load per iteration made the launch as long
                // as its longest face -- ~60 chunks x an L2 round trip = the kernel's 40 us)
                const int nch = (cnt + CHUNK - 1) / CHUNK;
                for (int s = 0; s < nch; s += 8) {
                    float2 o[8];
#pragma unroll
                    for (int u = 0; u < 8; u++)
                        if (s + u < nch) o[u] = P.chunk_out[c + s + u];
#pragma unroll
                    for (int u = 0; u < 8; u++)
                        if (s + u < nch) {
                            a0 += o[u].x;
                            a1 += o[u].y;
                        }
                }
                c += (uint32_t)nch;
                grad_face[w.pi0 * 3 + (1 - axis)] += a0;
                grad_face[w.pi1 * 3 + (1 - axis)] += a1;
            } else {
                // literal reference order: every term is subtracted straight from grad_face (rasterize.py:648,653)
                const MapReader M(P, bn);
                for (int d0 = w.d0_from; d0 <= w.d0_to; d0++)
                    edge_pixel(P, M, w, fn, axis, d0, use_alpha, use_rgb, 0, 1, grad_face[w.pi0 * 3 + (1 - axis)],
                               grad_face[w.pi1 * 3 + (1 - axis)]);
            }
        }
    }
    if (P.sink.grad_verts) {
        if (base == -2) return;   // (hidden, back-facing or without an edge pixel: every term is zero)
        bool any = false;
#pragma unroll
        for (int k = 0; k < 9; k++) any = any || grad_face[k] != 0.0f;
        if (!any) return;
        const int bn = (int)(i / P.nf), fn = (int)(i % P.nf);
        const bool twin = P.sink.fill_back && fn >= P.sink.nf0;
        const int32_t* idx = P.sink.faces_idx + (size_t)bn * P.sink.fstride + (size_t)(twin ? fn - P.sink.nf0 : fn) * 3;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float* dst = P.sink.grad_verts + ((size_t)bn * P.sink.nv + idx[twin ? 2 - k : k]) * 3;
#pragma unroll
            for (int d = 0; d < 3; d++)
                if (grad_face[3 * k + d] != 0.0f) unsafeAtomicAdd(&dst[d], grad_face[3 * k + d]);
        }
        return;
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

static size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

static void bwd_layout(int bs, int nf, int S, size_t off[14], uint32_t& cap, size_t& total)
{
    const size_t n = (size_t)bs * nf;
    cap = (uint32_t)(4 * n + 65536);
    off[0] = 0;                                         // counter (256 B slot) + visible + row_cnt: zeroed by ONE memset
    off[1] = 256;                                       // visible  u32[n]
    off[10] = off[1] + align256(n * sizeof(uint32_t));  // row_cnt  u32[2*bs*S, 2]
    off[2] = off[10] + align256((size_t)4 * bs * S * sizeof(uint32_t));  // chunk_base i32[n]
    off[3] = off[2] + align256(n * sizeof(int32_t));    // chunk_desc uint4[cap]
    off[4] = off[3] + align256((size_t)cap * sizeof(uint4));  // chunk_out float2[cap]
    off[5] = off[4] + align256((size_t)cap * sizeof(float2));          // (hmap: nothing stored)
    off[6] = off[5];   // hmapT float[bs*S*S]  (r06: the row-major h map is gone; off[5] only names the fast path's flag pointer)
    off[7] = off[6] + align256((size_t)bs * S * S * sizeof(float));                 // nz_cnt u16[2*bs*S*(S+1)]
    off[8] = off[7] + align256((size_t)2 * bs * S * (S + 1) * sizeof(uint16_t));    // nz_pos u16[2*bs*S*S]
    off[9] = off[8] + align256((size_t)2 * bs * S * S * sizeof(uint16_t));          // nz_val f32[2*bs*S*S]
    off[11] = off[9] + align256((size_t)2 * bs * S * S * sizeof(float));            // own_rec OwnerRec[2*bs*S * 3*S]
    off[12] = off[11] + align256((size_t)2 * bs * S * 3 * S * sizeof(OwnerRec));  // own_out float2[cap*8]
    off[13] = off[12] + align256((size_t)cap * 8 * sizeof(float2));         // chunk_mask u8[cap]
    total = off[13] + align256((size_t)cap);
}

namespace sdn {
const uint32_t* raster_bwd_visible_flags(const void* workspace) { return (const uint32_t*)((const char*)workspace + 256); }
}  // namespace sdn

SDN_API int sdn_raster_bwd_workspace_bytes(int bs, int nf, int S, size_t* out)
{
    if (bs <= 0 || nf <= 0 || S <= 0 || !out) return fail(SDN_EINVAL, "sdn_raster_bwd_workspace_bytes: bad sizes");
    size_t off[14], total;
    uint32_t cap;
    bwd_layout(bs, nf, S, off, cap, total);
    *out = total;
    return SDN_OK;
}

SDN_API int sdn_rasterize_bwd(const float* faces, const float* textures, int ts, int bs, int nf, int S, double eps,
                              int flags, const float* face_inv, const int32_t* face_index_map,
                              const float* weight_map, const float* depth_map, const float* rgb_map,
                              const float* g_rgb_out, const float* g_alpha_out, const float* g_depth_out,
                              float* grad_faces, float* grad_textures, void* workspace, size_t workspace_bytes,
                              sdnStream stream)
{
    return rasterize_bwd_core(nullptr, faces, textures, ts, bs, nf, S, eps, flags, face_inv, face_index_map, weight_map, depth_map,
                              rgb_map, g_rgb_out, g_alpha_out, g_depth_out, grad_faces, grad_textures, workspace, workspace_bytes,
                              stream);
}

int sdn::rasterize_bwd_core(const VertexSink* sink, const float* faces, const float* textures, int ts, int bs, int nf, int S,
                            double eps, int flags, const float* face_inv, const int32_t* face_index_map, const float* weight_map,
                            const float* depth_map, const float* rgb_map, const float* g_rgb_out, const float* g_alpha_out,
                            const float* g_depth_out, float* grad_faces, float* grad_textures, void* workspace,
                            size_t workspace_bytes, sdnStream stream)
{
    if (!faces || !face_inv || !face_index_map || !weight_map || !depth_map || (!grad_faces && !sink) || bs <= 0 || nf <= 0 ||
        S <= 0)
        return fail(SDN_EINVAL, "sdn_rasterize_bwd: missing state (was the forward run with SDN_SAVE_MAPS?)");
    if (sink && ((flags & (SDN_RGB | SDN_DEPTH | SDN_ACCUMULATE)) || !(flags & SDN_ALPHA) || !g_alpha_out || !sink->faces_idx ||
                 !sink->grad_verts || nf != (sink->fill_back ? 2 : 1) * sink->nf0))
        return fail(SDN_EINVAL, "rasterize_bwd_core: the vertex sink takes a silhouette-only pass");
    if ((flags & SDN_RGB) && (!rgb_map || !textures))
        return fail(SDN_EINVAL, "sdn_rasterize_bwd: rgb gradients need rgb_map and textures");
    if ((flags & SDN_AA) && (S & 1)) return fail(SDN_EINVAL, "sdn_rasterize_bwd: SDN_AA needs an even internal size");
    size_t off[14], need;
    uint32_t cap;
    bwd_layout(bs, nf, S, off, cap, need);
    if (!workspace || workspace_bytes < need)
        return fail(SDN_ENOMEM, "sdn_rasterize_bwd: workspace %zu < %zu bytes", workspace_bytes, need);
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
    char* ws = (char*)workspace;
    P.counter = (uint32_t*)(ws + off[0]);
    P.visible = (uint32_t*)(ws + off[1]);
    P.chunk_base = (int32_t*)(ws + off[2]);
    P.chunk_desc = (uint4*)(ws + off[3]);
    P.chunk_out = (float2*)(ws + off[4]);
    P.hmap = nullptr;
    P.hmapT = nullptr;
    P.nz_cnt = nullptr;
    P.nz_pos = nullptr;
    P.nz_val = nullptr;
    P.row_cnt = nullptr;
    P.own_rec = nullptr;
    P.own_out = nullptr;
    P.chunk_mask = nullptr;
    P.cap = (flags & SDN_SERIAL_EDGES) ? 0u : cap;
    P.eps = eps;
    P.ts = ts;
    P.bs = bs;
    P.nf = nf;
    P.S = S;
    P.flags = flags;
    P.sink = sink ? *sink : VertexSink{nullptr, 0, 0, 0, 0, nullptr};
    // the reference zero-fills a missing upstream gradient (rasterize.py:855-875): a NULL g_* disables that term,
    // but the edge pass still owns the store of grad_faces
    if (!P.g_rgb_out) P.flags &= ~SDN_RGB;
    if (!P.g_alpha_out) P.flags &= ~SDN_ALPHA;
    const long total = (long)bs * nf;
    const long npx = (long)bs * S * S;
    const bool edges = (P.flags & (SDN_ALPHA | SDN_RGB)) != 0;
    if (!edges) P.flags &= ~SDN_SPARSE_GRAD;   // (the visible flags are only built for the edge terms)
    int rc;
    if (edges) {
        // counter + visible flags are contiguous at the head of the workspace
        hipError_t e = hipMemsetAsync(ws, 0, off[2], st);
        if (e != hipSuccess) return fail(SDN_ELAUNCH, "hipMemsetAsync(plan): %s", hipGetErrorString(e));
        // (every reserved chunk slot below the cap is written by k_edge_plan, valid or marked invalid: no 88 MB memset)
        const bool sil_path = (P.flags & SDN_ALPHA) && !(P.flags & SDN_RGB) && !(flags & SDN_SERIAL_EDGES);
        if (!sil_path) {   // (k_hmap sets the flags on its way over the face-index map)
            hipLaunchKernelGGL(k_mark_visible, dim3(cdiv(npx, 256)), dim3(256), 0, st, face_index_map, npx, S, nf,
                               P.visible);
            if ((rc = check_launch("k_mark_visible"))) return rc;
        }
        if (sil_path) {
            float* hmap = (float*)(ws + off[5]);
            float* hmapT = (float*)(ws + off[6]);
            if (S > 65535) return fail(SDN_EINVAL, "sdn_rasterize_bwd: internal size %d exceeds the 16-bit row index", S);
            uint16_t* cnt = (uint16_t*)(ws + off[7]);
            uint16_t* pos = (uint16_t*)(ws + off[8]);
            float* val = (float*)(ws + off[9]);
            const size_t rows = (size_t)bs * S;
            hipLaunchKernelGGL(k_hmap, dim3(cdiv(S, 32), cdiv(S, 32), bs), dim3(256), 0, st, P, hmapT);
            if ((rc = check_launch("k_hmap"))) return rc;
            hipLaunchKernelGGL(k_compact_rows, dim3((unsigned)cdiv((long)(2 * rows), 4)), dim3(256), 0, st, P, hmapT, rows, S, cnt, pos,
                               val);
            if ((rc = check_launch("k_compact_rows"))) return rc;
            P.hmap = hmap;
            P.hmapT = hmapT;
            P.nz_cnt = cnt;
            P.nz_pos = pos;
            P.nz_val = val;
            if ((size_t)cap * 8 >= (1u << 30)) return fail(SDN_EINVAL, "sdn_rasterize_bwd: %u chunks exceed the 30-bit slot index", cap);
            P.row_cnt = (uint32_t*)(ws + off[10]);
            P.own_rec = (OwnerRec*)(ws + off[11]);
            P.own_out = (float2*)(ws + off[12]);
            P.chunk_mask = (uint8_t*)(ws + off[13]);
        }
    }
    hipLaunchKernelGGL(k_edge_plan, dim3(cdiv(total, PLAN_THREADS)), dim3(PLAN_THREADS), 0, st, P);
    if ((rc = check_launch("k_edge_plan"))) return rc;
    if (edges) {
        {
            TimedLaunch timed(TIME_EDGE_SCAN, st, 0.0);
            if (P.nz_cnt) {
                hipLaunchKernelGGL(k_edge_scan_sil, dim3(256 * 8), dim3(256), 0, st, P);
                const int nrows = 2 * bs * S;
                hipLaunchKernelGGL(k_edge_rows, dim3((unsigned)nrows), dim3(256), (size_t)S * sizeof(float2), st, P, nrows);
                hipLaunchKernelGGL(k_chunk_sum, dim3(cdiv((long)cap, 256)), dim3(256), 0, st, P);
            } else
                hipLaunchKernelGGL(k_edge_scan, dim3(256 * 8), dim3(256), 0, st, P);
        }
        if ((rc = check_launch("k_edge_scan"))) return rc;
    }
    hipLaunchKernelGGL(k_edge_reduce, dim3(cdiv(total, 256)), dim3(256), 0, st, P);
    if ((rc = check_launch("k_edge_reduce"))) return rc;

    const bool need_tex = (P.flags & SDN_RGB) && grad_textures;
    if (need_tex && !(flags & SDN_ACCUMULATE)) {
        const size_t n = (flags & SDN_FACE_COLOR) ? (size_t)total * 3 : (size_t)total * ts * ts * ts * 3;
        hipError_t e = hipMemsetAsync(grad_textures, 0, n * sizeof(float), st);
        if (e != hipSuccess) return fail(SDN_ELAUNCH, "hipMemsetAsync(grad_textures): %s", hipGetErrorString(e));
    }
    if (((P.flags & SDN_DEPTH) && P.g_depth_out) || need_tex) {
        hipLaunchKernelGGL(k_bwd_pixels, dim3(cdiv(npx, 256)), dim3(256), 0, st, P);
        if ((rc = check_launch("k_bwd_pixels"))) return rc;
    }
    return SDN_OK;
}
