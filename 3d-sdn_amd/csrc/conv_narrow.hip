// Weight gradient of the convolutions with a handful of output channels: exact fp32 on the vector ALUs.
//
// Reference: autograd of the head layers of the textural networks -- GlobalGenerator's c7s1-3 (networks.py:236),
// Encoder's c7s1-5 (:306), NLayerDiscriminator's last 4x4 conv to one channel (:437) -- run by loss_G.backward() /
// loss_D.backward(), textural/train.py:88-95.
//
//     dW[r, t, c] = sum_{n, q}  a(rows[n, q, r]) * b(gath[n, q + d_t, c])        r < R <= 8, stride 1
//
// On the matrix cores (conv_wgrad.hip) such a layer pays for a 32-row tile with 1-5 useful rows AND re-gathers and
// re-splits the input once per tap (49 times for a 7x7 kernel); at 384 x 1248 the generator head alone took 3.3 ms.
// Here a workgroup keeps a spatial tile of the gathered operand WITH its halo in LDS (read from HBM once per tile, not
// once per tap), a thread owns the columns (tap, channel) = (tg + TG j, cc), and walks the tile's positions: one LDS
// read of x (consecutive lanes = consecutive channels, conflict free) feeds R FMAs against the broadcast d(out) values.
// Roofline: fp32 VALU (2 * positions * taps * C * R flops), not HBM: the tile + halo is read 2-3 times over all tiles.
#include <cstdlib>
#include "conv_common.h"
#include "sdn_common.h"

namespace sdn {

struct NarrowTaps {
    int n;
    signed char dy[CONV_MAX_TAPS];
    signed char dx[CONV_MAX_TAPS];
};

struct NarrowParams {
    const float* rows;  // [N, QH, QW, Cr]   d(out) (conv) -- only channels < R are read
    const float* gath;  // [N, GH, GW, Cc]
    float* dw;          // [Cr, ntaps * Cc] fp32, added to (atomics)
    int N, QH, QW, Cr, GH, GW, Cc;
    int pad_mode, relu_rows, relu_gath;
    int TH, tiles_x, tiles_per_image;  // TH x 32 output positions per tile
    int dy_min, dx_min, HH, HW;        // halo box of a tile: HH x HW input positions, first at (y0 + dy_min, x0 + dx_min)
    unsigned hw_magic;                 // 65536 / HW + 1: pos / HW == (pos * hw_magic) >> 16 for pos < 65536 / HW
    NarrowTaps taps;
};

constexpr int NARROW_TW = 32;

template <int R, int CH, int MAXJ>
__global__ __launch_bounds__(256) void k_wgrad_narrow(const NarrowParams P)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TG = 256 / CH;               // tap groups: thread (tg, cc) owns taps tg, tg + TG, ...
    constexpr int RP = R == 1 ? 1 : (R <= 4 ? 4 : 8);  // d(out) values per position in LDS
    float* xs = lds;                           // [HH][HW][CH]
    float* dzs = lds + P.HH * P.HW * CH;       // [TH * 32][RP]
    const int tid = threadIdx.x, cc = tid % CH, tg = tid / CH;
    const int c0 = blockIdx.y * CH;

    int toff[MAXJ];
#pragma unroll
    for (int j = 0; j < MAXJ; j++) {
        const int t = tg + TG * j;
        // taps behind the last one read a valid slot and accumulate into registers that are never written out
        toff[j] = t < P.taps.n ? ((P.taps.dy[t] - P.dy_min) * P.HW + (P.taps.dx[t] - P.dx_min)) * CH + cc : cc;
    }
    float acc[MAXJ][R];
#pragma unroll
    for (int j = 0; j < MAXJ; j++)
#pragma unroll
        for (int r = 0; r < R; r++) acc[j][r] = 0.f;

    const int total = P.tiles_per_image * P.N;
    const int npos = P.TH * NARROW_TW;
    for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
        const int n = tile / P.tiles_per_image;
        const int ti = tile - n * P.tiles_per_image;
        const int y0 = (ti / P.tiles_x) * P.TH, x0 = (ti % P.tiles_x) * NARROW_TW;
        __syncthreads();  // the previous tile is consumed
        // gathered operand with halo: CH / 4 float4 per position, coalesced over channels; eight loads in flight per
        // thread (a single dependent load -> LDS store chain per iteration left the tile fill latency-bound)
        constexpr int V = CH / 4, PL = 256 / V;  // float4 per position, positions per pass
        const int c4 = tid % V, pw = tid / V;
        const int hpos = P.HH * P.HW;
        for (int pb = 0; pb < hpos; pb += 8 * PL) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int pos = pb + u * PL + pw;
                const int hy = (int)(((unsigned)pos * P.hw_magic) >> 16), hx = pos - hy * P.HW;
                int gy = y0 + hy + P.dy_min, gx = x0 + hx + P.dx_min;
                v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (pos < hpos && resolve_coord(gy, P.GH, P.pad_mode) && resolve_coord(gx, P.GW, P.pad_mode))
                    v[u] = *reinterpret_cast<const f32x4*>(P.gath + (((size_t)n * P.GH + gy) * P.GW + gx) * P.Cc + c0 + 4 * c4);
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int pos = pb + u * PL + pw;
                if (P.relu_gath) {
#pragma unroll
                    for (int e = 0; e < 4; e++) v[u][e] = fmaxf(v[u][e], 0.f);
                }
                if (pos < hpos) *reinterpret_cast<f32x4*>(xs + pos * CH + 4 * c4) = v[u];
            }
        }
        // d(out) of the tile's positions (zero outside the image: those positions contribute nothing)
        for (int idx = tid; idx < npos * RP; idx += 256) {
            const int p = idx / RP, r = idx - p * RP;
            const int qy = y0 + p / NARROW_TW, qx = x0 + p % NARROW_TW;
            float v = 0.f;
            if (r < R && qy < P.QH && qx < P.QW) {
                v = P.rows[(((size_t)n * P.QH + qy) * P.QW + qx) * P.Cr + r];
                if (P.relu_rows) v = fmaxf(v, 0.f);
            }
            dzs[idx] = v;
        }
        __syncthreads();
        // Walk the tile two positions at a time, software-pipelined: the LDS reads of pair i + 1 are issued before the
        // FMAs of pair i (one wave per SIMD here -- nothing else would hide the LDS latency).
        const int npairs = P.TH * (NARROW_TW / 2);
        float xa[MAXJ], xb[MAXJ], na[MAXJ], nb[MAXJ];
        float da[RP], db[RP], nda[RP], ndb[RP];
#define NARROW_LOAD(pi, A, B, DA, DB)                                                                                  \
    {                                                                                                                  \
        const int py_ = (pi) / (NARROW_TW / 2), px_ = ((pi) % (NARROW_TW / 2)) * 2;                                    \
        const float* xp_ = xs + (py_ * P.HW + px_) * CH;                                                               \
        const float* dp_ = dzs + (py_ * NARROW_TW + px_) * RP;                                                         \
        _Pragma("unroll") for (int j = 0; j < MAXJ; j++)                                                               \
        {                                                                                                              \
            A[j] = xp_[toff[j]];                                                                                       \
            B[j] = xp_[toff[j] + CH];                                                                                  \
        }                                                                                                              \
        _Pragma("unroll") for (int r = 0; r < RP; r++)                                                                 \
        {                                                                                                              \
            DA[r] = dp_[r];      /* same address in every lane: broadcast */                                          \
            DB[r] = dp_[RP + r];                                                                                       \
            /* opaque copies: the LDS read returns (DA, DB) as one register pair and the compiler would splat DB into  \
               its packed FMAs from the pair's HIGH half (`v_pk_fma_f32 ... op_sel:[1,0,0]`), the operand form class   \
               of the finding below (k_wgrad_narrow_row); this kernel runs beside the MFMA chain of the train step */  \
            asm volatile("" : "+v"(DA[r]), "+v"(DB[r]));                                                               \
        }                                                                                                              \
    }
#define NARROW_FMA(A, B, DA, DB)                                                                                       \
    _Pragma("unroll") for (int j = 0; j < MAXJ; j++) _Pragma("unroll") for (int r = 0; r < R; r++) acc[j][r] =         \
        fmaf(DA[r], A[j], acc[j][r]);                                                                                  \
    _Pragma("unroll") for (int j = 0; j < MAXJ; j++) _Pragma("unroll") for (int r = 0; r < R; r++) acc[j][r] =         \
        fmaf(DB[r], B[j], acc[j][r]);
        NARROW_LOAD(0, xa, xb, da, db);
        for (int pi = 0; pi < npairs; pi += 2) {  // npairs = 16 TH is even: two register sets, no copies
            NARROW_LOAD(pi + 1, na, nb, nda, ndb);
            __builtin_amdgcn_sched_barrier(0);
            NARROW_FMA(xa, xb, da, db);
            __builtin_amdgcn_sched_barrier(0);
            const int nxt = pi + 2 < npairs ? pi + 2 : pi;
            NARROW_LOAD(nxt, xa, xb, da, db);
            __builtin_amdgcn_sched_barrier(0);
            NARROW_FMA(na, nb, nda, ndb);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const int ncols = P.taps.n * P.Cc;
#pragma unroll
    for (int j = 0; j < MAXJ; j++) {
        const int t = tg + TG * j;
        if (t >= P.taps.n) continue;
#pragma unroll
        for (int r = 0; r < R; r++) unsafeAtomicAdd(P.dw + (size_t)r * ncols + t * P.Cc + c0 + cc, acc[j][r]);
    }
}

template <int R, int CH>
static int launch_narrow(const NarrowParams& P, int need_j, dim3 grid, size_t lds_bytes, hipStream_t st)
{
    hipError_t e;
    if (need_j <= 4) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_narrow<R, CH, 4>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e == hipSuccess) hipLaunchKernelGGL((k_wgrad_narrow<R, CH, 4>), grid, dim3(256), lds_bytes, st, P);
    } else {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_narrow<R, CH, 13>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e == hipSuccess) hipLaunchKernelGGL((k_wgrad_narrow<R, CH, 13>), grid, dim3(256), lds_bytes, st, P);
    }
    if (e != hipSuccess) return fail(SDN_ELAUNCH, "sdn_conv_wgrad_narrow: LDS size: %s", hipGetErrorString(e));
    return check_launch("k_wgrad_narrow");
}

// ---- r05: the same sum for DENSE 7 x 7 windows (the generator / encoder heads), one LDS read of x per 7 R FMAs.
// k_wgrad_narrow reads one x value from LDS per R FMAs (a thread's taps are unrelated positions of the patch): at R = 4 that is
// 10 LDS instructions per 16 packed FMAs, the LDS pipe saturates before the vector ALUs do (1.45 ms for the generator head where
// the FMAs alone need ~0.4; the exact-row builds of r05l confirmed it: fewer FMAs, same time).  Here a thread owns one tap ROW
// (dy, all 7 dx) of one channel and walks output rows: the 32 + 6 inputs of a row are read once and every output position x
// uses row[x .. x + 6] against the R broadcast d(out) values -- 38 + 32 LDS reads for 224 R FMAs.
//   workgroup = 7 waves (wave = dy), lane = channel + 16 q, thread (dy, q, channel) takes the tile's output rows q and q + 4;
//   LDS row pitch 41 positions x 16 channels (= 16 banks mod 64): the four rows a wave reads are conflict-free; the d(out) rows
//   are padded by 4 floats for the same reason (four broadcast addresses per read);
//   partial sums of the four q meet by two cross-lane adds at the end, one atomic per (tap, channel, row) and workgroup.
// Measured (MI355X, bs 4, 384 x 1248; profiles/r05n..t_narrow_lab.log): generator head 64 -> 3  1.34 -> 0.75-0.81 ms, encoder head
// 16 -> 5  0.52 -> 0.28 ms; the same sums to seven digits.  What bounds it: the packed FMAs -- 14 v_pk_fma_f32 per position and
// thread are 0.37 ms at 4 cycles each and 2.0 GHz; with the loads skipped the kernel takes 0.52-0.60 ms on zeros and 0.67 on random
// data (the clock gives way under full-rate fp32), the patch loads add ~0.13.  Tried on top and dropped (no gain, r05r..t): the next
// tile's loads issued before this tile's FMAs (registers: one workgroup per CU), d(out) reads pipelined two positions ahead.
constexpr int NROW_K = 7, NROW_TH = 8, NROW_NT = NROW_K * 64, NROW_HW = NARROW_TW + NROW_K - 1, NROW_HWP = 41;
constexpr int NROW_HH = NROW_TH + NROW_K - 1;
static_assert(NROW_HWP >= NROW_HW && NROW_HWP % 4 == 1, "row pitch: 16 banks further per patch row");

struct NarrowRowParams {
    const float* rows;  // [N, QH, QW, Cr]
    const float* gath;  // [N, GH, GW, Cc]
    float* dw;          // [Cr, 49 * Cc] fp32, added to (atomics)
    int N, QH, QW, Cr, GH, GW, Cc;
    int pad_mode, relu_rows, relu_gath;
    int dy_min, dx_min, rows_used;
    int tiles_x, tiles_per_image;
    unsigned char tap_of[NROW_K * NROW_K];   // position of tap (dy_min + i, dx_min + j) in the caller's tap list
};

typedef float nrow_f32x2 __attribute__((ext_vector_type(2)));
// acc.xy += g.xy * (xp.x, xp.x) / (xp.y, xp.y): v_pk_fma_f32 with one half of a register pair broadcast, so the 38 inputs of a
// row stay in the 19 pairs ds_read2_b32 delivered them in.
// r06: NO inline asm.  r05 wrote these as `v_pk_fma_f32 ... op_sel:[0,1,0] op_sel_hi:[1,1,1]` (the LOW result lane reading the
// HIGH half of the input pair), to save the v_mov per odd element that hipcc spends to bring it into a low half.  That form
// returns WRONG LOW RESULTS on gfx950 whenever a wave of an MFMA kernel shares the SIMD: alone on the chip the kernel is exact
// (5e-7), beside sdn_conv_head_mfma or sdn_conv_gemm on another stream every EVEN output row of the head's weight gradient was
// off by 1e-4 ... 2e-3 and every odd row exact, whatever buffers the other kernel touched (tools/lab/head_race3.py; a build with
// two scalar v_fma_f32 in asm is exact too).  In the product this is the weight-gradient side stream beside the data-gradient
// chain: the generator / encoder heads' gradients of the r05 default schedule carried that error (found by the batch-4 oracle
// gate of tests/test_gpu_textural_fullsize.py, 3e-4).  The compiler's own selection only ever broadcasts a LOW half
// (`op_sel_hi:[1,0,1]`) and moves odd elements first -- 448 packed FMAs + ~60 moves per row pair instead of 448.
#if defined(SDN_LAB_PKFMA) && SDN_LAB_PKFMA == 3      // lab: the r05 asm (WRONG beside an MFMA kernel, see above)
__device__ __forceinline__ void pk_fma_bcast_lo(nrow_f32x2& acc, const nrow_f32x2 g, const nrow_f32x2 xp)
{
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(g), "v"(xp));
}
__device__ __forceinline__ void pk_fma_bcast_hi(nrow_f32x2& acc, const nrow_f32x2 g, const nrow_f32x2 xp)
{
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(g), "v"(xp));
}
#else
__device__ __forceinline__ void pk_fma_bcast_lo(nrow_f32x2& acc, const nrow_f32x2 g, const nrow_f32x2 xp)
{
    acc[0] = __builtin_fmaf(g[0], xp[0], acc[0]);
    acc[1] = __builtin_fmaf(g[1], xp[0], acc[1]);
}
__device__ __forceinline__ void pk_fma_bcast_hi(nrow_f32x2& acc, const nrow_f32x2 g, const nrow_f32x2 xp)
{
    acc[0] = __builtin_fmaf(g[0], xp[1], acc[0]);
    acc[1] = __builtin_fmaf(g[1], xp[1], acc[1]);
}
#endif

// NP: pairs of d(out) rows a thread accumulates (rows_used 2..4 -> 2, 5..6 -> 3, 7..8 -> 4)
template <int NP>
__global__ __launch_bounds__(NROW_NT) void k_wgrad_narrow_row(const NarrowRowParams P)
{
    constexpr int RP = NP <= 2 ? 4 : 8, DZP = NARROW_TW * RP + 4;   // d(out) values per position / floats per d(out) row
    static_assert(NROW_HW % 2 == 0, "row inputs are held as register pairs");
    __shared__ __attribute__((aligned(16))) float xs[NROW_HH * NROW_HWP * 16];
    __shared__ __attribute__((aligned(16))) float dzs[NROW_TH * DZP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int dyi = tid >> 6, q = lane >> 4, cc = lane & 15;
    const int c0 = blockIdx.y * 16;
    const int R = P.rows_used;

    nrow_f32x2 acc[NROW_K][NP];
#pragma unroll
    for (int j = 0; j < NROW_K; j++)
#pragma unroll
        for (int h = 0; h < NP; h++) acc[j][h] = nrow_f32x2{0.f, 0.f};

    const int total = P.tiles_per_image * P.N;
    for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
        const int n = tile / P.tiles_per_image;
        const int ti = tile - n * P.tiles_per_image;
        const int y0 = (ti / P.tiles_x) * NROW_TH, x0 = (ti % P.tiles_x) * NARROW_TW;
        __syncthreads();  // the previous tile is consumed
        {   // the patch (four float4 per position) and the tile's d(out) values (RP / 4 float4 per position): every load of a
            // thread is in flight before the first LDS store (a load -> store loop pays one memory latency per iteration;
            // the d(out) loop alone was three of them)
            constexpr int PL = NROW_NT / 4, HPOS = NROW_HH * NROW_HW, NU = (HPOS + PL - 1) / PL;
            constexpr int DV = RP / 4, DN = NROW_TH * NARROW_TW * DV, ND = (DN + NROW_NT - 1) / NROW_NT;
            const int c4 = tid & 3, pw = tid >> 2;
            f32x4 v[NU], d[ND];
#pragma unroll
            for (int u = 0; u < NU; u++) {
                const int pos = u * PL + pw;
                const int hy = pos / NROW_HW, hx = pos - hy * NROW_HW;
                int gy = y0 + hy + P.dy_min, gx = x0 + hx + P.dx_min;
                v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (pos < HPOS && resolve_coord(gy, P.GH, P.pad_mode) && resolve_coord(gx, P.GW, P.pad_mode))
                    v[u] = *reinterpret_cast<const f32x4*>(P.gath + (((size_t)n * P.GH + gy) * P.GW + gx) * P.Cc + c0 + 4 * c4);
            }
#pragma unroll
            for (int u = 0; u < ND; u++) {
                const int idx = u * NROW_NT + tid;
                const int p = idx / DV, part = idx - p * DV;
                const int py = p / NARROW_TW, px = p - py * NARROW_TW;
                const int qy = y0 + py, qx = x0 + px;
                d[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (idx < DN && qy < P.QH && qx < P.QW)   // (Cr is a multiple of 16: channels 0 .. RP - 1 exist)
                    d[u] = *reinterpret_cast<const f32x4*>(P.rows + (((size_t)n * P.QH + qy) * P.QW + qx) * P.Cr + 4 * part);
            }
#pragma unroll
            for (int u = 0; u < NU; u++) {
                const int pos = u * PL + pw;
                const int hy = pos / NROW_HW, hx = pos - hy * NROW_HW;
                if (P.relu_gath) {
#pragma unroll
                    for (int e = 0; e < 4; e++) v[u][e] = fmaxf(v[u][e], 0.f);
                }
                if (pos < HPOS) *reinterpret_cast<f32x4*>(xs + (hy * NROW_HWP + hx) * 16 + 4 * c4) = v[u];
            }
            // rows behind rows_used are zeroed: they contribute nothing
#pragma unroll
            for (int u = 0; u < ND; u++) {
                const int idx = u * NROW_NT + tid;
                const int p = idx / DV, part = idx - p * DV;
                const int py = p / NARROW_TW, px = p - py * NARROW_TW;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float w = d[u][e];
                    if (P.relu_rows) w = fmaxf(w, 0.f);
                    d[u][e] = 4 * part + e < R ? w : 0.f;
                }
                if (idx < DN) *reinterpret_cast<f32x4*>(dzs + py * DZP + px * RP + 4 * part) = d[u];
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int yy = 0; yy < 2; yy++) {
            const int y = q + 4 * yy;
            const float* xr = xs + ((y + dyi) * NROW_HWP) * 16 + cc;
            const float* dr = dzs + y * DZP;
            nrow_f32x2 rowp[NROW_HW / 2];
#pragma unroll
            for (int k = 0; k < NROW_HW / 2; k++) rowp[k] = nrow_f32x2{xr[(2 * k) * 16], xr[(2 * k + 1) * 16]};
#pragma unroll
            for (int x = 0; x < NARROW_TW; x++) {
                nrow_f32x2 g[NP];
#pragma unroll
                for (int h = 0; h < NP; h++) g[h] = *reinterpret_cast<const nrow_f32x2*>(dr + x * RP + 2 * h);   // one address per q
#pragma unroll
                for (int j = 0; j < NROW_K; j++)
#pragma unroll
                    for (int h = 0; h < NP; h++) {
                        if ((x + j) & 1)
                            pk_fma_bcast_hi(acc[j][h], g[h], rowp[(x + j) >> 1]);
                        else
                            pk_fma_bcast_lo(acc[j][h], g[h], rowp[(x + j) >> 1]);
                    }
            }
        }
    }
    // the four q of a (dy, channel) meet in lane q = 0
    const int ncols = NROW_K * NROW_K * P.Cc;
#pragma unroll
    for (int j = 0; j < NROW_K; j++) {
        const int t = P.tap_of[dyi * NROW_K + j];
#pragma unroll
        for (int r = 0; r < 2 * NP; r++) {
            float v = acc[j][r >> 1][r & 1];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (q == 0 && r < R) unsafeAtomicAdd(P.dw + (size_t)r * ncols + t * P.Cc + c0 + cc, v);
        }
    }
}

template <int NP>
static int launch_narrow_row(const NarrowRowParams& P, dim3 grid, hipStream_t st)
{
    hipLaunchKernelGGL((k_wgrad_narrow_row<NP>), grid, dim3(NROW_NT), 0, st, P);
    return check_launch("k_wgrad_narrow_row");
}

}  // namespace sdn

using namespace sdn;

// the row kernel takes dense 7 x 7 windows with 2..8 rows (SDN_WGRAD_NARROW_ROW=0: the column kernel for everything, lab switch)
static bool narrow_row_window(int ntaps, const int8_t* dy, const int8_t* dx, int dy_min, int dx_min, unsigned char* tap_of)
{
    static const bool off = [] { const char* e = getenv("SDN_WGRAD_NARROW_ROW"); return e && e[0] == '0'; }();
    if (off || ntaps != NROW_K * NROW_K) return false;
    bool seen[NROW_K * NROW_K] = {};
    for (int t = 0; t < ntaps; t++) {
        const int i = dy[t] - dy_min, j = dx[t] - dx_min;
        if (i < 0 || i >= NROW_K || j < 0 || j >= NROW_K || seen[i * NROW_K + j]) return false;
        seen[i * NROW_K + j] = true;
        tap_of[i * NROW_K + j] = (unsigned char)t;
    }
    return true;
}

SDN_API int sdn_conv_wgrad_narrow(const float* rows, const float* gath, float* dw, int N, int QH, int QW, int Cr,
                                  int rows_used, int GH, int GW, int Cc, int ntaps, const int8_t* dy, const int8_t* dx,
                                  int pad_mode, int relu_rows, int relu_gath, sdnStream stream)
{
    if (!rows || !gath || !dw || !dy || !dx) return fail(SDN_EINVAL, "sdn_conv_wgrad_narrow: null pointer");
    if (ntaps < 1 || ntaps > CONV_MAX_TAPS) return fail(SDN_EINVAL, "sdn_conv_wgrad_narrow: ntaps %d not in 1..%d", ntaps, CONV_MAX_TAPS);
    if ((Cr & 15) || (Cc & 15)) return fail(SDN_EINVAL, "sdn_conv_wgrad_narrow: channel counts must be padded to 16 (%d, %d)", Cr, Cc);
    if (rows_used < 1 || rows_used > 8 || rows_used > Cr) return fail(SDN_EINVAL, "sdn_conv_wgrad_narrow: rows_used %d not in 1..8", rows_used);
    if (N < 1 || QH < 1 || QW < 1) return fail(SDN_EINVAL, "sdn_conv_wgrad_narrow: bad geometry");
    NarrowParams P;
    P.rows = rows; P.gath = gath; P.dw = dw;
    P.N = N; P.QH = QH; P.QW = QW; P.Cr = Cr; P.GH = GH; P.GW = GW; P.Cc = Cc;
    P.pad_mode = pad_mode; P.relu_rows = relu_rows; P.relu_gath = relu_gath;
    P.taps.n = ntaps;
    int dy_min = dy[0], dy_max = dy[0], dx_min = dx[0], dx_max = dx[0];
    for (int t = 0; t < ntaps; t++) {
        P.taps.dy[t] = dy[t];
        P.taps.dx[t] = dx[t];
        dy_min = dy[t] < dy_min ? dy[t] : dy_min;
        dy_max = dy[t] > dy_max ? dy[t] : dy_max;
        dx_min = dx[t] < dx_min ? dx[t] : dx_min;
        dx_max = dx[t] > dx_max ? dx[t] : dx_max;
    }
    if (rows_used >= 2) {
        NarrowRowParams Q;
        if (narrow_row_window(ntaps, dy, dx, dy_min, dx_min, Q.tap_of)) {
            Q.rows = rows; Q.gath = gath; Q.dw = dw;
            Q.N = N; Q.QH = QH; Q.QW = QW; Q.Cr = Cr; Q.GH = GH; Q.GW = GW; Q.Cc = Cc;
            Q.pad_mode = pad_mode; Q.relu_rows = relu_rows; Q.relu_gath = relu_gath;
            Q.dy_min = dy_min; Q.dx_min = dx_min; Q.rows_used = rows_used;
            Q.tiles_x = (QW + NARROW_TW - 1) / NARROW_TW;
            Q.tiles_per_image = Q.tiles_x * ((QH + NROW_TH - 1) / NROW_TH);
            const int chunks = Cc / 16, total = Q.tiles_per_image * N;
            int workers = (2 * 256 + chunks - 1) / chunks;   // two workgroups per CU (registers): every one walks total / workers tiles
            if (workers > total) workers = total;
            const dim3 grid((unsigned)workers, (unsigned)chunks);
            hipStream_t st = (hipStream_t)stream;
            TimedLaunch timed(TIME_CONV_NARROW, st, 2.0 * (double)N * QH * QW * ntaps * rows_used * Cc);
            if (rows_used <= 4) return launch_narrow_row<2>(Q, grid, st);
            if (rows_used <= 6) return launch_narrow_row<3>(Q, grid, st);
            return launch_narrow_row<4>(Q, grid, st);
        }
    }
    // 64-channel chunks (one 140 KB workgroup per CU) only when there are too few tiles to occupy the chip four deep
    const long ntile_est = (long)N * ((QH + 7) / 8) * ((QW + NARROW_TW - 1) / NARROW_TW);
    const int CH = (Cc % 64 == 0 && ntile_est < 256) ? 64 : 16;
    const int RP = rows_used == 1 ? 1 : (rows_used <= 4 ? 4 : 8);
    P.dy_min = dy_min; P.dx_min = dx_min;
    P.HW = NARROW_TW + dx_max - dx_min;
    P.hw_magic = 65536u / (unsigned)P.HW + 1u;
    // tile height: as tall as fits in ~144 KB of LDS for 64 channels (one workgroup per CU), ~36 KB for 16 (four)
    const size_t budget = CH == 64 ? 144 * 1024 : 36 * 1024;
    int TH = 8;
    for (;;) {
        const size_t need = ((size_t)(TH + dy_max - dy_min) * P.HW * CH + (size_t)TH * NARROW_TW * RP) * sizeof(float);
        if (need <= budget || TH == 1) break;
        TH >>= 1;
    }
    P.TH = TH;
    P.HH = TH + dy_max - dy_min;
    const size_t lds_bytes = ((size_t)P.HH * P.HW * CH + (size_t)TH * NARROW_TW * RP) * sizeof(float);
    if (lds_bytes > 160 * 1024 || (size_t)P.HH * P.HW * P.HW >= 65536)
        return fail(SDN_EINVAL, "sdn_conv_wgrad_narrow: tap window too large");
    P.tiles_x = (QW + NARROW_TW - 1) / NARROW_TW;
    P.tiles_per_image = P.tiles_x * ((QH + TH - 1) / TH);
    const int chunks = Cc / CH;
    const int total = P.tiles_per_image * N;
    const int per_cu = CH == 64 ? 1 : 4;
    int workers = (256 * per_cu + chunks - 1) / chunks;
    if (workers > total) workers = total;
    const dim3 grid((unsigned)workers, (unsigned)chunks);
    const int TG = 256 / CH;
    const int need_j = (ntaps + TG - 1) / TG;
    if (need_j > 13) return fail(SDN_EINVAL, "sdn_conv_wgrad_narrow: %d taps do not fit the column schedule", ntaps);
    hipStream_t st = (hipStream_t)stream;
    TimedLaunch timed(TIME_CONV_NARROW, st, 2.0 * (double)N * QH * QW * ntaps * rows_used * Cc);
    const int R = rows_used == 1 ? 1 : (rows_used <= 4 ? 4 : 8);
    if (CH == 64) {
        if (R == 1) return launch_narrow<1, 64>(P, need_j, grid, lds_bytes, st);
        if (R == 4) return launch_narrow<4, 64>(P, need_j, grid, lds_bytes, st);
        return launch_narrow<8, 64>(P, need_j, grid, lds_bytes, st);
    }
    if (R == 1) return launch_narrow<1, 16>(P, need_j, grid, lds_bytes, st);
    if (R == 4) return launch_narrow<4, 16>(P, need_j, grid, lds_bytes, st);
    return launch_narrow<8, 16>(P, need_j, grid, lds_bytes, st);
}

// ---------------------------------------------------------------------------------------------------------------------
// Forward (and narrow data gradient) of stride-1 convolutions with <= 8 output channels: exact fp32 on the vector ALUs.
//
//     out[n, q, r] = act( bias[r] + sum_{dy, dx, c}  f(in[n, q + (dy, dx), c]) * W[dy, dx, c, r] )       r < R <= 8
//
// Reference layers: GlobalGenerator c7s1-3 + tanh (networks.py:236), Encoder c7s1-5 + tanh (:306), the discriminators'
// last 4x4 conv (:437); also the data gradient of the generator's stem restricted to the encoder-feature channels.
// On the matrix cores these layers are bound by re-gathering and re-splitting the input once per tap for a 32-column
// tile of which 1-5 columns are used (2.0 ms for the generator head at 384 x 1248).  Here a workgroup fills LDS with the
// tile + halo ONCE, as channel planes [c][y][x] (lanes = consecutive x: conflict-free 16-B reads), a thread owns 4
// consecutive outputs x R channels, each wave a quarter of the input channels (partial sums meet in LDS at the end), and
// the weights are wave-uniform scalar loads: per (c, dy) three 16-B LDS reads feed KW x 4 x R FMAs.  fp32 VALU roofline.
namespace sdn {

struct NarrowFwdParams {
    const float* in;    // [N, IH, IW, Cip]
    float* out;         // [N, OH, OW, Cop]   (OH, OW) = (QH, QW)
    const float* w;     // [KH][KW][Cip][RP]  dense tap window, zero where a tap / channel / row is absent
    const float* bias;  // [>= R] or null
    int N, IH, IW, Cip, QH, QW, Cop;
    int KH, dy_min, dx_min;
    int pad_mode, in_relu, act;
    int tiles_x, tiles_per_image;
    int HH, HWp, plane;  // halo rows, padded row pitch (floats), plane size (floats)
    unsigned hw_magic;   // pos / HW for the fill loop (HW = 32 + KW - 1 columns)
    int HW;
    int ch_pass;         // input channels per LDS pass: 16 (36 KB, four workgroups per CU hide the scalar-load and LDS
                         // latencies of each other) when there are tiles enough to fill the chip that way, else 64
};

constexpr int NF_TH = 8, NF_TW = 32;  // 256 outputs per tile

template <int R, int KW>
__global__ __launch_bounds__(256) void k_conv_narrow_fwd(const NarrowFwdParams P)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int RP = R == 1 ? 1 : (R <= 4 ? 4 : 8);
    constexpr int NX = (4 + KW - 1 + 3) / 4;  // 16-B reads covering the 4 + KW - 1 inputs of one row
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int py = lane >> 3, px0 = (lane & 7) * 4;
    const int tile = blockIdx.x;
    const int n = tile / P.tiles_per_image;
    const int ti = tile - n * P.tiles_per_image;
    const int y0 = (ti / P.tiles_x) * NF_TH, x0 = (ti % P.tiles_x) * NF_TW;

    float acc[4][R];
#pragma unroll
    for (int o = 0; o < 4; o++)
#pragma unroll
        for (int r = 0; r < R; r++) acc[o][r] = 0.f;

    const int hpos = P.HH * P.HW;
    for (int c0 = 0; c0 < P.Cip; c0 += P.ch_pass) {
        const int nch = min(P.ch_pass, P.Cip - c0);  // multiple of 16
        __syncthreads();                         // planes of the previous pass are consumed
        // fill: lanes = consecutive positions of one channel quad (conflict-free plane stores; the 64-B input records
        // are revisited by the other quads out of the vector cache); eight loads in flight per thread
        const int nquads = nch >> 2;  // multiple of 4
        for (int pb = 0; pb < hpos; pb += 256) {
            const int pos = pb + tid;
            const int hy = (int)(((unsigned)pos * P.hw_magic) >> 16), hx = pos - hy * P.HW;
            int gy = y0 + hy + P.dy_min, gx = x0 + hx + P.dx_min;
            const bool ok = pos < hpos && resolve_coord(gy, P.IH, P.pad_mode) && resolve_coord(gx, P.IW, P.pad_mode);
            const float* src = P.in + (((size_t)n * P.IH + gy) * P.IW + gx) * P.Cip + c0;
            float* dst = lds + hy * P.HWp + hx;
            for (int qb = 0; qb < nquads; qb += 4) {
                f32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (ok) v[u] = *reinterpret_cast<const f32x4*>(src + 4 * (qb + u));
                }
                if (pos < hpos) {
#pragma unroll
                    for (int u = 0; u < 4; u++)
#pragma unroll
                        for (int e = 0; e < 4; e++)
                            dst[(4 * (qb + u) + e) * P.plane] = P.in_relu ? fmaxf(v[u][e], 0.f) : v[u][e];
                }
            }
        }
        __syncthreads();
        // this wave's channels: c = wave, wave + 4, ...
        for (int c = wave; c < nch; c += 4) {
            const float* pl = lds + c * P.plane + py * P.HWp + px0;
            const float* wc = P.w + (size_t)(c0 + c) * RP;
#pragma unroll
            for (int dyi = 0; dyi < KW; dyi++) {  // square windows: KH == KW (checked by the launcher)
                float xr[4 * NX];
#pragma unroll
                for (int k = 0; k < NX; k++) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(pl + dyi * P.HWp + 4 * k);
                    xr[4 * k] = t[0];
                    xr[4 * k + 1] = t[1];
                    xr[4 * k + 2] = t[2];
                    xr[4 * k + 3] = t[3];
                    // opaque copies: the FMAs below are paired over r with x splat; taken from the 16-byte LDS read as it lies, every
                    // other x would be splat from the HIGH half of its register pair (`v_pk_fma_f32 ... op_sel:[1,0,0]`: 196 of them
                    // in the 8-row 7 x 7 build) -- the operand class of the gfx950 finding above
                    asm volatile("" : "+v"(xr[4 * k]), "+v"(xr[4 * k + 1]), "+v"(xr[4 * k + 2]), "+v"(xr[4 * k + 3]));
                }
#pragma unroll
                for (int dxi = 0; dxi < KW; dxi++) {
                    const float* wt = wc + (size_t)(dyi * KW + dxi) * P.Cip * RP;  // wave-uniform: scalar loads
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const float wv = wt[r];
#pragma unroll
                        for (int o = 0; o < 4; o++) acc[o][r] = fmaf(xr[o + dxi], wv, acc[o][r]);
                    }
                }
            }
        }
    }
    // partial sums of the four waves -> LDS [wave][output][R] -> one thread per output position
    __syncthreads();
    float* red = lds;
#pragma unroll
    for (int o = 0; o < 4; o++)
#pragma unroll
        for (int r = 0; r < R; r++) red[(wave * 256 + py * NF_TW + px0 + o) * R + r] = acc[o][r];
    __syncthreads();
    const int oy = y0 + (tid >> 5), ox = x0 + (tid & 31);
    if (oy < P.QH && ox < P.QW) {
        float* dst = P.out + (((size_t)n * P.QH + oy) * P.QW + ox) * P.Cop;
        float res[8];
#pragma unroll
        for (int r = 0; r < 8; r++) res[r] = 0.f;
#pragma unroll
        for (int r = 0; r < R; r++) {
            float v = red[tid * R + r] + red[(256 + tid) * R + r] + red[(512 + tid) * R + r] + red[(768 + tid) * R + r];
            if (P.bias) v += P.bias[r];
            if (P.act == 1)
                v = v > 0.f ? v : 0.2f * v;
            else if (P.act == 2)
                v = tanhf(v);
            res[r] = v;
        }
        // rows_used <= R: the rows behind it have zero weights and zero bias; the padded channels stay zero
        for (int cq = 0; cq < P.Cop; cq += 4) {
            f32x4 o4 = {0.f, 0.f, 0.f, 0.f};
            if (cq < 8) o4 = f32x4{res[cq], res[cq + 1], res[cq + 2], res[cq + 3]};
            *reinterpret_cast<f32x4*>(dst + cq) = o4;
        }
    }
}

template <int R, int KW>
static int launch_narrow_fwd(const NarrowFwdParams& P, dim3 grid, size_t lds_bytes, hipStream_t st)
{
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_narrow_fwd<R, KW>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return fail(SDN_ELAUNCH, "sdn_conv_narrow_fwd: LDS size: %s", hipGetErrorString(e));
    hipLaunchKernelGGL((k_conv_narrow_fwd<R, KW>), grid, dim3(256), lds_bytes, st, P);
    return check_launch("k_conv_narrow_fwd");
}

}  // namespace sdn

SDN_API int sdn_conv_narrow_fwd(const float* in, int N, int IH, int IW, int Cip, float* out, int QH, int QW, int Cop,
                                int rows_used, const float* w_dense, int KH, int KW, int dy_min, int dx_min, int pad_mode,
                                int in_relu, const float* bias, int act, sdnStream stream)
{
    if (!in || !out || !w_dense) return fail(SDN_EINVAL, "sdn_conv_narrow_fwd: null pointer");
    if ((Cip & 15) || (Cop & 15)) return fail(SDN_EINVAL, "sdn_conv_narrow_fwd: channel counts must be padded to 16 (%d, %d)", Cip, Cop);
    if (rows_used < 1 || rows_used > 8 || rows_used > Cop) return fail(SDN_EINVAL, "sdn_conv_narrow_fwd: rows_used %d not in 1..8", rows_used);
    if (KW != 7 && KW != 4 && KW != 3) return fail(SDN_EINVAL, "sdn_conv_narrow_fwd: kernel width %d not built (3, 4, 7)", KW);
    if (KH != KW || N < 1 || QH < 1 || QW < 1) return fail(SDN_EINVAL, "sdn_conv_narrow_fwd: bad geometry (square windows only)");
    NarrowFwdParams P;
    P.in = in; P.out = out; P.w = w_dense; P.bias = bias;
    P.N = N; P.IH = IH; P.IW = IW; P.Cip = Cip; P.QH = QH; P.QW = QW; P.Cop = Cop;
    P.KH = KH; P.dy_min = dy_min; P.dx_min = dx_min;
    P.pad_mode = pad_mode; P.in_relu = in_relu; P.act = act;
    P.tiles_x = (QW + NF_TW - 1) / NF_TW;
    P.tiles_per_image = P.tiles_x * ((QH + NF_TH - 1) / NF_TH);
    P.HH = NF_TH + KH - 1;
    P.HW = NF_TW + KW - 1;
    P.HWp = 40;                    // 32 + KW - 1 <= 38 columns, rows 16-B aligned
    P.plane = P.HH * P.HWp + 4;    // + 4: consecutive planes start 4 banks apart
    P.hw_magic = 65536u / (unsigned)P.HW + 1u;
    if ((size_t)P.HH * P.HW * P.HW >= 65536) return fail(SDN_EINVAL, "sdn_conv_narrow_fwd: tap window too large");
    P.ch_pass = (P.tiles_per_image * N >= 256 || Cip < 64) ? 16 : 64;
    const int nch = Cip < P.ch_pass ? Cip : P.ch_pass;
    size_t lds_bytes = (size_t)nch * P.plane * sizeof(float);
    const size_t red_bytes = 4 * 256 * 8 * sizeof(float);
    if (lds_bytes < red_bytes) lds_bytes = red_bytes;
    if (lds_bytes > 160 * 1024) return fail(SDN_EINVAL, "sdn_conv_narrow_fwd: tile does not fit in LDS");
    const dim3 grid((unsigned)(P.tiles_per_image * N));
    hipStream_t st = (hipStream_t)stream;
    TimedLaunch timed(TIME_CONV_NARROW, st, 2.0 * (double)N * QH * QW * KH * KW * Cip * rows_used);
    // accumulator rows per thread: the layers this kernel serves have 1 (discriminator heads), 3 (generator head) and 5
    // (encoder head, stem data gradient towards the encoder features) output channels; the kernel is FMA-bound, so the 3-
    // and 5-row builds do 25 % / 37 % fewer FMAs than the padded 4 / 8 (the weight layout stays padded: RP in the kernel)
    const int R = rows_used == 1 ? 1 : (rows_used <= 3 ? 3 : (rows_used == 4 ? 4 : (rows_used == 5 ? 5 : 8)));
#define NF_CASE(RR, KK) if (R == RR && KW == KK) return launch_narrow_fwd<RR, KK>(P, grid, lds_bytes, st);
    NF_CASE(1, 7) NF_CASE(3, 7) NF_CASE(4, 7) NF_CASE(5, 7) NF_CASE(8, 7)
    NF_CASE(1, 4) NF_CASE(3, 4) NF_CASE(4, 4) NF_CASE(5, 4) NF_CASE(8, 4)
    NF_CASE(1, 3) NF_CASE(3, 3) NF_CASE(4, 3) NF_CASE(5, 3) NF_CASE(8, 3)
    return fail(SDN_EINVAL, "sdn_conv_narrow_fwd: unsupported shape");
}
