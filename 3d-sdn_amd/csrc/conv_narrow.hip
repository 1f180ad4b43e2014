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

}  // namespace sdn

using namespace sdn;

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
    const int CH = (Cc % 64 == 0) ? 64 : 16;
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
    TimedLaunch timed(TIME_CONV_WGRAD, st, 2.0 * (double)N * QH * QW * ntaps * Cr * Cc);
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
