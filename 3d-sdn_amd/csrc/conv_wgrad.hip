// Weight gradient of the textural convolutions on the CDNA4 matrix cores.
//
// Reference: autograd of Conv2d / ConvTranspose2d in textural/models/networks.py (cuDNN wgrad in the reference), run by
// loss_G.backward() / loss_D.backward(), textural/train.py:88-95.
//
//     dW[r, t, c] = sum_{n, q}  a(rows[n, q, r]) * b(gath[n, q*is + d_t, c])
//   - Conv2d:          rows = d loss / d conv-output over the output grid, gath = the layer input (ReLU on load when
//                      the producer deferred it; zero or reflected borders), r = cout, c = cin;
//   - ConvTranspose2d: rows = the layer input over the input grid, gath = d loss / d output (is = stride), r = cin, c = cout.
// GEMM view: M = r (tile 128 or 32), N = (tap, c) in 16-channel groups (tile 128), K = all positions of all images,
// 32 per step, optionally split over blockIdx.z (fp32 atomics combine the slices).  Both operands are channel-major in
// HBM but K-major for the MFMA, so staging transposes: a thread converts 16 channels of TWO neighbouring positions and
// stores (position, position+1) bf16 pairs with 4-byte LDS writes into [channel][k] tiles (conv_common.h explains the
// pitch).  Output dW is fp32 in the packed [r][tap * Cc + c] layout of the forward weights; sdn_conv_unpack_grad maps it
// back to the torch OIHW / IOHW parameter layout.
#include "conv_common.h"
#include "sdn_common.h"

namespace sdn {

struct ConvTapsW {
    int n;
    signed char dy[CONV_MAX_TAPS];
    signed char dx[CONV_MAX_TAPS];
};

struct WgradParams {
    const float* rows;  // [N, QH, QW, Cr]
    const float* gath;  // [N, GH, GW, Cc]
    float* dw;          // [Cr_rows, ntaps * Cc]  fp32, added to
    int N, QH, QW, Cr, GH, GW, Cc;
    int istride, pad_mode, relu_rows, relu_gath;
    int steps_per_split;
    int row_tiles, col_tiles;
    ConvTapsW taps;
};

struct Pos {
    int n, y, x;
};

__device__ __forceinline__ void pos_advance(Pos& p, int by, int QH, int QW)
{
    p.x += by;
    while (p.x >= QW) {
        p.x -= QW;
        p.y++;
    }
    while (p.y >= QH) {
        p.y -= QH;
        p.n++;
    }
}

template <int WM, int WN, int TM, int TN, int NPART>
// 3 workgroups per CU (146 VGPRs, 44 KB LDS each): more latency hiding, and 544-block grids still fit in one round
__global__ __launch_bounds__(256, 3) void k_conv_wgrad(const WgradParams P)
{
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    static_assert(WM * WN == 4 && BN == 128 && (BM == 128 || BM == 32), "tile shapes");
    constexpr int A_ELEMS = lds_tile_elems(BM), B_ELEMS = lds_tile_elems(BN);
    constexpr int A_GROUPS = BM / 16, B_GROUPS = BN / 16;  // 16-channel groups per tile: threads = 16 pairs x groups
    __shared__ __attribute__((aligned(16))) __bf16 smem[NPART * (A_ELEMS + B_ELEMS)];
    __bf16* As = smem;
    __bf16* Bs = smem + NPART * A_ELEMS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware tile order (hardware block b runs on XCD b % 8): every XCD gets a contiguous range of tiles, row tile
    // fastest, so that the blocks sharing a gathered-operand tile (all row tiles of one column tile) and the few row
    // tiles themselves meet in the same L2.
    const unsigned nblk = gridDim.x;
    const unsigned xcd = blockIdx.x & 7u, jj = blockIdx.x >> 3;
    const unsigned q8 = nblk >> 3, r8 = nblk & 7u;
    const unsigned vid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + jj;
    const int rt = (int)(vid % (unsigned)P.row_tiles);
    const unsigned rest = vid / (unsigned)P.row_tiles;
    const int ct = (int)(rest % (unsigned)P.col_tiles);
    const int zs = (int)(rest / (unsigned)P.col_tiles);
    const int r0 = rt * BM;
    const int c0 = ct * BN;  // first flattened (tap, channel) column
    const int Q = P.QH * P.QW;
    const long Ptot = (long)P.N * Q;
    const int total_steps = (int)((Ptot + CONV_BK - 1) / CONV_BK);
    const int step_lo = zs * P.steps_per_split;
    const int step_hi = min(step_lo + P.steps_per_split, total_steps);
    if (step_lo >= step_hi) return;

    // role: threads [0, 16*A_GROUPS) stage the rows operand, threads [128, 128 + 16*B_GROUPS) the gathered operand
    const bool is_a = tid < 16 * A_GROUPS;
    const bool is_b = tid >= 128 && tid < 128 + 16 * B_GROUPS;
    const int lt = is_b ? tid - 128 : tid;
    const int pair = lt & 15, grp = lt >> 4;

    // gathered operand: this thread's 16-channel group has a fixed tap
    const int gpt = P.Cc >> 4;
    const int cg = (c0 >> 4) + grp;
    const int b_tap = cg / gpt;
    const int b_ch = (cg - b_tap * gpt) * 16;
    const bool b_ok = is_b && b_tap < P.taps.n;
    const int b_dy = b_ok ? P.taps.dy[b_tap] : 0, b_dx = b_ok ? P.taps.dx[b_tap] : 0;
    const int a_ch = r0 + grp * 16;
    const bool a_ok = is_a && a_ch < P.Cr;

    // position of the first element of this thread's pair at step_lo
    Pos p0;
    {
        const long pp = (long)step_lo * CONV_BK + 2 * pair;
        p0.n = (int)(pp / Q);
        const int q = (int)(pp - (long)p0.n * Q);
        p0.y = q / P.QW;
        p0.x = q - p0.y * P.QW;
    }

    // next step's data, held in registers behind the MFMAs: two positions x 16 channels
    f32x4 u0, u1, u2, u3, v0, v1, v2, v3;

#define WG_LOAD_ONE(p, d0, d1, d2, d3)                                                                                 \
    {                                                                                                                  \
        const float* src = nullptr;                                                                                    \
        if ((p).n < P.N) {                                                                                             \
            if (is_a) {                                                                                                \
                if (a_ok) src = P.rows + (((size_t)(p).n * P.QH + (p).y) * P.QW + (p).x) * P.Cr + a_ch;                \
            } else if (b_ok) {                                                                                         \
                int iy = (p).y * P.istride + b_dy, ix = (p).x * P.istride + b_dx;                                      \
                if (resolve_coord(iy, P.GH, P.pad_mode) && resolve_coord(ix, P.GW, P.pad_mode))                        \
                    src = P.gath + (((size_t)(p).n * P.GH + iy) * P.GW + ix) * P.Cc + b_ch;                            \
            }                                                                                                          \
        }                                                                                                              \
        d0 = d1 = d2 = d3 = f32x4{0.f, 0.f, 0.f, 0.f};                                                                 \
        if (src) {                                                                                                     \
            const f32x4* s4 = reinterpret_cast<const f32x4*>(src);                                                     \
            d0 = s4[0];                                                                                                \
            d1 = s4[1];                                                                                                \
            d2 = s4[2];                                                                                                \
            d3 = s4[3];                                                                                                \
        }                                                                                                              \
    }

#define WG_LOAD_GLOBAL()                                                                                               \
    if (is_a || is_b) {                                                                                                \
        Pos p1 = p0;                                                                                                   \
        pos_advance(p1, 1, P.QH, P.QW);                                                                                \
        WG_LOAD_ONE(p0, u0, u1, u2, u3);                                                                               \
        WG_LOAD_ONE(p1, v0, v1, v2, v3);                                                                               \
        pos_advance(p0, CONV_BK, P.QH, P.QW);                                                                          \
    }

    const bool relu = is_a ? (P.relu_rows != 0) : (P.relu_gath != 0);
    __bf16* const st_base = is_a ? As : Bs;
    const int st_part = is_a ? A_ELEMS : B_ELEMS;
    auto put4 = [&](f32x4 x0, f32x4 x1, int j) {
#pragma unroll
        for (int e = 0; e < 4; e++) {
            float f0 = x0[e], f1 = x1[e];
            if (relu) {
                f0 = fmaxf(f0, 0.f);
                f1 = fmaxf(f1, 0.f);
            }
            const SplitBf16 s = split2(f0, f1);
            const int off = lds_row(grp * 16 + j * 4 + e) + 2 * pair;
            *reinterpret_cast<bf16x2*>(st_base + off) = s.hi;
            if constexpr (NPART == 2) *reinterpret_cast<bf16x2*>(st_base + st_part + off) = s.lo;
        }
    };
#define WG_STORE_LDS()                                                                                                 \
    if (is_a || is_b) {                                                                                                \
        put4(u0, v0, 0);                                                                                               \
        put4(u1, v1, 1);                                                                                               \
        put4(u2, v2, 2);                                                                                               \
        put4(u3, v3, 3);                                                                                               \
    }

    constexpr int NACC = (TM * TN == 1 && NPART == 2) ? 3 : 1;  // see mfma_step
    f32x16 accs[NACC][TM][TN];
#pragma unroll
    for (int pp = 0; pp < NACC; pp++)
#pragma unroll
        for (int mt = 0; mt < TM; mt++)
#pragma unroll
            for (int nt = 0; nt < TN; nt++)
#pragma unroll
                for (int r = 0; r < 16; r++) accs[pp][mt][nt][r] = 0.f;

    const int wm0 = (wave / WN) * TM * 32, wn0 = (wave % WN) * TN * 32;
    WG_LOAD_GLOBAL();
    for (int step = step_lo; step < step_hi; step++) {
        WG_STORE_LDS();
        __syncthreads();
        if (step + 1 < step_hi) WG_LOAD_GLOBAL();
        mfma_step<TM, TN, NPART, NACC>(As, Bs, wm0, wn0, A_ELEMS, B_ELEMS, lane, accs);
        __syncthreads();
    }

    f32x16(&acc)[TM][TN] = accs[0];
    if constexpr (NACC == 3) acc[0][0] = (accs[0][0][0] + accs[1][0][0]) + accs[2][0][0];
    const int ncols = P.taps.n * P.Cc;
    const int col = lane & 31;
#pragma unroll
    for (int nt = 0; nt < TN; nt++) {
        const int c = c0 + wn0 + nt * 32 + col;
        if (c >= ncols) continue;
#pragma unroll
        for (int mt = 0; mt < TM; mt++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = r0 + wm0 + mt * 32 + mfma_row(r, lane);
                if (row < P.Cr) unsafeAtomicAdd(P.dw + (size_t)row * ncols + c, acc[mt][nt][r]);
            }
    }
}

}  // namespace sdn

using namespace sdn;

SDN_API int sdn_conv_wgrad(const float* rows, const float* gath, float* dw, int N, int QH, int QW, int Cr, int GH,
                           int GW, int Cc, int istride, int ntaps, const int8_t* dy, const int8_t* dx, int pad_mode,
                           int relu_rows, int relu_gath, int splits, int precision, sdnStream stream)
{
    if (!rows || !gath || !dw || !dy || !dx) return fail(SDN_EINVAL, "sdn_conv_wgrad: null pointer");
    if (ntaps < 1 || ntaps > CONV_MAX_TAPS) return fail(SDN_EINVAL, "sdn_conv_wgrad: ntaps %d not in 1..%d", ntaps, CONV_MAX_TAPS);
    if ((Cr & 15) || (Cc & 15)) return fail(SDN_EINVAL, "sdn_conv_wgrad: channel counts must be padded to 16 (%d, %d)", Cr, Cc);
    if (precision != 1 && precision != 3) return fail(SDN_EINVAL, "sdn_conv_wgrad: precision must be 1 or 3");
    if (N < 1 || QH < 1 || QW < 1 || istride < 1 || splits < 1) return fail(SDN_EINVAL, "sdn_conv_wgrad: bad geometry");
    WgradParams P;
    P.rows = rows; P.gath = gath; P.dw = dw;
    P.N = N; P.QH = QH; P.QW = QW; P.Cr = Cr; P.GH = GH; P.GW = GW; P.Cc = Cc;
    P.istride = istride; P.pad_mode = pad_mode; P.relu_rows = relu_rows; P.relu_gath = relu_gath;
    P.taps.n = ntaps;
    for (int t = 0; t < ntaps; t++) {
        P.taps.dy[t] = dy[t];
        P.taps.dx[t] = dx[t];
    }
    const long ptot = (long)N * QH * QW;
    const int total_steps = (int)((ptot + CONV_BK - 1) / CONV_BK);
    if (splits > total_steps) splits = total_steps;
    P.steps_per_split = (total_steps + splits - 1) / splits;
    const int zs = (total_steps + P.steps_per_split - 1) / P.steps_per_split;
    const int ncols = ntaps * Cc;
    hipStream_t st = (hipStream_t)stream;
    const int npart = precision == 3 ? 2 : 1;
    TimedLaunch timed(TIME_CONV_WGRAD, st, 2.0 * (double)ptot * ntaps * Cr * Cc);
    P.col_tiles = (ncols + 127) / 128;
    P.row_tiles = Cr > 32 ? (Cr + 127) / 128 : 1;
    if (Cr > 32) {
        const dim3 grid((unsigned)(P.row_tiles * P.col_tiles * zs));
        if (npart == 2)
            hipLaunchKernelGGL((k_conv_wgrad<2, 2, 2, 2, 2>), grid, dim3(256), 0, st, P);
        else
            hipLaunchKernelGGL((k_conv_wgrad<2, 2, 2, 2, 1>), grid, dim3(256), 0, st, P);
    } else {
        const dim3 grid((unsigned)(P.col_tiles * zs));
        if (npart == 2)
            hipLaunchKernelGGL((k_conv_wgrad<1, 4, 1, 1, 2>), grid, dim3(256), 0, st, P);
        else
            hipLaunchKernelGGL((k_conv_wgrad<1, 4, 1, 1, 1>), grid, dim3(256), 0, st, P);
    }
    return check_launch("k_conv_wgrad");
}
