// Weight gradient of the textural convolutions on the CDNA4 matrix cores.
//
// Reference: autograd of Conv2d / ConvTranspose2d in textural/models/networks.py (cuDNN wgrad in the reference), run by
// loss_G.backward() / loss_D.backward(), textural/train.py:88-95.
//
//     dW[r, t, c] = sum_{n, q}  a(rows[n, q, r]) * b(gath[n, q*is + d_t, c])
//   - Conv2d:          rows = d loss / d conv-output over the output grid, gath = the layer input (ReLU on load when
//                      the producer deferred it; zero or reflected borders), r = cout, c = cin;
//   - ConvTranspose2d: rows = the layer input over the input grid, gath = d loss / d output (is = stride), r = cin, c = cout.
// GEMM view: M = r (tile 128, 64 or 32), N = (tap, c) in 16-channel groups (tile 128), K = all positions of all images,
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
    float* partials;    // deterministic mode: slice z stores to partials[z * Cr * ncols + ...], reduced in slice order
    int N, QH, QW, Cr, GH, GW, Cc;
    int istride, pad_mode, relu_rows, relu_gath;
    int steps_per_split;
    int row_tiles, col_tiles;
    ConvTapsW taps;
};

// floor(v / d) for the small ranges that occur when a 32-position step crosses rows / images: a compare when v < 2 d,
// else a 16.16 reciprocal multiply (exact for v < 256 and d < 256, see the call sites).
__device__ __forceinline__ int small_div(int v, int d, bool d_large, unsigned magic)
{
    return d_large ? (int)(v >= d) : (int)(((unsigned)v * magic) >> 16);
}

template <int WM, int WN, int TM, int TN, int NPART, bool RELU>
// 3 workgroups per CU (<= 168 VGPRs, 44 KB LDS each)
__global__ __launch_bounds__(256, 3) void k_conv_wgrad(const WgradParams P)
{
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    static_assert(WM * WN == 4 && BN == 128 && (BM == 128 || BM == 64 || BM == 32), "tile shapes");
    constexpr int A_ELEMS = lds_tile_elems(BM), B_ELEMS = lds_tile_elems(BN);
    constexpr int A_GROUPS = BM / 16, B_GROUPS = BN / 16;  // 16-channel groups per tile: threads = 16 pairs x groups
    __shared__ __attribute__((aligned(16))) __bf16 smem[NPART * (A_ELEMS + B_ELEMS)];
    __bf16* As = smem;
    __bf16* Bs = smem + NPART * A_ELEMS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    const int Ptot = P.N * Q;
    const int total_steps = (Ptot + CONV_BK - 1) / CONV_BK;
    const int step_lo = zs * P.steps_per_split;
    const int step_hi = min(step_lo + P.steps_per_split, total_steps);
    if (step_lo >= step_hi) return;

    // role: waves 0-1 stage the rows operand (threads [0, 16*A_GROUPS) of them), waves 2-3 the gathered operand; each
    // thread converts 16 channels of TWO neighbouring positions per step.  The role is a SCALAR (per wave): one load
    // path with a scalar-selected buffer descriptor, no divergent branches around the loads (a per-thread role made the
    // compiler merge the two sides' loads through waterfall loops and register copies that waited for the data right
    // after issuing it; 4-8 % on the 128-row tiles, tools/wgrad_lab.py --ab against the previous build on one box).
    const bool side_a = wave < 2;
    const int lt = tid & 127;
    const int pair = lt & 15, grp = lt >> 4;
    const bool active = side_a ? lt < 16 * A_GROUPS : lt < 16 * B_GROUPS;

    // gathered operand: this thread's 16-channel group has a fixed tap
    const int gpt = P.Cc >> 4;
    const int cg = (c0 >> 4) + grp;
    const int b_tap = cg / gpt;
    const int b_ch = (cg - b_tap * gpt) * 16;
    const bool b_ok = !side_a && active && b_tap < P.taps.n;
    const int b_dy = b_ok ? P.taps.dy[b_tap] : 0, b_dx = b_ok ? P.taps.dx[b_tap] : 0;
    const int a_ch = r0 + grp * 16;
    const bool a_ok = side_a && active && a_ch < P.Cr;

    // Raw buffer loads: positions behind the last one, channel groups behind the last tap and coordinates outside the
    // image (zero padding) become out-of-range offsets, which the hardware answers with zeros -- no branches.
    const float* side_base = side_a ? P.rows : P.gath;
    const size_t side_bytes = side_a ? (size_t)Ptot * P.Cr * 4 : (size_t)P.N * P.GH * P.GW * P.Cc * 4;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)side_base, 0, (int)side_bytes, 0x00020000);
    // scalar position of the step's first element; per-thread positions are that plus 2 * pair (+ 1)
    int sbase = step_lo * CONV_BK;
    int sn = sbase / Q;
    int sy = (sbase - sn * Q) / P.QW;
    int sx = sbase - sn * Q - sy * P.QW;
    const bool qw_large = P.QW >= 32, qh_large = P.QH >= 64;
    const unsigned qw_magic = 65536u / (unsigned)P.QW + 1u, qh_magic = 65536u / (unsigned)P.QH + 1u;
    const int gh2 = 2 * P.GH - 2, gw2 = 2 * P.GW - 2;

#define WG_GATHER_OFF(n_, y_, x_, off_)                                                                                \
    {                                                                                                                  \
        int iy = (y_)*P.istride + b_dy, ix = (x_)*P.istride + b_dx;                                                    \
        if (P.pad_mode) {                                                                                              \
            iy = max(iy, -iy);                                                                                         \
            ix = max(ix, -ix);                                                                                         \
            iy = min(iy, gh2 - iy);                                                                                    \
            ix = min(ix, gw2 - ix);                                                                                    \
        }                                                                                                              \
        const bool ok = b_ok && (n_) < P.N && (unsigned)iy < (unsigned)P.GH && (unsigned)ix < (unsigned)P.GW;          \
        off_ = ok ? (unsigned)(((((n_)*P.GH + iy) * P.GW + ix) * P.Cc + b_ch) * 4) : 0x80000000u;                      \
    }

    // byte offsets (o0: position p0, o1: p0 + 1) of this thread's 64 + 64 B of the tile that starts at position sbase,
    // then (sbase, sn, sy, sx) advance by one step
#define WG_OFFSETS(o0, o1)                                                                                             \
    {                                                                                                                  \
        if (side_a) {                                                                                                  \
            const int p = sbase + 2 * pair;                                                                            \
            o0 = (a_ok && p < Ptot) ? (unsigned)((p * P.Cr + a_ch) * 4) : 0x80000000u;                                 \
            o1 = (a_ok && p + 1 < Ptot) ? o0 + (unsigned)(P.Cr * 4) : 0x80000000u;                                     \
        } else {                                                                                                       \
            int x = sx + 2 * pair, y = sy, n = sn;                                                                     \
            const int wx = small_div(x, P.QW, qw_large, qw_magic);                                                     \
            x -= wx * P.QW;                                                                                            \
            y += wx;                                                                                                   \
            const int wy = small_div(y, P.QH, qh_large, qh_magic);                                                     \
            y -= wy * P.QH;                                                                                            \
            n += wy;                                                                                                   \
            int x1 = x + 1, y1 = y, n1 = n;                                                                            \
            if (x1 >= P.QW) {                                                                                          \
                x1 = 0;                                                                                                \
                y1++;                                                                                                  \
                if (y1 >= P.QH) {                                                                                      \
                    y1 = 0;                                                                                            \
                    n1++;                                                                                              \
                }                                                                                                      \
            }                                                                                                          \
            WG_GATHER_OFF(n, y, x, o0);                                                                                \
            WG_GATHER_OFF(n1, y1, x1, o1);                                                                             \
        }                                                                                                              \
        sbase += CONV_BK;                                                                                              \
        sx += CONV_BK;                                                                                                 \
        while (sx >= P.QW) {                                                                                           \
            sx -= P.QW;                                                                                                \
            sy++;                                                                                                      \
        }                                                                                                              \
        while (sy >= P.QH) {                                                                                           \
            sy -= P.QH;                                                                                                \
            sn++;                                                                                                      \
        }                                                                                                              \
    }
#define WG_LOAD16(off) __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0))

    const bool relu = side_a ? (P.relu_rows != 0) : (P.relu_gath != 0);
    const float relu_floor = relu ? 0.f : -__builtin_inff();
    __bf16* const st_base = side_a ? As : Bs;
    const int st_part = side_a ? A_ELEMS : B_ELEMS;
    constexpr int NPROD = NPART == 2 ? 3 : 1;
    constexpr int NACC = (TM * TN == 1 && NPART == 2) ? 3 : 1;  // single-tile waves: one accumulator per product
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
    const int fr = lane & 31, fkq = (lane >> 5) * 8;

    // Per step: split + transposing LDS stores of the raw tile -> barrier -> loads of the next tile (in flight behind
    // the MFMAs) -> MFMAs -> barrier.
    // The split is NOT interleaved with this wave's MFMAs: with both operands staged through LDS the other workgroups of
    // the CU cover it better than an in-wave schedule did (measured).
    f32x4 U[4], V[4];  // raw tile in flight: positions p0 (U) and p0 + 1 (V), 16 channels each; constant indices only
    {
        unsigned o0, o1;
        WG_OFFSETS(o0, o1);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            U[j] = WG_LOAD16(o0 + 16 * j);
            V[j] = WG_LOAD16(o1 + 16 * j);
        }
    }
    for (int step = step_lo; step < step_hi; step++) {
        const bool more = step + 1 < step_hi;
        unsigned next0 = 0x80000000u, next1 = 0x80000000u;   // (names the offset macro's locals do not shadow)
        if (more) WG_OFFSETS(next0, next1);
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float fx = U[j][k], fy = V[j][k];
                if constexpr (RELU) {
                    asm("v_max_f32 %0, %1, %2" : "=v"(fx) : "v"(fx), "v"(relu_floor));
                    asm("v_max_f32 %0, %1, %2" : "=v"(fy) : "v"(fy), "v"(relu_floor));
                }
                const bf16x2 h = __builtin_convertvector(f32x2{fx, fy}, bf16x2);
                const uint32_t hw = __builtin_bit_cast(uint32_t, h);
                const int off = lds_row(grp * 16 + 4 * j + k) + 2 * pair;
                if (active) *reinterpret_cast<uint32_t*>(st_base + off) = hw;
                if constexpr (NPART == 2) {
                    const float bx = __builtin_bit_cast(float, hw << 16), by = __builtin_bit_cast(float, hw & 0xffff0000u);
                    const bf16x2 l = __builtin_convertvector(f32x2{fx - bx, fy - by}, bf16x2);
                    if (active) *reinterpret_cast<uint32_t*>(st_base + st_part + off) = __builtin_bit_cast(uint32_t, l);
                }
            }
        }
        __syncthreads();
        // the next tile's loads fly behind the MFMAs.  Unconditional: behind the last step the offsets are out of range
        // and the loads return zeros (a branch here makes the compiler keep two register sets and copy between them).
        // (Reloading each quarter's registers right after its split instead, i.e. one split phase earlier, measured
        // the same on 128-row tiles and 6 % slower on the 64-row tile of the full-resolution layers.)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            U[j] = WG_LOAD16(next0 + 16 * j);
            V[j] = WG_LOAD16(next1 + 16 * j);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            bf16x8 a[NPART][TM], b[NPART][TN];
#pragma unroll
            for (int mt = 0; mt < TM; mt++) {
                const int off = lds_row(wm0 + mt * 32 + fr) + ks * 16 + fkq;
                a[0][mt] = *reinterpret_cast<const bf16x8*>(As + off);
                if constexpr (NPART == 2) a[NPART - 1][mt] = *reinterpret_cast<const bf16x8*>(As + A_ELEMS + off);
            }
#pragma unroll
            for (int nt = 0; nt < TN; nt++) {
                const int off = lds_row(wn0 + nt * 32 + fr) + ks * 16 + fkq;
                b[0][nt] = *reinterpret_cast<const bf16x8*>(Bs + off);
                if constexpr (NPART == 2) b[NPART - 1][nt] = *reinterpret_cast<const bf16x8*>(Bs + B_ELEMS + off);
            }
#pragma unroll
            for (int pp = 0; pp < NPROD; pp++)
#pragma unroll
                for (int mt = 0; mt < TM; mt++)
#pragma unroll
                    for (int nt = 0; nt < TN; nt++) {
                        const int ia = (NPART == 2 && pp == 0) ? NPART - 1 : 0, ib = (NPART == 2 && pp == 1) ? NPART - 1 : 0;
                        f32x16& dst = NACC == 3 ? accs[pp][mt][nt] : accs[0][mt][nt];
                        dst = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ia][mt], b[ib][nt], dst, 0, 0, 0);
                    }
        }
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
                if (row >= P.Cr) continue;
                if (P.partials)
                    P.partials[((size_t)zs * P.Cr + row) * ncols + c] = acc[mt][nt][r];
                else
                    unsafeAtomicAdd(P.dw + (size_t)row * ncols + c, acc[mt][nt][r]);
            }
    }
}

// deterministic mode: dw += partials[0] + partials[1] + ... in slice order
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ partials, int zs, size_t elems4,
                                                      float* __restrict__ dw)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= elems4) return;
    f32x4 v = reinterpret_cast<const f32x4*>(dw)[i];
    for (int z = 0; z < zs; z++) v += reinterpret_cast<const f32x4*>(partials)[(size_t)z * elems4 + i];
    reinterpret_cast<f32x4*>(dw)[i] = v;
}

}  // namespace sdn

using namespace sdn;

SDN_API int sdn_conv_wgrad(const float* rows, const float* gath, float* dw, int N, int QH, int QW, int Cr, int GH,
                           int GW, int Cc, int istride, int ntaps, const int8_t* dy, const int8_t* dx, int pad_mode,
                           int relu_rows, int relu_gath, int splits, int precision, void* workspace,
                           size_t workspace_bytes, sdnStream stream)
{
    if (!rows || !gath || !dw || !dy || !dx) return fail(SDN_EINVAL, "sdn_conv_wgrad: null pointer");
    if (ntaps < 1 || ntaps > CONV_MAX_TAPS) return fail(SDN_EINVAL, "sdn_conv_wgrad: ntaps %d not in 1..%d", ntaps, CONV_MAX_TAPS);
    if ((Cr & 15) || (Cc & 15)) return fail(SDN_EINVAL, "sdn_conv_wgrad: channel counts must be padded to 16 (%d, %d)", Cr, Cc);
    if (precision != 1 && precision != 3) return fail(SDN_EINVAL, "sdn_conv_wgrad: precision must be 1 or 3");
    if (N < 1 || QH < 1 || QW < 1 || istride < 1 || splits < 1) return fail(SDN_EINVAL, "sdn_conv_wgrad: bad geometry");
    // both operands are addressed through 32-bit buffer offsets
    if ((size_t)N * QH * QW * Cr * 4 >= 0x7fffff00u || (size_t)N * GH * GW * Cc * 4 >= 0x7fffff00u)
        return fail(SDN_EINVAL, "sdn_conv_wgrad: operands must stay below 2 GiB");
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
    P.partials = nullptr;
    if (workspace && zs > 1) {  // deterministic mode
        if (workspace_bytes < (size_t)zs * Cr * ncols * sizeof(float))
            return fail(SDN_ENOMEM, "sdn_conv_wgrad: workspace %zu < %zu bytes", workspace_bytes,
                        (size_t)zs * Cr * ncols * sizeof(float));
        P.partials = (float*)workspace;
    }
    TimedLaunch timed(TIME_CONV_WGRAD, st, 2.0 * (double)ptot * ntaps * Cr * Cc);
    P.col_tiles = (ncols + 127) / 128;
    P.row_tiles = Cr > 64 ? (Cr + 127) / 128 : 1;
    const bool relu = relu_rows || relu_gath;
#define WG_LAUNCH(WM, WN, TM, TN, NP)                                                                                  \
    if (relu)                                                                                                          \
        hipLaunchKernelGGL((k_conv_wgrad<WM, WN, TM, TN, NP, true>), grid, dim3(256), 0, st, P);                       \
    else                                                                                                               \
        hipLaunchKernelGGL((k_conv_wgrad<WM, WN, TM, TN, NP, false>), grid, dim3(256), 0, st, P);
    if (Cr > 32 && Cr <= 64) {  // 64-row tile: the 64-channel layers at full resolution do not pay for 128 rows
        const dim3 grid((unsigned)(P.col_tiles * zs));
        if (npart == 2) {
            WG_LAUNCH(1, 4, 2, 1, 2)
        } else {
            WG_LAUNCH(1, 4, 2, 1, 1)
        }
    } else if (Cr > 32) {
        const dim3 grid((unsigned)(P.row_tiles * P.col_tiles * zs));
        if (npart == 2) {
            WG_LAUNCH(2, 2, 2, 2, 2)
        } else {
            WG_LAUNCH(2, 2, 2, 2, 1)
        }
    } else {
        const dim3 grid((unsigned)(P.col_tiles * zs));
        if (npart == 2) {
            WG_LAUNCH(1, 4, 1, 1, 2)
        } else {
            WG_LAUNCH(1, 4, 1, 1, 1)
        }
    }
    if (P.partials)
        hipLaunchKernelGGL(k_wgrad_reduce, dim3(cdiv((long)((size_t)Cr * ncols / 4), 256)), dim3(256), 0, st, P.partials, zs,
                           (size_t)Cr * ncols / 4, dw);
    return check_launch("k_conv_wgrad");
}
