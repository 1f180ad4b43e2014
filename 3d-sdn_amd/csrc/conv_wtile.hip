// k_wgrad_tile: the weight gradient of conv_wgrad.hip re-built around LDS-DMA, bf16 operand planes and the hardware
// transpose read (r04).
//
// Reference: autograd of Conv2d / ConvTranspose2d in /root/reference/textural/models/networks.py:211-283, 412-461 (cuDNN wgrad in
// the reference), run by loss_G.backward() / loss_D.backward(), /root/reference/textural/train.py:88-95.
//
//     dW[r, t, c] += sum_{n, q}  rows[n, q, r] * gath[n, q * is + d_t, c]
// (Conv2d: rows = d loss / d output, gath = layer input;  ConvTranspose2d: rows = layer input, gath = d loss / d output.)
// GEMM view: M = r, N = (tap, c) in 16-channel groups, K = all positions of all images.  Both operands are channel-contiguous
// in HBM but the MFMA wants 8 consecutive K per lane.  k_conv_wgrad transposed while staging (4-byte LDS stores, two
// barriers per 32-deep step, the split re-done by every workgroup: 24 % matrix-pipe busy).  Here
//   - both operands arrive as bf16 (hi, lo) planes, ReLU already applied by the producer (conv_planes.hip);
//   - LDS-DMA (`global_load_lds_dwordx4`) copies them as they lie: one wave instruction fills a [32 positions][16 channels]
//     sub-tile (1 KiB, position rows of 32 B);
//   - `ds_read_b64_tr_b16` delivers the sub-tile's 4 x 16 blocks transposed: lane i of a 16-lane group receives channel i's
//     four consecutive positions, i.e. half an MFMA operand fragment (two reads per fragment, 256 B / clk, no VALU);
//     odd 16-channel blocks store their positions rotated by four rows (a source-side rotation: free) so that the two
//     16-lane groups a 32-lane LDS cycle serves touch different bank halves;
//   - 8 waves on a (64 TM) x 256 tile (TM in {1, 2}), three LDS stages, counted vmcnt + one raw barrier per step, copies
//     spread between the MFMA groups: the structure of k_conv_tile;
//   - work is dealt out stream-K style: the (tile, K step) space is cut into one contiguous range per workgroup (256 = one
//     per CU), so any shape fills the chip -- 288 tiles of the 1024-channel layers, or 5 tiles x 15k steps of the 64-channel
//     ones -- and a tile receives partial sums (float atomics into the zeroed dW) from the few workgroups that share it.
//     (Deterministic mode keeps k_conv_wgrad's ordered reduction.)
// Numerics: three v_mfma_f32_32x32x16_bf16 products (lo*hi, hi*lo, hi*hi), fp32 accumulation, as everywhere.
#include <cstdlib>
#include <type_traits>

#include "conv_common.h"
#include "conv_dma.h"
#include "sdn_common.h"

namespace sdn {

static __device__ __attribute__((aligned(256))) unsigned g_zero_page_w[64];  // what outside lanes copy from

struct WTileTaps {
    int n;
    signed char dy[CONV_MAX_TAPS];
    signed char dx[CONV_MAX_TAPS];
};

struct WTileParams {
    const __bf16* rows;   // planes [2][N * QH * QW, Cr]
    long rows_stride;
    const __bf16* gath;   // planes [2][N, GH, GW, Cc]
    long gath_stride;
    float* dw;            // [Cr, ntaps * Cc] fp32, added to
    int N, QH, QW, Cr, GH, GW, Cc;
    int istride, pad_mode;
    int row_tiles, col_tiles, total_steps;
    int kslices, slice_steps;   // kslices > 0: the "sliced" decomposition (see k_wgrad_tile), else stream-K ranges
    float inv_q, inv_qw;
    WTileTaps taps;
};

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int WT_STAGES = 3;
constexpr int WT_OUTSIDE = -(1 << 14);

// floor(p / d) for 0 <= p < 2^24, 0 < d < 2^24 through the float reciprocal (exact after one correction each way)
__device__ __forceinline__ int fdiv(int p, int d, float inv)
{
    int q = (int)((float)p * inv);
    int r = p - q * d;
    q += r >= d ? 1 : 0;
    q -= r < 0 ? 1 : 0;
    return q;
}

template <int TM>
__global__ __launch_bounds__(512, 2) void k_wgrad_tile(const WTileParams P)
{
    constexpr int TN = 2;
    constexpr int BM = 64 * TM, BN = 256;                 // waves: 2 (M) x 4 (N), each (32 TM) x 64
    constexpr int A_PLANE = (BM / 16) * 1024, B_PLANE = (BN / 16) * 1024;   // bytes
    constexpr int STAGE = 2 * A_PLANE + 2 * B_PLANE;
    constexpr int GA = TM == 2 ? 2 : 1;
    constexpr int G = GA + 4;                             // copies per thread and step
    __shared__ __attribute__((aligned(1024))) char smem[WT_STAGES * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Q = P.QH * P.QW;
    const int Ptot = P.N * Q;
    const int ncols = P.taps.n * P.Cc;
    const int gpt = P.Cc >> 4;

    // ---- stream-K range of this workgroup (hardware block b runs on XCD b % 8: neighbours in the range share an L2)
    const unsigned nblk = gridDim.x;
    const unsigned xcd = blockIdx.x & 7u, jj = blockIdx.x >> 3;
    const unsigned q8 = nblk >> 3, r8 = nblk & 7u;
    const unsigned vid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + jj;
    const long T = (long)P.row_tiles * P.col_tiles * P.total_steps;
    long lo = T * vid / nblk;
    const long hi = T * (vid + 1) / nblk;

    // ---- copy roles.  One wave instruction fills one [32 positions][16 channels] sub-tile of one plane: lane l -> LDS row
    // (l >> 1), channel half (l & 1).  Sub-tiles of ODD 16-channel blocks hold position k in row (k + 4) & 31; a wave only
    // ever fills blocks of its own parity (A block `wave` or `wave & 3`, B blocks `wave` and `wave + 8`), so the rotation is
    // a per-wave constant of the position this thread fetches.
    const int kslot = ((lane >> 1) - 4 * (wave & 1)) & 31;
    const int chalf = (lane & 1) * 8;
    const int a_cb = TM == 2 ? wave : (wave & 3);
    const int a_pl = TM == 2 ? 0 : (wave >> 2);          // TM == 1: waves 0-3 copy the hi plane, 4-7 the lo plane
    const char* zero = (const char*)g_zero_page_w;
    const char* rows_b = (const char*)P.rows;
    const char* gath_b = (const char*)P.gath;
    const long rows_lo = P.rows_stride * 2, gath_lo = P.gath_stride * 2;
    const int gh2 = 2 * P.GH - 2, gw2 = 2 * P.GW - 2;
    const bool reflect = P.pad_mode != 0;

    // fragment read addresses (ds_read_b64_tr_b16): 16-lane group g -> 16-channel block (g & 1) of the 32-row MFMA tile,
    // K half (g >> 1); lane j of the group reads 8 B at position row key0 + (j >> 2), column quarter (j & 3); the four rows
    // come back transposed: this lane receives channel j's positions key0 .. key0 + 3.
    const int wm0 = (wave >> 2) * TM * 32, wn0 = (wave & 3) * TN * 32;
    const int g4 = lane >> 4, j16 = lane & 15;
    int aoffb[TM][2][2], boffb[TN][2][2];   // [tile][ks][e]
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int key = 16 * ks + 8 * (g4 >> 1) + 4 * e + (j16 >> 2);
            const int slot = (key + 4 * (g4 & 1)) & 31;
#pragma unroll
            for (int mt = 0; mt < TM; mt++)
                aoffb[mt][ks][e] = ((wm0 + mt * 32) / 16 + (g4 & 1)) * 1024 + slot * 32 + (j16 & 3) * 8;
#pragma unroll
            for (int nt = 0; nt < TN; nt++)
                boffb[nt][ks][e] = 2 * A_PLANE + ((wn0 + nt * 32) / 16 + (g4 & 1)) * 1024 + slot * 32 + (j16 & 3) * 8;
        }

    // Sliced decomposition (kslices > 0): work items (K slice, tile), tile fastest, dealt round-robin -- at any time the
    // workgroups of an XCD multiply the SAME K slice of neighbouring tiles (all row tiles of a few column tiles), so the
    // operand panels they stream are shared in that XCD's L2.  (Stream-K ranges put every workgroup at a different K offset
    // of a different tile: 5 % L2 hit rate on the 1024-channel layers, 1.6 GB over the fabric per launch.)
    const long ntile = (long)P.row_tiles * P.col_tiles;
    const long items = ntile * P.kslices;
    long item = vid;
    for (;;) {
        // ---- one segment: K steps [s0, s1) of one tile
        int tile, s0, s1;
        if (P.kslices > 0) {
            if (item >= items) break;
            tile = (int)(item % ntile);
            s0 = (int)(item / ntile) * P.slice_steps;
            s1 = min(P.total_steps, s0 + P.slice_steps);
            item += nblk;
            if (s0 >= s1) continue;
        } else {
            if (lo >= hi) break;
            tile = (int)(lo / P.total_steps);
            s0 = (int)(lo - (long)tile * P.total_steps);
            s1 = (int)min((long)P.total_steps, s0 + (hi - lo));
            lo += s1 - s0;
        }
        const int rt = tile % P.row_tiles, ct = tile / P.row_tiles;   // row tile fastest: neighbours share the gathered operand
        const int r0 = rt * BM, c0 = ct * BN;

        // this wave's sub-tiles: A block a_cb (channels r0 + 16 a_cb ..), B blocks wave, wave + 8 (columns c0 + 16 * block ..)
        const int a_ch = r0 + a_cb * 16 + chalf;
        const bool a_ok = r0 + a_cb * 16 < P.Cr;
        int b_dy[2], b_dx[2], b_ch[2];
        bool b_ok[2];
#pragma unroll
        for (int jb = 0; jb < 2; jb++) {
            const int cg = (c0 >> 4) + wave + 8 * jb;
            const int t = cg / gpt;
            b_ok[jb] = t < P.taps.n;
            const int tt = b_ok[jb] ? t : 0;
            b_dy[jb] = P.taps.dy[tt];
            b_dx[jb] = P.taps.dx[tt];
            b_ch[jb] = (cg - t * gpt) * 16 + chalf;
        }

        // Addresses: `asrc` / `bsrc` / `aval` / `bval` belong to the step whose copies are issued next; the set of the step after
        // that is computed in six pieces (addr_piece<0..5>) placed between the MFMA groups of a step, so that the ~100 VALU
        // instructions of a position decode + three addresses issue in the matrix pipe's shadow instead of in front of it.
        const char* asrc;
        const char* bsrc[2];
        bool aval, bval[2];
        const char* asrc_n;
        const char* bsrc_n[2];
        bool aval_n, bval_n[2];
        int st = s0;   // step whose addresses are computed next
        int a_p, a_n, a_y, a_x, a_iy[2], a_ix[2];
        bool a_pok;
        auto addr_piece = [&](auto k_c) __attribute__((always_inline)) {
            constexpr int k = decltype(k_c)::value;
            if constexpr (k == 0) {
                a_p = st * 32 + kslot;
                a_pok = a_p < Ptot;
                a_n = fdiv(a_p, Q, P.inv_q);
            } else if constexpr (k == 1) {
                const int rem = a_p - a_n * Q;
                a_y = fdiv(rem, P.QW, P.inv_qw);
                a_x = rem - a_y * P.QW;
            } else if constexpr (k == 2) {
                aval_n = ((int)a_pok & (int)a_ok) != 0;
                asrc_n = select_ptr(aval_n, rows_b + (unsigned)(a_p * P.Cr + a_ch) * 2u, zero);
#pragma unroll
                for (int jb = 0; jb < 2; jb++) {
                    a_iy[jb] = a_y * P.istride + b_dy[jb];
                    a_ix[jb] = a_x * P.istride + b_dx[jb];
                }
            } else if constexpr (k == 3) {
#pragma unroll
                for (int jb = 0; jb < 2; jb++) {
                    int ry = max(a_iy[jb], -a_iy[jb]), rx = max(a_ix[jb], -a_ix[jb]);
                    ry = min(ry, gh2 - ry);
                    rx = min(rx, gw2 - rx);
                    a_iy[jb] = reflect ? ry : a_iy[jb];
                    a_ix[jb] = reflect ? rx : a_ix[jb];
                    bval_n[jb] = ((int)a_pok & (int)b_ok[jb] & (int)((unsigned)a_iy[jb] < (unsigned)P.GH) &
                                  (int)((unsigned)a_ix[jb] < (unsigned)P.GW)) != 0;
                }
            } else if constexpr (k == 4 || k == 5) {
                constexpr int jb = k - 4;
                const unsigned off = (unsigned)(((a_n * P.GH + a_iy[jb]) * P.GW + a_ix[jb]) * P.Cc + b_ch[jb]) * 2u;
                bsrc_n[jb] = select_ptr(bval_n[jb], gath_b + off, zero);
                if constexpr (k == 5) st++;
            }
        };
        auto addr_all = [&]() __attribute__((always_inline)) {
            addr_piece(std::integral_constant<int, 0>{});
            addr_piece(std::integral_constant<int, 1>{});
            addr_piece(std::integral_constant<int, 2>{});
            addr_piece(std::integral_constant<int, 3>{});
            addr_piece(std::integral_constant<int, 4>{});
            addr_piece(std::integral_constant<int, 5>{});
        };
        auto addr_rotate = [&]() __attribute__((always_inline)) {
            asrc = asrc_n;
            aval = aval_n;
#pragma unroll
            for (int jb = 0; jb < 2; jb++) {
                bsrc[jb] = bsrc_n[jb];
                bval[jb] = bval_n[jb];
            }
        };
        auto issue_one = [&](auto idx_c, int sb) __attribute__((always_inline)) {
            constexpr int idx = decltype(idx_c)::value;
            if constexpr (idx < GA) {
                if constexpr (TM == 2) {
                    char* d = smem + sb + idx * A_PLANE + a_cb * 1024;
                    glds16(idx ? select_ptr(aval, asrc + rows_lo, asrc) : asrc, lds_addr(d));
                } else {
                    char* d = smem + sb + a_pl * A_PLANE + a_cb * 1024;
                    glds16(select_ptr(((int)(a_pl != 0) & (int)aval) != 0, asrc + rows_lo, asrc), lds_addr(d));
                }
            } else if constexpr (idx < G) {
                constexpr int k = idx - GA, jb = k >> 1, pl = k & 1;
                char* d = smem + sb + 2 * A_PLANE + pl * B_PLANE + (wave + 8 * jb) * 1024;
                glds16(pl ? select_ptr(bval[jb], bsrc[jb] + gath_lo, bsrc[jb]) : bsrc[jb], lds_addr(d));
            }
        };
        auto issue_all = [&](int sb) __attribute__((always_inline)) {
            issue_one(std::integral_constant<int, 0>{}, sb);
            issue_one(std::integral_constant<int, 1>{}, sb);
            issue_one(std::integral_constant<int, 2>{}, sb);
            issue_one(std::integral_constant<int, 3>{}, sb);
            issue_one(std::integral_constant<int, 4>{}, sb);
            issue_one(std::integral_constant<int, 5>{}, sb);
        };

        f32x16 acc[TM][TN];
#pragma unroll
        for (int mt = 0; mt < TM; mt++)
#pragma unroll
            for (int nt = 0; nt < TN; nt++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[mt][nt][r] = 0.f;

        s16x8 af[2][2][TM], bf[2][2][TN];   // [ks][hi, lo][tile]
#define WT_FRAG(dst, base, off)                                                                                        \
    {                                                                                                                  \
        const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)((base) + (off)[0]));                    \
        const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)((base) + (off)[1]));                    \
        dst = __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);                                               \
    }
#define WT_GROUP(ks, pp)                                                                                               \
    _Pragma("unroll") for (int mt = 0; mt < TM; mt++) _Pragma("unroll") for (int nt = 0; nt < TN; nt++)               \
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                                                         \
            __builtin_bit_cast(bf16x8, af[ks][(pp) == 0 ? 1 : 0][mt]), __builtin_bit_cast(bf16x8, bf[ks][(pp) == 1 ? 1 : 0][nt]), \
            acc[mt][nt], 0, 0, 0);
#define WT_STEP(ISSUE)                                                                                                 \
    {                                                                                                                  \
        const char* S = smem + cur;                                                                                    \
        _Pragma("unroll") for (int ks = 0; ks < 2; ks++)                                                               \
        {                                                                                                              \
            _Pragma("unroll") for (int mt = 0; mt < TM; mt++)                                                          \
            {                                                                                                          \
                WT_FRAG(af[ks][0][mt], S, aoffb[mt][ks]);                                                              \
                WT_FRAG(af[ks][1][mt], S + A_PLANE, aoffb[mt][ks]);                                                    \
            }                                                                                                          \
            _Pragma("unroll") for (int nt = 0; nt < TN; nt++)                                                          \
            {                                                                                                          \
                WT_FRAG(bf[ks][0][nt], S, boffb[nt][ks]);                                                              \
                WT_FRAG(bf[ks][1][nt], S + B_PLANE, boffb[nt][ks]);                                                    \
            }                                                                                                          \
        }                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        WT_GROUP(0, 0);                                                                                                \
        if (ISSUE) {                                                                                                   \
            issue_one(std::integral_constant<int, 0>{}, nxt);                                                          \
            addr_piece(std::integral_constant<int, 0>{});                                                              \
        }                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        WT_GROUP(0, 1);                                                                                                \
        if (ISSUE) {                                                                                                   \
            issue_one(std::integral_constant<int, 1>{}, nxt);                                                          \
            addr_piece(std::integral_constant<int, 1>{});                                                              \
        }                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        WT_GROUP(0, 2);                                                                                                \
        if (ISSUE) {                                                                                                   \
            issue_one(std::integral_constant<int, 2>{}, nxt);                                                          \
            addr_piece(std::integral_constant<int, 2>{});                                                              \
        }                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        WT_GROUP(1, 0);                                                                                                \
        if (ISSUE) {                                                                                                   \
            issue_one(std::integral_constant<int, 3>{}, nxt);                                                          \
            addr_piece(std::integral_constant<int, 3>{});                                                              \
        }                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        WT_GROUP(1, 1);                                                                                                \
        if (ISSUE) {                                                                                                   \
            issue_one(std::integral_constant<int, 4>{}, nxt);                                                          \
            addr_piece(std::integral_constant<int, 4>{});                                                              \
        }                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        WT_GROUP(1, 2);                                                                                                \
        if (ISSUE) {                                                                                                   \
            issue_one(std::integral_constant<int, 5>{}, nxt);                                                          \
            addr_piece(std::integral_constant<int, 5>{});                                                              \
            addr_rotate();                                                                                             \
        }                                                                                                              \
        cur = cur + STAGE == WT_STAGES * STAGE ? 0 : cur + STAGE;                                                      \
        nxt = nxt + STAGE == WT_STAGES * STAGE ? 0 : nxt + STAGE;                                                      \
    }
#define WT_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")\n\ts_barrier" ::: "memory")

        // ---- prologue: two steps in flight, the addresses of the third ready
        addr_all();
        addr_rotate();
        issue_all(0);
        if (s0 + 1 < s1) {
            addr_all();
            addr_rotate();
            issue_all(STAGE);
        }
        addr_all();
        addr_rotate();
        int cur = 0, nxt = 2 * STAGE;
        int s = s0;
        for (; s + 2 < s1; s++) {
            if constexpr (G == 6) WT_WAIT(6); else WT_WAIT(5);
            WT_STEP(true);
        }
        if (s + 1 < s1) {
            if constexpr (G == 6) WT_WAIT(6); else WT_WAIT(5);
            WT_STEP(false);
            s++;
        }
        WT_WAIT(0);
        WT_STEP(false);
#undef WT_STEP
#undef WT_GROUP
#undef WT_FRAG
#undef WT_WAIT

        // ---- partial sums of this segment -> dW (zeroed by the caller; the segments of a tile meet in float atomics)
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
        __syncthreads();   // the next segment's first copies overwrite stages this segment's last steps read
    }
}

template <int TM>
static int launch_wtile(WTileParams P, hipStream_t st, int max_blocks)
{
    constexpr int BM = 64 * TM;
    P.row_tiles = (P.Cr + BM - 1) / BM;
    P.col_tiles = (P.taps.n * P.Cc + 255) / 256;
    const long T = (long)P.row_tiles * P.col_tiles * P.total_steps;
    // one workgroup per CU; fewer when there is less than ~8 steps of work for each
    long blocks = max_blocks;
    if (T < blocks * 8) blocks = (T + 7) / 8;
    if (blocks < 1) blocks = 1;
    // decomposition: K slices such that (tiles x slices) fills whole rounds of the workgroups, >= 16 steps per slice; a shape
    // no slice count fits (efficiency < 0.9) keeps the stream-K ranges
    P.kslices = 0;
    P.slice_steps = P.total_steps;
    {
        const char* e = getenv("SDN_WTILE_MODE");   // lab switch: 0 = stream-K always, 1 = sliced when it fits (default)
        const long tiles = (long)P.row_tiles * P.col_tiles;
        if (!(e && e[0] == '0')) {
            double best = 0.0;
            for (int S = 1; S <= 64; S++) {
                const int ss = (P.total_steps + S - 1) / S;
                if (S > 1 && ss < 16) break;
                const long its = tiles * S;
                const double eff = (double)its / (double)(((its + blocks - 1) / blocks) * blocks);
                if (eff > best + 0.02) {
                    best = eff;
                    P.kslices = S;
                    P.slice_steps = ss;
                }
            }
            if (best < 0.9) P.kslices = 0;
        }
    }
    TimedLaunch timed(TIME_CONV_WGRAD, st, 2.0 * P.N * P.QH * P.QW * (double)P.taps.n * P.Cc * P.Cr);
    hipLaunchKernelGGL((k_wgrad_tile<TM>), dim3((unsigned)blocks), dim3(512), 0, st, P);
    return check_launch("k_wgrad_tile");
}

}  // namespace sdn

using namespace sdn;

SDN_API int sdn_conv_wgrad_tile(const void* rows_planes, long rows_stride, const void* gath_planes, long gath_stride, float* dw,
                                int N, int QH, int QW, int Cr, int GH, int GW, int Cc, int istride, int ntaps,
                                const int8_t* dy, const int8_t* dx, int pad_mode, sdnStream stream)
{
    if (!rows_planes || !gath_planes || !dw || !dy || !dx) return fail(SDN_EINVAL, "sdn_conv_wgrad_tile: null pointer");
    if (ntaps < 1 || ntaps > CONV_MAX_TAPS) return fail(SDN_EINVAL, "sdn_conv_wgrad_tile: ntaps %d not in 1..%d", ntaps, CONV_MAX_TAPS);
    if ((Cr & 15) || (Cc & 15)) return fail(SDN_EINVAL, "sdn_conv_wgrad_tile: channel counts must be padded to 16 (%d, %d)", Cr, Cc);
    if (N < 1 || QH < 1 || QW < 1 || GH < 1 || GW < 1 || istride < 1) return fail(SDN_EINVAL, "sdn_conv_wgrad_tile: bad geometry");
    const long Ptot = (long)N * QH * QW;
    if (Ptot >= (1 << 24)) return fail(SDN_EINVAL, "sdn_conv_wgrad_tile: %ld positions exceed 2^24", Ptot);
    if ((size_t)Ptot * Cr * 2 >= 0xffffff00u || (size_t)N * GH * GW * Cc * 2 >= 0xffffff00u)
        return fail(SDN_EINVAL, "sdn_conv_wgrad_tile: an operand plane must stay below 4 GiB");
    if (GH >= -WT_OUTSIDE / 2 || GW >= -WT_OUTSIDE / 2) return fail(SDN_EINVAL, "sdn_conv_wgrad_tile: image side above %d", -WT_OUTSIDE / 2);
    WTileParams P;
    P.rows = (const __bf16*)rows_planes; P.rows_stride = rows_stride;
    P.gath = (const __bf16*)gath_planes; P.gath_stride = gath_stride;
    P.dw = dw;
    P.N = N; P.QH = QH; P.QW = QW; P.Cr = Cr; P.GH = GH; P.GW = GW; P.Cc = Cc;
    P.istride = istride; P.pad_mode = pad_mode;
    P.total_steps = (int)((Ptot + 31) / 32);
    P.inv_q = 1.0f / (float)(QH * QW);
    P.inv_qw = 1.0f / (float)QW;
    P.taps.n = ntaps;
    for (int t = 0; t < ntaps; t++) {
        P.taps.dy[t] = dy[t];
        P.taps.dx[t] = dx[t];
    }
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    hipStream_t st = (hipStream_t)stream;
    if (Cr > 64) return launch_wtile<2>(P, st, cus);
    return launch_wtile<1>(P, st, cus);
}
