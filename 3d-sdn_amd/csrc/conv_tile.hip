// k_conv_tile: the implicit-GEMM convolution of conv_gemm.hip re-built around LDS-DMA and bf16 operand PLANES (r04).
//
// Reference: the Conv2d / ConvTranspose2d layers of GlobalGenerator, Encoder, NLayerDiscriminator
// (/root/reference/textural/models/networks.py:211-239, 286-308, 412-449), forward and data gradient (cuDNN there).
//
// What bounded k_conv_gemm: every workgroup re-split the fp32 activations
// it gathered (56 VALU per 24 MFMAs), each operand byte went through VGPRs on its way to LDS, weight fragments were
// fetched twice per workgroup, and 48 KB of L1 fills fed only 96 MFMAs.  Here
//   - activations arrive as two bf16 planes (hi, lo: x = hi + lo to ~2^-17), written ONCE per tensor by its producer
//     (conv_planes.hip: sdn_split_planes and the plane outputs of the norm / activation kernels), ReLU already applied;
//   - both operands are copied HBM/L2 -> LDS by `global_load_lds_dwordx4` (no VGPRs, no VALU, no ds_write): a wave
//     instruction moves 16 rows x 64 B; the LDS image is lane-linear, so the XOR swizzle that makes the fragment reads
//     conflict-free is applied to the SOURCE address (guide rule 21);
//   - a workgroup is 8 waves on a (128 TM) x (64 TN) output tile (TM, TN in {1, 2}: 256 x 128 for the wide layers), 32-deep K
//     steps, THREE LDS stages: the copies of step s + 2 are issued while step s is multiplied and are only waited for with a
//     counted `s_waitcnt vmcnt(G)` + raw `s_barrier` one step later, so a tile's worth of loads stays in flight across every
//     barrier (one barrier per step);
//   - per 32-deep step a 256 x 128 workgroup moves 48 KB through the L1 for 192 MFMAs (k_conv_gemm: 48 KB for 96).
// Numerics are k_conv_gemm's: three v_mfma_f32_32x32x16_bf16 products (lo*hi, hi*lo, hi*hi) accumulated in fp32.
// K order is channel-block-major (step = cb * ntaps + t), which needs Cip % 32 == 0; other layers stay on k_conv_gemm.
#include <cstdlib>
#include <type_traits>

#include "conv_common.h"
#include "conv_dma.h"
#include "conv_pack.h"
#include "sdn_common.h"

namespace sdn {

static __device__ __attribute__((aligned(256))) unsigned g_zero_page[64];  // what out-of-image / out-of-tile lanes copy from

struct TileTaps {
    int n;
    signed char dy[CONV_MAX_TAPS];
    signed char dx[CONV_MAX_TAPS];
};

struct ConvTileParams {
    const __bf16* in;        // planes [2 (hi, lo)][N, IH, IW, Cip]
    long plane_stride;       // elements between the planes
    float* out;              // [N, OH, OW, Cop]
    __bf16* out_planes;      // optional [2][N, OH, OW, Cop]: the stored value (after the activation), split
    long out_plane_stride;
    const __bf16* w;         // [w_rows][nsteps][2 (hi, lo)][32]   (sdn_conv_pack_weights_kmajor)
    const float* bias;
    double* stats;
    int N, IH, IW, Cip;
    int OH, OW, Cop;
    int QH, QW, istride, ostride, py, px;
    int nsteps, w_rows;
    int pad_mode, act, accumulate, planes_relu;
    int ntiles;
    // K-split tail (r04): position tiles `tail_from`.. of every image are cut into `ksplit` K slices, one workgroup each,
    // summed by float atomics into rows the launcher zeroed -- a 26 x 80 data-gradient grid is 8 full tiles + 32 rows per
    // image, and without the split those 32 rows cost a whole second round of the 256 CUs
    int tail_from, ksplit, nfull;
    TileTaps taps;
};


constexpr int TILE_STAGES = 3;
constexpr int TILE_OUTSIDE = -(1 << 14);

// V: 0 = the product kernel; other values are PROBE builds (-DSDN_TILE_PROBES, tools/tile_lab.py --probes; timing only):
// 1 no copies in the K loop (wrong results), 2 no MFMAs (wrong results)
template <int TM, int TN, int V = 0>
// 8 waves = 2 per SIMD, one workgroup per CU (144 KB of LDS): up to 256 VGPRs per wave.
// (Measured and dropped, r04: a ninth wave that touched the weight rows of step s + 8 -- the weights are a pure stream, every
// copy of them an L2 miss -- raised the L2 hit rate to 95 % and made the kernel 3 % SLOWER: the copies are not bound by HBM
// latency but by what one CU can keep in flight towards its L2, see DESIGN.md.)
__global__ __launch_bounds__(512, 2) void k_conv_tile(const ConvTileParams P)
{
    constexpr int WN = 2;
    constexpr int BM = 4 * TM * 32, BN = WN * TN * 32;
    constexpr int A_PLANE = BM * 64, B_PLANE = BN * 64;   // bytes: rows of 32 bf16
    constexpr int STAGE = 2 * A_PLANE + 2 * B_PLANE;
    constexpr int GB = TN == 2 ? 2 : 1;                    // weight copies per thread and step
    constexpr int G = 2 * TM + GB;                         // copies per thread and step
    // ONE shared object (a second one makes hipcc drain the DMA queue before every fragment read)
    __shared__ __attribute__((aligned(1024))) char smem[TILE_STAGES * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Q = P.QH * P.QW;
    const int mtiles = (Q + BM - 1) / BM;
    // XCD-aware tile order (as k_conv_gemm): hardware block b runs on XCD b % 8; every XCD gets a contiguous range of
    // position tiles with all channel tiles of each, so co-resident blocks share activation rows and weight columns in L2
    // The K slices of a split tail are the blocks after the nfull whole tiles, permuted among themselves: the whole tiles
    // are dispatched first and spread over all eight XCDs.
    const int ntiles = P.ntiles;
    const bool split = (int)blockIdx.x >= P.nfull;          // workgroup-uniform
    const unsigned b0 = split ? blockIdx.x - (unsigned)P.nfull : blockIdx.x;
    const unsigned nblk = split ? gridDim.x - (unsigned)P.nfull : (unsigned)P.nfull;
    const unsigned xcd = b0 & 7u, j = b0 >> 3;
    const unsigned q8 = nblk >> 3, r8 = nblk & 7u;
    const unsigned v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + j + (split ? (unsigned)P.nfull : 0u);
    int n, mtile, n0, kslice = 0;
    if (!split) {
        const int mt_global = (int)(v / (unsigned)ntiles);
        n = mt_global / P.tail_from;
        mtile = mt_global - n * P.tail_from;
        n0 = (int)(v % (unsigned)ntiles) * BN;
    } else {
        unsigned u = v - (unsigned)P.nfull;
        kslice = (int)(u % (unsigned)P.ksplit);
        u /= (unsigned)P.ksplit;
        n0 = (int)(u % (unsigned)ntiles) * BN;
        u /= (unsigned)ntiles;
        const unsigned tails = (unsigned)(mtiles - P.tail_from);
        mtile = P.tail_from + (int)(u % tails);
        n = (int)(u / tails);
    }
    const int m0 = mtile * BM;
    // this workgroup's K steps: all of them, or slice `kslice` of `ksplit`
    const int s_begin = split ? (int)((long)kslice * P.nsteps / P.ksplit) : 0;
    const int s_end = split ? (int)((long)(kslice + 1) * P.nsteps / P.ksplit) : P.nsteps;

    // ---- copy roles.  One wave instruction covers 16 rows x 64 B of one plane: lane l -> row (l >> 2), PHYSICAL 16-B chunk
    // (l & 3); the chunk it must fetch is the logical one, (l & 3) ^ ((row >> 2) & 3): the same for all rows of a thread
    // (they differ by multiples of 16... of 128), constant over the steps.
    const int srow = tid >> 2;                                  // 0..127
    const int lchunk = (lane & 3) ^ ((srow >> 2) & 3);
    int iy0[TM], ix0[TM];
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int q = m0 + srow + 128 * i;
        const bool ok = q < Q;
        const int qy = ok ? q / P.QW : 0, qx = ok ? q - qy * P.QW : 0;
        iy0[i] = ok ? qy * P.istride : TILE_OUTSIDE;
        ix0[i] = qx * P.istride;
    }
    // tap table in the lanes of one VGPR (v_readlane with a scalar index: no memory, no second shared object)
    int tapv = TILE_OUTSIDE * 256;
    if (lane < P.taps.n) tapv = ((int)P.taps.dy[lane] << 8) | ((int)P.taps.dx[lane] & 0xff);
    const int ntaps = P.taps.n;
    const char* in_n = (const char*)(P.in + (size_t)n * P.IH * P.IW * P.Cip);
    const long lo_bytes = P.plane_stride * 2;
    const char* zero = (const char*)g_zero_page;
    const int ih2 = 2 * P.IH - 2, iw2 = 2 * P.IW - 2;
    // weights: row n0 + brow, 128 B per (row, step): hi 64 B, lo 64 B
    const int brow = TN == 2 ? srow : (srow & 63);
    const int bplane = TN == 2 ? 0 : (wave >> 2);
    const char* wsrc = (const char*)P.w + ((size_t)(n0 + brow) * P.nsteps + s_begin) * 128 + bplane * 64 + lchunk * 16;

    const int nsteps = s_end - s_begin;
    // (tap, channel block) of the step whose addresses are computed next: wave-uniform
    int st_cb = s_begin / P.taps.n, st_t = s_begin - st_cb * P.taps.n;

    // `asrc` / `alo`: this thread's A sources (hi / lo plane; the zero page when outside) of the step whose copies are issued
    // next.  The set of the step after that is computed in pieces (addr_piece<0..5>) placed between the MFMA groups of a
    // step, so that the address arithmetic issues in the matrix pipe's shadow (branch-free: the K loop must stay ONE basic
    // block for the pinned schedule).
    const char* asrc[TM];
    const char* alo[TM];
    const char* asrc_n[TM];
    const char* alo_n[TM];
    const bool reflect = P.pad_mode != 0;
    int a_iy[TM], a_ix[TM];
    unsigned a_off[TM];
    bool a_ok[TM];
    auto addr_piece = [&](auto k_c) __attribute__((always_inline)) {
        constexpr int k = decltype(k_c)::value;
        if constexpr (k == 0) {
            const int tp = __builtin_amdgcn_readlane(tapv, st_t);
            const int dy = tp >> 8, dx = (int)(signed char)(tp & 0xff);
#pragma unroll
            for (int i = 0; i < TM; i++) {
                a_iy[i] = iy0[i] + dy;
                a_ix[i] = ix0[i] + dx;
            }
        } else if constexpr (k == 1) {
#pragma unroll
            for (int i = 0; i < TM; i++) {
                int ry = max(a_iy[i], -a_iy[i]), rx = max(a_ix[i], -a_ix[i]);   // ReflectionPad2d: |v|, mirrored at the far edge
                ry = min(ry, ih2 - ry);
                rx = min(rx, iw2 - rx);
                a_iy[i] = reflect ? ry : a_iy[i];
                a_ix[i] = reflect ? rx : a_ix[i];
            }
        } else if constexpr (k == 2) {
#pragma unroll
            for (int i = 0; i < TM; i++) {
                a_ok[i] = ((int)((unsigned)a_iy[i] < (unsigned)P.IH) & (int)((unsigned)a_ix[i] < (unsigned)P.IW)) != 0;
                a_off[i] = (unsigned)((a_iy[i] * P.IW + a_ix[i]) * P.Cip + st_cb * 32 + lchunk * 8) * 2u;
            }
        } else if constexpr (k == 3) {
#pragma unroll
            for (int i = 0; i < TM; i++) asrc_n[i] = select_ptr(a_ok[i], in_n + a_off[i], zero);
        } else if constexpr (k == 4) {
#pragma unroll
            for (int i = 0; i < TM; i++) alo_n[i] = select_ptr(a_ok[i], in_n + a_off[i] + lo_bytes, zero);
        } else {
            const int t1 = st_t + 1;
            const bool wrap = t1 >= ntaps;
            st_t = wrap ? 0 : t1;
            st_cb = wrap ? st_cb + 1 : st_cb;
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
#pragma unroll
        for (int i = 0; i < TM; i++) {
            asrc[i] = asrc_n[i];
            alo[i] = alo_n[i];
        }
    };
    // copy number `idx` (0 .. G-1) of a step into the stage at byte offset `sb`; weights of step `bs`
    auto issue_one = [&](auto idx_c, int sb, int bs) __attribute__((always_inline)) {
        constexpr int idx = decltype(idx_c)::value;
        if constexpr (idx < 2 * TM) {
            constexpr int i = idx >> 1, lo = idx & 1;
            char* d = smem + sb + lo * A_PLANE + (128 * i + 16 * wave) * 64;
            glds16((lo ? alo[i] : asrc[i]), lds_addr(d));
        } else if constexpr (idx < G) {
            constexpr int lo = idx - 2 * TM;
            const char* wp = wsrc + (size_t)bs * 128 + lo * 64;
            if constexpr (TN == 2) {
                char* d = smem + sb + 2 * A_PLANE + lo * B_PLANE + 16 * wave * 64;
                glds16(wp, lds_addr(d));
            } else {
                char* d = smem + sb + 2 * A_PLANE + bplane * B_PLANE + 16 * (wave & 3) * 64;
                glds16(wp, lds_addr(d));
            }
        }
    };
    auto issue_all = [&](int sb, int bs) __attribute__((always_inline)) {
        issue_one(std::integral_constant<int, 0>{}, sb, bs);
        issue_one(std::integral_constant<int, 1>{}, sb, bs);
        issue_one(std::integral_constant<int, 2>{}, sb, bs);
        issue_one(std::integral_constant<int, 3>{}, sb, bs);
        issue_one(std::integral_constant<int, 4>{}, sb, bs);
        issue_one(std::integral_constant<int, 5>{}, sb, bs);
    };

    const int wm0 = (wave >> 1) * TM * 32, wn0 = (wave & 1) * TN * 32;
    const int fr = lane & 31, fkh = lane >> 5;
    // byte offsets of this lane's fragment rows inside a plane: row * 64 + (logical chunk ^ swizzle) * 16
    int aoffb[TM][2], boffb[TN][2];
#pragma unroll
    for (int mt = 0; mt < TM; mt++)
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            const int row = wm0 + mt * 32 + fr;
            aoffb[mt][ks] = row * 64 + (((2 * ks + fkh) ^ ((row >> 2) & 3)) << 4);
        }
#pragma unroll
    for (int nt = 0; nt < TN; nt++)
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            const int row = wn0 + nt * 32 + fr;
            boffb[nt][ks] = 2 * A_PLANE + row * 64 + (((2 * ks + fkh) ^ ((row >> 2) & 3)) << 4);
        }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int mt = 0; mt < TM; mt++)
#pragma unroll
        for (int nt = 0; nt < TN; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[mt][nt][r] = 0.f;

    // ---- K loop.  Step s multiplies stage s % 3.  Its two 16-deep halves use fragment sets F0 / F1:
    //     top of step s:   read F1 <- (stage s, k half 1);  MFMAs of half 0 on F0, with the address arithmetic of the copies
    //                      issued below in their gaps;
    //     middle:          counted wait + barrier: stage s + 1 has landed for everybody, and everybody has issued its last
    //                      reads of stage s;  read F0 <- (stage s + 1, k half 0);
    //     second half:     MFMAs of half 1 on F1, with the copies of step s + 3 (into stage s % 3, free since the barrier)
    //                      spread between them -- two 1 KiB copies per MFMA group, about the rate at which the texture-address
    //                      unit retires the eight waves' copies.
    // So every fragment read has half a step of MFMAs in front of its first use, the barrier sits where the matrix pipe still
    // has a group queued, and a copy has two full steps to land.  sched_barrier pins the order inside the basic block.
    bf16x8 af[2][2][TM], bf[2][2][TN];   // [k half][hi, lo][tile]
#define TILE_MFMA(ks, pp, mt, nt)                                                                                      \
    if constexpr (V != 2)                                                                                              \
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][(pp) == 0 ? 1 : 0][mt], bf[ks][(pp) == 1 ? 1 : 0][nt], \
                                                              acc[mt][nt], 0, 0, 0);                                   \
    else                                                                                                               \
        asm volatile("" : "+v"(acc[mt][nt]) : "v"(af[ks][(pp) == 0 ? 1 : 0][mt]), "v"(bf[ks][(pp) == 1 ? 1 : 0][nt]));
#define TILE_GROUP(ks, pp)                                                                                             \
    _Pragma("unroll") for (int mt = 0; mt < TM; mt++) _Pragma("unroll") for (int nt = 0; nt < TN; nt++)               \
        TILE_MFMA(ks, pp, mt, nt)
#define TILE_READ(ks, sb)                                                                                              \
    {                                                                                                                  \
        const char* S = smem + (sb);                                                                                   \
        _Pragma("unroll") for (int mt = 0; mt < TM; mt++)                                                              \
        {                                                                                                              \
            af[ks][0][mt] = *reinterpret_cast<const bf16x8*>(S + aoffb[mt][ks]);                                       \
            af[ks][1][mt] = *reinterpret_cast<const bf16x8*>(S + A_PLANE + aoffb[mt][ks]);                             \
        }                                                                                                              \
        _Pragma("unroll") for (int nt = 0; nt < TN; nt++)                                                              \
        {                                                                                                              \
            bf[ks][0][nt] = *reinterpret_cast<const bf16x8*>(S + boffb[nt][ks]);                                       \
            bf[ks][1][nt] = *reinterpret_cast<const bf16x8*>(S + B_PLANE + boffb[nt][ks]);                             \
        }                                                                                                              \
    }
#define TILE_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")\n\ts_barrier" ::: "memory")
    // ISSUE: the copies of step s + 3 are issued (and their addresses computed); MID: 1 = counted wait (the copies of step
    // s + 2 may still fly), 0 = wait for everything, -1 = last step (nothing follows)
#define TILE_STEP(ISSUE, MID, bs)                                                                                      \
    {                                                                                                                  \
        TILE_READ(1, cur);                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        TILE_GROUP(0, 0);                                                                                              \
        if (ISSUE) {                                                                                                   \
            addr_piece(std::integral_constant<int, 0>{});                                                              \
            addr_piece(std::integral_constant<int, 1>{});                                                              \
        }                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        TILE_GROUP(0, 1);                                                                                              \
        if (ISSUE) {                                                                                                   \
            addr_piece(std::integral_constant<int, 2>{});                                                              \
            addr_piece(std::integral_constant<int, 3>{});                                                              \
        }                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        TILE_GROUP(0, 2);                                                                                              \
        if (ISSUE) {                                                                                                   \
            addr_piece(std::integral_constant<int, 4>{});                                                              \
            addr_piece(std::integral_constant<int, 5>{});                                                              \
            addr_rotate();                                                                                             \
        }                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        if constexpr ((MID) >= 0) {                                                                                    \
            if constexpr ((MID) == 1) {                                                                                \
                if constexpr (G == 6) { TILE_WAIT(6); } else { TILE_WAIT(5); }                                         \
            } else {                                                                                                   \
                TILE_WAIT(0);                                                                                          \
            }                                                                                                          \
            TILE_READ(0, nx1);                                                                                         \
        }                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        TILE_GROUP(1, 0);                                                                                              \
        if (ISSUE) {                                                                                                   \
            if constexpr (V != 1) {                                                                                    \
                issue_one(std::integral_constant<int, 0>{}, cur, bs);                                                  \
                issue_one(std::integral_constant<int, 1>{}, cur, bs);                                                  \
            }                                                                                                          \
        }                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        TILE_GROUP(1, 1);                                                                                              \
        if (ISSUE) {                                                                                                   \
            if constexpr (V != 1) {                                                                                    \
                issue_one(std::integral_constant<int, 2>{}, cur, bs);                                                  \
                issue_one(std::integral_constant<int, 3>{}, cur, bs);                                                  \
            }                                                                                                          \
        }                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        TILE_GROUP(1, 2);                                                                                              \
        if (ISSUE) {                                                                                                   \
            if constexpr (V != 1) {                                                                                    \
                issue_one(std::integral_constant<int, 4>{}, cur, bs);                                                  \
                issue_one(std::integral_constant<int, 5>{}, cur, bs);                                                  \
            }                                                                                                          \
        }                                                                                                              \
        cur = nx1;                                                                                                     \
        nx1 = nx1 + STAGE == TILE_STAGES * STAGE ? 0 : nx1 + STAGE;                                                    \
    }

    // ---- prologue: three steps in flight, F0 of step 0 read
    addr_all();
    addr_rotate();
    issue_all(0, 0);
    if (nsteps > 1) {
        addr_all();
        addr_rotate();
        issue_all(STAGE, 1);
    }
    if (nsteps > 2) {
        addr_all();
        addr_rotate();
        issue_all(2 * STAGE, 2);
    }
    if (nsteps > 2) {
        if constexpr (G == 6) asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(10)\n\ts_barrier" ::: "memory");
    } else {
        TILE_WAIT(0);
    }
    int cur = 0, nx1 = STAGE;   // byte offsets of the stage of step s / of step s + 1
    TILE_READ(0, 0);
    int s = 0;
    for (; s + 3 < nsteps; s++) TILE_STEP(true, 1, s + 3);
    for (; s + 1 < nsteps; s++) TILE_STEP(false, 0, 0);
    TILE_STEP(false, -1, 0);
#undef TILE_STEP
#undef TILE_READ
#undef TILE_WAIT
#undef TILE_MFMA
#undef TILE_GROUP
    __syncthreads();   // every wave is done with the stages: the epilogue reuses the LDS

    // ---- epilogue (k_conv_gemm's, on a BM-row tile): bias, activation, InstanceNorm statistics, channel-contiguous stores
    int* s_outpix = reinterpret_cast<int*>(smem);
    float* red = reinterpret_cast<float*>(smem + 4096);
    if (tid < BM) {
        const int q = m0 + tid;
        int o = -1;
        if (q < Q) {
            const int qy = q / P.QW, qx = q - qy * P.QW;
            o = (n * P.OH + qy * P.ostride + P.py) * P.OW + qx * P.ostride + P.px;
        }
        s_outpix[tid] = o;
    }
    __syncthreads();
    const int col = lane & 31;
#pragma unroll
    for (int nt = 0; nt < TN; nt++) {
        const int co = n0 + wn0 + nt * 32 + col;
        const bool co_ok = co < P.Cop;
        const float bias = (co_ok && P.bias && kslice == 0) ? P.bias[co] : 0.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int mt = 0; mt < TM; mt++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = wm0 + mt * 32 + mfma_row(r, lane);
                const int o = s_outpix[row];
                if (o < 0 || !co_ok) continue;
                float x = acc[mt][nt][r] + bias;
                if (split) {    // a K slice: no activation / statistics / planes (the launcher refuses them), rows zeroed
                    unsafeAtomicAdd(P.out + (size_t)o * P.Cop + co, x);
                    continue;
                }
                s1 += x;
                s2 += x * x;
                if (P.act == 1)
                    x = x > 0.f ? x : 0.2f * x;
                else if (P.act == 2)
                    x = tanhf(x);
                float* dst = P.out + (size_t)o * P.Cop + co;
                if (P.accumulate) x += *dst;
                *dst = x;
                if (P.out_planes) {
                    const float y = P.planes_relu ? fmaxf(x, 0.f) : x;
                    const __bf16 h = (__bf16)y;
                    P.out_planes[(size_t)o * P.Cop + co] = h;
                    P.out_planes[P.out_plane_stride + (size_t)o * P.Cop + co] = (__bf16)(y - (float)h);
                }
            }
        }
        if (P.stats) {
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (lane < 32) {
                const int slot = ((wave >> 1) * BN + wn0 + nt * 32 + col) * 2;
                red[slot] = s1;
                red[slot + 1] = s2;
            }
        }
    }
    if (P.stats) {
        __syncthreads();
        if (tid < BN) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                s1 += red[(w * BN + tid) * 2];
                s2 += red[(w * BN + tid) * 2 + 1];
            }
            const int co = n0 + tid;
            if (co < P.Cop) {
                const int slot = mtile & (STAT_SLOTS - 1);
                double* st = P.stats + (((size_t)n * STAT_SLOTS + slot) * P.Cop + co) * 2;
                unsafeAtomicAdd(st, (double)s1);
                unsafeAtomicAdd(st + 1, (double)s2);
            }
        }
    }
}

// ---- Measured and removed again, r06 (VERDICT r05 #2b; the kernel is in the git history as k_conv_tile_pc, commit "Probe: k_conv_tile_pc"):
// a PRODUCER / CONSUMER split of this tile -- waves 0..7 only read fragments and issue MFMAs behind a bare s_barrier (their K loop
// compiles to 16 ds_read_b128 + 24 MFMAs per step and nothing else), 2 or 4 extra waves own the LDS-DMA copies and their address
// arithmetic under the same three-stage protocol (copies of step s + 3 behind the barrier of step s, counted vmcnt in front of
// the next).  Parity green (tests/test_gpu_conv_tile.py), and SLOWER on every layer: residual data gradient 0.42 ms (product) ->
// 0.54-0.56 (2 producer waves) / 0.47 (4); 512 -> 1024 s2 forward 0.19 -> 0.26-0.27 / 0.22 (profiles/r06j_tile_pc_*.log).  With
// the copies' issue slots out of the MFMA waves the step is paced by the producers' serial address arithmetic and by the
// barrier they share with the consumers; interleaving the copies between each wave's own MFMA groups, as above, hides more.

// ---- weights, K-major: packed[r][step][part][k % 32] for step = cb * ntaps + t, c = cb * 32 + k % 32
__global__ __launch_bounds__(256) void k_pack_weights_kmajor(const float* __restrict__ w, int R, int C, long sr, long sc,
                                                             const int* __restrict__ tapidx, int ntaps, int Ccp, int rows,
                                                             __bf16* __restrict__ packed)
{
    pack_weights_kmajor_group8((long)blockIdx.x * 256 + threadIdx.x, w, R, C, sr, sc, tapidx, ntaps, Ccp, rows, packed);
}

template <int TM, int TN>
static int launch_tile(ConvTileParams P, hipStream_t st)
{
    constexpr int BM = 4 * TM * 32, BN = 2 * TN * 32;
    const int Q = P.QH * P.QW;
    P.ntiles = (P.Cop + BN - 1) / BN;
    if (P.w_rows < P.ntiles * BN) return fail(SDN_EINVAL, "sdn_conv_tile: weight rows %d < %d", P.w_rows, P.ntiles * BN);
    const int mtiles = (Q + BM - 1) / BM;
    long tiles = (long)mtiles * P.N * P.ntiles;
    P.tail_from = mtiles;
    P.nfull = (int)tiles;
    if (P.ksplit > 1) {
        // the last position tile of each image in `ksplit` K slices (see ConvTileParams): its output rows are contiguous per
        // image (checked by the caller below), zeroed here, then summed by atomics
        if (P.nsteps < 2 * P.ksplit) return fail(SDN_EINVAL, "sdn_conv_tile: %d K steps cannot be split %d ways", P.nsteps, P.ksplit);
        P.tail_from = mtiles - 1;
        P.nfull = P.tail_from * P.N * P.ntiles;
        tiles = (long)P.nfull + (long)P.N * P.ntiles * P.ksplit;
        const size_t row0 = (size_t)P.tail_from * BM;
        if (hipMemset2DAsync(P.out + row0 * P.Cop, (size_t)Q * P.Cop * 4, 0, (size_t)(Q - row0) * P.Cop * 4, (size_t)P.N, st) != hipSuccess)
            return fail(SDN_ELAUNCH, "sdn_conv_tile: memset of the split rows failed");
    } else {
        P.ksplit = 1;
    }
    TimedLaunch timed(TIME_CONV_GEMM, st, 2.0 * P.N * Q * (double)P.taps.n * P.Cip * P.Cop);
#ifdef SDN_TILE_PROBES
    {
        const char* e = getenv("SDN_TILE_VARIANT");
        const int var = e ? atoi(e) : 0;
        if constexpr (TM == 2 && TN == 2) {
            switch (var) {
            case 1: hipLaunchKernelGGL((k_conv_tile<TM, TN, 1>), dim3((unsigned)tiles), dim3(512), 0, st, P); return check_launch("k_conv_tile");
            case 2: hipLaunchKernelGGL((k_conv_tile<TM, TN, 2>), dim3((unsigned)tiles), dim3(512), 0, st, P); return check_launch("k_conv_tile");
            default: break;
            }
        }
    }
#endif
    hipLaunchKernelGGL((k_conv_tile<TM, TN>), dim3((unsigned)tiles), dim3(512), 0, st, P);
    return check_launch("k_conv_tile");
}

}  // namespace sdn

using namespace sdn;

SDN_API int sdn_conv_pack_weights_kmajor(const float* w, int R, int C, long sr, long sc, const int32_t* tapidx, int ntaps,
                                         int Ccp, int rows, void* packed, sdnStream stream)
{
    if (!w || !tapidx || !packed || (Ccp & 31) || rows < R || (rows & 63) || Ccp < C || ntaps < 1)
        return fail(SDN_EINVAL, "sdn_conv_pack_weights_kmajor: bad argument");
    hipLaunchKernelGGL(k_pack_weights_kmajor, dim3(cdiv((long)rows * ntaps * Ccp / 8, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       R, C, sr, sc, tapidx, ntaps, Ccp, rows, (__bf16*)packed);
    return check_launch("k_pack_weights_kmajor");
}

SDN_API int sdn_conv_tile(const void* in_planes, long plane_stride, int N, int IH, int IW, int Cip, float* out,
                          void* out_planes, long out_plane_stride, int planes_relu, int OH, int OW, int Cop, int QH, int QW,
                          int istride, int ostride, int py, int px, int ntaps, const int8_t* dy, const int8_t* dx,
                          int pad_mode, const void* w_kmajor, int w_rows, const float* bias, int act, double* stats,
                          int accumulate, int ksplit, sdnStream stream)
{
    if (!in_planes || !out || !w_kmajor || !dy || !dx) return fail(SDN_EINVAL, "sdn_conv_tile: null pointer");
    if (ksplit > 1 && (act || stats || out_planes || accumulate || ostride != 1 || py || px || QH != OH || QW != OW || ksplit > 64))
        return fail(SDN_EINVAL, "sdn_conv_tile: a K-split tail needs a plain dense output (no activation, statistics, planes, "
                                "accumulation; the launch grid = the output grid) and ksplit <= 64");
    if (ntaps < 1 || ntaps > CONV_MAX_TAPS) return fail(SDN_EINVAL, "sdn_conv_tile: ntaps %d not in 1..%d", ntaps, CONV_MAX_TAPS);
    if ((Cip & 31) || (Cop & 15)) return fail(SDN_EINVAL, "sdn_conv_tile: Cip %% 32, Cop %% 16 (%d, %d)", Cip, Cop);
    if (N < 1 || QH < 1 || QW < 1 || istride < 1 || ostride < 1) return fail(SDN_EINVAL, "sdn_conv_tile: bad geometry");
    if ((QH - 1) * ostride + py >= OH || (QW - 1) * ostride + px >= OW) return fail(SDN_EINVAL, "sdn_conv_tile: output grid exceeds the output tensor");
    if ((size_t)IH * IW * Cip * 2 >= 0x7fffff00u) return fail(SDN_EINVAL, "sdn_conv_tile: one input image plane must stay below 2 GiB");
    if (IH >= -TILE_OUTSIDE / 2 || IW >= -TILE_OUTSIDE / 2) return fail(SDN_EINVAL, "sdn_conv_tile: image side above %d", -TILE_OUTSIDE / 2);
    ConvTileParams P;
    P.in = (const __bf16*)in_planes; P.plane_stride = plane_stride; P.out = out;
    P.out_planes = (__bf16*)out_planes; P.out_plane_stride = out_plane_stride; P.planes_relu = planes_relu;
    P.w = (const __bf16*)w_kmajor; P.bias = bias; P.stats = stats;
    P.N = N; P.IH = IH; P.IW = IW; P.Cip = Cip; P.OH = OH; P.OW = OW; P.Cop = Cop;
    P.QH = QH; P.QW = QW; P.istride = istride; P.ostride = ostride; P.py = py; P.px = px;
    P.nsteps = ntaps * (Cip >> 5); P.w_rows = w_rows;
    P.pad_mode = pad_mode; P.act = act; P.accumulate = accumulate; P.ksplit = ksplit;
    P.taps.n = ntaps;
    for (int t = 0; t < ntaps; t++) {
        P.taps.dy[t] = dy[t];
        P.taps.dx[t] = dx[t];
    }
    hipStream_t st = (hipStream_t)stream;
    if (Cop > 64) return launch_tile<2, 2>(P, st);
    return launch_tile<2, 1>(P, st);
}
