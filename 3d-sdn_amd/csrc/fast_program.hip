// Launch lists (include/sdn_hip.h, "launch lists"): the launches of one conv-chain pass, planned once by the host and
// replayed with one call.  The reference has no counterpart as code -- its passes are PyTorch walking nn.Sequential
// (/root/reference/textural/models/networks.py:238-239) and the autograd graph (textural/train.py:88-95), one Python /
// dispatcher round trip per kernel; a GAN step here is ~1800 launches, and issuing them from Python cost as much wall time
// as the kernels take.  sdn_program_run is a loop over records that calls this library's own launchers.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "conv_pack.h"
#include "sdn_common.h"

namespace sdn {

// ---- runs of pack / unpack / small-copy records as ONE launch (r04).  A GAN step re-packs ~180 weight
// tensors and unpacks ~70 gradients, most of them a few KB: 250 launches of 5-10 us each.  sdn_program_run gathers
// consecutive records of these kinds on one stream into a descriptor table passed BY VALUE (kernel arguments: no device
// table to build or upload) and k_weights_multi finds its tensor from blockIdx; the per-element bodies are the per-tensor
// kernels' own (conv_pack.h), so the bytes written are the same.  The timed mode (SDN_PROFILE) keeps one launch per record.
constexpr int MULTI_PER_THREAD = 4;   // elements per thread: the tensor look-up is paid once per 1024 elements
constexpr int MULTI_GROUPS = 2;       // packs (r05): groups of 8 columns per thread -> 4096 elements per block
__host__ __device__ constexpr long multi_block_elems(int kind) { return kind <= 1 ? 256L * 8 * MULTI_GROUPS : 256L * MULTI_PER_THREAD; }
constexpr int MULTI_MAX = 44;   // 44 x 80 B + header: well inside the 4 KB kernel-argument limit
struct WDesc {
    const void* src;
    void* dst;
    const int* tapidx;
    long sr, sc;         // kind 3: sr = bytes to copy
    int R, C, ntaps, Ccp, Kp, rows, kind, accumulate;   // kind 0 pack (fragment order), 1 pack K-major, 2 unpack, 3 copy, 4 unpack by rows
    unsigned first_block, pad;
};
struct WMulti {
    int n, pad;
    WDesc d[MULTI_MAX];
};

__global__ __launch_bounds__(256) void k_weights_multi(const WMulti M)
{
    // the tensor of this block: binary search over first_block (block-uniform: scalar loads from the argument segment)
    int lo = 0, hi = M.n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (blockIdx.x >= M.d[mid].first_block)
            lo = mid;
        else
            hi = mid - 1;
    }
    const WDesc D = M.d[lo];
    if (D.kind <= 1) {   // weight packs: a thread owns groups of eight consecutive columns (conv_pack.h)
        const long g0 = (long)(blockIdx.x - D.first_block) * (256 * MULTI_GROUPS) + threadIdx.x;
#pragma unroll
        for (int u = 0; u < MULTI_GROUPS; u++) {
            const long g = g0 + 256 * u;
            if (D.kind == 0)
                pack_weights_group8(g, (const float*)D.src, D.R, D.C, D.sr, D.sc, D.tapidx, D.ntaps, D.Ccp, D.Kp, D.rows, (__bf16*)D.dst);
            else
                pack_weights_kmajor_group8(g, (const float*)D.src, D.R, D.C, D.sr, D.sc, D.tapidx, D.ntaps, D.Ccp, D.rows, (__bf16*)D.dst);
        }
        return;
    }
    if (D.kind == 4) {   // gradient unpack as a transpose through LDS (taps innermost in the parameter layout: conv_pack.h)
        __shared__ float tile[UNPACK_CB * (UNPACK_MAX_SC + 1)];
        __shared__ int inv[UNPACK_MAX_SC];
        unpack_grad_rows((long)(blockIdx.x - D.first_block), (const float*)D.src, D.R, D.C, D.sr, D.sc, D.tapidx, D.ntaps, D.Ccp,
                         (float*)D.dst, D.accumulate, tile, inv);
        return;
    }
    if (D.kind == 2) {   // gradient unpack: groups of four columns, 1024 elements per block as before
        unpack_grad_group4((long)(blockIdx.x - D.first_block) * 256 + threadIdx.x, (const float*)D.src, D.R, D.C, D.sr, D.sc,
                           D.tapidx, D.ntaps, D.Ccp, (float*)D.dst, D.accumulate);
        return;
    }
    const long base = (long)(blockIdx.x - D.first_block) * (256 * MULTI_PER_THREAD) + threadIdx.x;
#pragma unroll
    for (int u = 0; u < MULTI_PER_THREAD; u++) {
        const long i = base + 256 * u;
        if (i < (D.sr >> 2))
            ((float*)D.dst)[i] = ((const float*)D.src)[i];
    }
}

static bool multi_kind(const sdn_op& o)
{
    return o.code == SDN_OP_PACK_WEIGHTS || o.code == SDN_OP_PACK_WEIGHTS_KMAJOR || o.code == SDN_OP_UNPACK_GRAD ||
           (o.code == SDN_OP_COPY && o.l[0] > 0 && o.l[0] <= 65536 && (o.l[0] & 3) == 0);
}

typedef __attribute__((ext_vector_type(4))) float f32x4;

// (out may be a or b: the planner adds a gradient in place -- no __restrict__)
__global__ __launch_bounds__(256) void k_add(float* out, const float* a, const float* b, long n4)
{
    const long stride = (long)gridDim.x * 256;
    for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < n4; k += stride)
        reinterpret_cast<f32x4*>(out)[k] = reinterpret_cast<const f32x4*>(a)[k] + reinterpret_cast<const f32x4*>(b)[k];
}

// out[c] = sum_rows g[row, c] in a fixed order: one workgroup per 4 channels, every thread a fixed row subset (fp64
// partials), then a fixed-shape tree.  Only the deterministic mode uses it (the atomic path lives in sdn_act_bwd).
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ g, long rows, int pitch, int C,
                                                float* __restrict__ out)
{
    __shared__ double red[256][4];
    const int c0 = blockIdx.x * 4;
    double s[4] = {0, 0, 0, 0};
    for (long r = threadIdx.x; r < rows; r += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(g + r * pitch + c0);
#pragma unroll
        for (int e = 0; e < 4; e++) s[e] += (double)v[e];
    }
#pragma unroll
    for (int e = 0; e < 4; e++) red[threadIdx.x][e] = s[e];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
#pragma unroll
            for (int e = 0; e < 4; e++) red[threadIdx.x][e] += red[threadIdx.x + o][e];
        __syncthreads();
    }
    if (threadIdx.x < 4 && c0 + (int)threadIdx.x < C) out[c0 + threadIdx.x] = (float)red[0][threadIdx.x];
}

}  // namespace sdn

using namespace sdn;

struct sdn_program {
    std::vector<sdn_op> ops;
    std::vector<int8_t> taps;
    int n_slots;
};

namespace {

// how many buf[] entries each record kind reads (for validation) and whether it carries a tap pair
struct OpInfo {
    int nbuf;
    bool taps;
    int ntaps_arg;  // index into i[] of the tap count
};
const OpInfo kInfo[SDN_OP_CODES] = {
    {0, false, 0},   // (0 unused)
    {6, true, 13},   // CONV_GEMM
    {4, false, 0},   // CONV_NARROW_FWD
    {8, false, 0},   // IN_APPLY
    {5, false, 0},   // IN_BWD
    {4, false, 0},   // ACT_BWD
    {2, false, 0},   // REFLECT_FOLD
    {4, true, 8},    // CONV_WGRAD
    {3, true, 8},    // CONV_WGRAD_NARROW
    {3, false, 0},   // PACK_WEIGHTS
    {3, false, 0},   // UNPACK_GRAD
    {1, false, 0},   // MEMSET
    {2, false, 0},   // COPY
    {3, false, 0},   // ADD
    {2, false, 0},   // COLSUM
    {0, false, 0},   // FORK
    {0, false, 0},   // JOIN
    {2, false, 0},   // SPLIT_PLANES
    {3, false, 0},   // PACK_WEIGHTS_KMAJOR
    {6, true, 14},   // CONV_TILE
    {5, true, 7},    // CONV_HALO
    {3, true, 8},    // CONV_WGRAD_TILE
    {8, false, 0},   // CONV_GEMM_PHASES (its tap lists are checked apart: one (dy, dx) pair per phase)
    {5, false, 0},   // CONV_HEAD_MFMA
    {3, true, 8},    // CONV_WGRAD_HEAD
};

// two events per calling thread and device: FORK / JOIN record one and make the other stream wait for it; a later record
// of the same event does not disturb a wait already enqueued.  Thread-local: nn.DataParallel replays programs from one
// Python thread per GPU.
struct EventPair {
    int device = -1;
    hipEvent_t fork = nullptr, join = nullptr;
};
thread_local std::vector<EventPair> t_events;

int events_for_current_device(EventPair** out)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail(SDN_ELAUNCH, "sdn_program_run: hipGetDevice failed");
    for (auto& e : t_events)
        if (e.device == dev) {
            *out = &e;
            return SDN_OK;
        }
    EventPair p;
    p.device = dev;
    if (hipEventCreateWithFlags(&p.fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&p.join, hipEventDisableTiming) != hipSuccess)
        return fail(SDN_ELAUNCH, "sdn_program_run: cannot create events");
    t_events.push_back(p);
    *out = &t_events.back();
    return SDN_OK;
}

inline int hip_ok(hipError_t e, const char* what)
{
    return e == hipSuccess ? SDN_OK : fail(SDN_ELAUNCH, "sdn_program_run: %s: %s", what, hipGetErrorString(e));
}

}  // namespace

SDN_API int sdn_program_create(const sdn_op* ops, int n_ops, const int8_t* taps, size_t tap_bytes, int n_slots,
                               sdn_program** out)
{
    if (!ops || n_ops < 1 || n_slots < 1 || !out || (tap_bytes && !taps))
        return fail(SDN_EINVAL, "sdn_program_create: bad arguments");
    for (int k = 0; k < n_ops; k++) {
        const sdn_op& o = ops[k];
        if (o.code < 1 || o.code >= SDN_OP_CODES) return fail(SDN_EINVAL, "sdn_program_create: record %d: unknown code %d", k, o.code);
        if (o.stream != 0 && o.stream != 1) return fail(SDN_EINVAL, "sdn_program_create: record %d: stream %d", k, o.stream);
        const OpInfo& inf = kInfo[o.code];
        for (int b = 0; b < inf.nbuf; b++)
            if (o.buf[b] < -1 || o.buf[b] >= n_slots)
                return fail(SDN_EINVAL, "sdn_program_create: record %d: slot %d of %d", k, o.buf[b], n_slots);
        if (inf.taps) {
            const int nt = o.i[inf.ntaps_arg];
            if (nt < 1 || o.taps < 0 || (size_t)o.taps + 2 * (size_t)nt > tap_bytes)
                return fail(SDN_EINVAL, "sdn_program_create: record %d: tap pair outside the blob", k);
        }
        if (o.code == SDN_OP_CONV_GEMM_PHASES) {
            const int np = o.i[9];
            size_t total = 0;
            if (np < 1 || np > 4) return fail(SDN_EINVAL, "sdn_program_create: record %d: %d phases", k, np);
            for (int q = 0; q < np; q++) {
                const int nt = o.i[16 + 6 * q + 4];
                if (nt < 1) return fail(SDN_EINVAL, "sdn_program_create: record %d: phase %d without taps", k, q);
                total += 2 * (size_t)nt;
            }
            if (o.taps < 0 || (size_t)o.taps + total > tap_bytes)
                return fail(SDN_EINVAL, "sdn_program_create: record %d: tap lists outside the blob", k);
        }
    }
    sdn_program* p = new sdn_program;
    p->ops.assign(ops, ops + n_ops);
    if (tap_bytes) p->taps.assign(taps, taps + tap_bytes);
    p->n_slots = n_slots;
    *out = p;
    return SDN_OK;
}

SDN_API int sdn_program_destroy(sdn_program* prog)
{
    delete prog;
    return SDN_OK;
}

SDN_API int sdn_program_run(const sdn_program* prog, void* const* slots, int n_slots, sdnStream main_, sdnStream side_,
                            float* op_ms, int* failed_op)
{
    if (!prog || !slots) return fail(SDN_EINVAL, "sdn_program_run: null pointer");
    if (n_slots != prog->n_slots) return fail(SDN_EINVAL, "sdn_program_run: %d slots given, the program has %d", n_slots, prog->n_slots);
    hipStream_t streams[2] = {(hipStream_t)main_, (hipStream_t)side_};
    const bool two = streams[0] != streams[1];
    EventPair* ev = nullptr;
    std::vector<hipEvent_t> marks;  // measurement mode: 2 events per record
    if (op_ms) marks.assign(2 * prog->ops.size(), nullptr);
    auto P = [&](int slot) -> void* { return slot < 0 ? nullptr : slots[slot]; };
    const int8_t* blob = prog->taps.data();
    int rc = SDN_OK;
    size_t k = 0;
    for (; k < prog->ops.size() && rc == SDN_OK; k++) {
        const sdn_op& o = prog->ops[k];
        hipStream_t st = streams[o.stream];
        const int32_t* i = o.i;
        const int8_t* dy = o.taps >= 0 ? blob + o.taps : nullptr;
        if (op_ms) {
            if (hipEventCreate(&marks[2 * k]) != hipSuccess || hipEventCreate(&marks[2 * k + 1]) != hipSuccess) {
                rc = fail(SDN_ELAUNCH, "sdn_program_run: cannot create timing events");
                k++;   // (leave the loop as a failing record does: failed_op = k - 1 below)
                break;
            }
            (void)hipEventRecord(marks[2 * k], st);
        }
        if (!op_ms && multi_kind(o) && k + 1 < prog->ops.size() && multi_kind(prog->ops[k + 1]) &&
            prog->ops[k + 1].stream == o.stream) {
            // a run of pack / unpack / small-copy records on this stream: one launch (see k_weights_multi)
            WMulti M;
            M.n = 0;
            M.pad = 0;
            unsigned blocks = 0;
            const char* ext_src[MULTI_MAX];
            const char* ext_dst[MULTI_MAX];
            size_t ext_sn[MULTI_MAX], ext_dn[MULTI_MAX];
            size_t j = k;
            for (; j < prog->ops.size() && M.n < MULTI_MAX && multi_kind(prog->ops[j]) && prog->ops[j].stream == o.stream; j++) {
                const sdn_op& q = prog->ops[j];
                WDesc& D = M.d[M.n];
                std::memset(&D, 0, sizeof D);
                long elems = 0;
                if (q.code == SDN_OP_COPY) {
                    D.kind = 3; D.dst = P(q.buf[0]); D.src = P(q.buf[1]); D.sr = (long)q.l[0];
                    elems = (long)q.l[0] >> 2;
                    if (!D.src || !D.dst) { rc = fail(SDN_EINVAL, "sdn_program_run: COPY with a null pointer"); break; }
                } else {
                    D.src = P(q.buf[0]); D.tapidx = (const int*)P(q.buf[1]); D.dst = P(q.buf[2]);
                    D.R = q.i[0]; D.C = q.i[1]; D.ntaps = q.i[2]; D.Ccp = q.i[3]; D.sr = (long)q.l[0]; D.sc = (long)q.l[1];
                    if (!D.src || !D.dst || !D.tapidx || D.Ccp < D.C || D.ntaps < 1) {
                        rc = fail(SDN_EINVAL, "sdn_program_run: bad pack / unpack record %d", (int)j);
                        break;
                    }
                    if (q.code == SDN_OP_PACK_WEIGHTS) {
                        D.kind = 0; D.Kp = q.i[4]; D.rows = q.i[5];
                        if (D.Kp < D.ntaps * D.Ccp || (D.Kp & 31) || D.rows < D.R || (D.rows & 31) || (D.Ccp & 7)) {
                            rc = fail(SDN_EINVAL, "sdn_program_run: bad PACK_WEIGHTS record %d", (int)j);
                            break;
                        }
                        elems = (long)D.rows * D.Kp;
                    } else if (q.code == SDN_OP_PACK_WEIGHTS_KMAJOR) {
                        D.kind = 1; D.rows = q.i[4];
                        if ((D.Ccp & 31) || D.rows < D.R || (D.rows & 63)) {
                            rc = fail(SDN_EINVAL, "sdn_program_run: bad PACK_WEIGHTS_KMAJOR record %d", (int)j);
                            break;
                        }
                        elems = (long)D.rows * D.ntaps * D.Ccp;
                    } else {
                        D.kind = 2; D.accumulate = q.i[4];
                        if ((D.Ccp & 3) || ((uintptr_t)D.src & 15)) {   // 16-byte loads of four columns of one (row, tap)
                            rc = fail(SDN_EINVAL, "sdn_program_run: bad UNPACK_GRAD record %d", (int)j);
                            break;
                        }
                        elems = (long)D.R * D.ntaps * D.Ccp;
                        if (unpack_rows_ok(D.sc, D.ntaps) && D.sc < D.sr) D.kind = 4;
                    }
                }
                // The tensors of a run are processed concurrently by ONE launch: a record whose buffers OVERLAP (byte ranges,
                // not just equal base pointers: a COPY into a sub-range of an arena piece, two unpacks into offsets of one
                // gradient buffer) what an earlier record of the run writes, or that writes what an earlier one reads, ends
                // the run and starts the next one -- records keep their program order across the boundary.
                {
                    auto wbytes = [&](const WDesc& d) {   // a weight / weight-gradient tensor addressed through (sr, sc, tapidx)
                        const long asr = d.sr < 0 ? -d.sr : d.sr, asc = d.sc < 0 ? -d.sc : d.sc;
                        // (the taps of one (row, column) lie inside the innermost kh * kw block = the smaller of the two strides)
                        return (size_t)((long)(d.R - 1) * asr + (long)(d.C - 1) * asc + (asr < asc ? asr : asc)) * sizeof(float);
                    };
                    size_t sb, db;
                    if (D.kind == 3) sb = db = (size_t)D.sr;
                    else if (D.kind == 0) { sb = wbytes(D); db = (size_t)D.rows * D.Kp * 4; }
                    else if (D.kind == 1) { sb = wbytes(D); db = (size_t)D.rows * D.ntaps * D.Ccp * 4; }
                    else { sb = (size_t)D.R * D.ntaps * D.Ccp * sizeof(float); db = wbytes(D); }   // kinds 2 and 4
                    const char* s0 = (const char*)D.src;
                    const char* d0 = (const char*)D.dst;
                    auto meet = [](const char* a, size_t na, const char* b, size_t nb_) { return a < b + nb_ && b < a + na; };
                    bool clash = false;
                    for (int e = 0; e < M.n && !clash; e++)
                        clash = meet(ext_dst[e], ext_dn[e], s0, sb) || meet(ext_dst[e], ext_dn[e], d0, db) ||
                                meet(ext_src[e], ext_sn[e], d0, db);
                    if (clash) break;
                    ext_src[M.n] = s0; ext_sn[M.n] = sb; ext_dst[M.n] = d0; ext_dn[M.n] = db;
                }
                const long per_block = multi_block_elems(D.kind);
                const long nb = D.kind == 4 ? unpack_rows_blocks(D.R, D.C) : (elems + per_block - 1) / per_block;
                if (nb < 1 || (long)blocks + nb > 0x7fffffffL) { rc = fail(SDN_EINVAL, "sdn_program_run: pack run too large"); break; }
                D.first_block = blocks;
                blocks += (unsigned)nb;
                M.n++;
            }
            if (rc != SDN_OK) { k = j; k++; break; }    // failed_op = the record that was being gathered
            hipLaunchKernelGGL(k_weights_multi, dim3(blocks), dim3(256), 0, st, M);
            rc = check_launch("k_weights_multi");
            k = j - 1;     // the loop's k++ moves to the first record behind the run
            continue;
        }
        // conv records: the plan's true-channel flops, consumed by the record's first TimedLaunch; cleared behind the record so
        // that a record which fails validation or launches nothing timed cannot credit them to the next launch
        const bool declared = o.f[3] > 0.f;
        if (declared) timing_declare_work((double)o.f[3] * 1e9);
        switch (o.code) {
        case SDN_OP_CONV_GEMM:
            rc = sdn_conv_gemm((const float*)P(o.buf[0]), i[0], i[1], i[2], i[3], (float*)P(o.buf[1]), i[4], i[5], i[6], i[7],
                               i[8], i[9], i[10], i[11], i[12], i[13], dy, dy + i[13], i[14], i[15], P(o.buf[2]), i[16], i[17],
                               (const float*)P(o.buf[3]), i[18], (double*)P(o.buf[4]), i[19], i[20], P(o.buf[5]),
                               (size_t)o.l[0], st);
            break;
        case SDN_OP_CONV_NARROW_FWD:
            rc = sdn_conv_narrow_fwd((const float*)P(o.buf[0]), i[0], i[1], i[2], i[3], (float*)P(o.buf[1]), i[4], i[5], i[6],
                                     i[7], (const float*)P(o.buf[2]), i[8], i[9], i[10], i[11], i[12], i[13],
                                     (const float*)P(o.buf[3]), i[14], st);
            break;
        case SDN_OP_IN_APPLY:
            rc = sdn_in_apply((float*)P(o.buf[0]), (const double*)P(o.buf[1]), (float*)P(o.buf[2]), (const float*)P(o.buf[3]),
                              (float*)P(o.buf[4]), i[0], i[1], i[2], i[3], o.f[0], i[4], i[5], o.f[1], (float*)P(o.buf[5]),
                              (float*)P(o.buf[6]), P(o.buf[7]), (long)o.l[0], i[6], st);
            break;
        case SDN_OP_IN_BWD:
            rc = sdn_in_bwd((float*)P(o.buf[0]), (const float*)P(o.buf[1]), (const float*)P(o.buf[2]), (double*)P(o.buf[3]),
                            i[0], i[1], i[2], i[3], P(o.buf[4]), (long)o.l[0], st);
            break;
        case SDN_OP_ACT_BWD:
            rc = sdn_act_bwd((float*)P(o.buf[0]), (const float*)P(o.buf[1]), (float*)P(o.buf[2]), (long)o.l[0], i[0], i[1],
                             P(o.buf[3]), (long)o.l[1], st);
            break;
        case SDN_OP_REFLECT_FOLD:
            rc = sdn_reflect_fold((const float*)P(o.buf[0]), (float*)P(o.buf[1]), i[0], i[1], i[2], i[3], i[4], i[5], st);
            break;
        case SDN_OP_CONV_WGRAD:
            rc = sdn_conv_wgrad((const float*)P(o.buf[0]), (const float*)P(o.buf[1]), (float*)P(o.buf[2]), i[0], i[1], i[2],
                                i[3], i[4], i[5], i[6], i[7], i[8], dy, dy + i[8], i[9], i[10], i[11], i[12], i[13],
                                P(o.buf[3]), (size_t)o.l[0], st);
            break;
        case SDN_OP_CONV_WGRAD_NARROW:
            rc = sdn_conv_wgrad_narrow((const float*)P(o.buf[0]), (const float*)P(o.buf[1]), (float*)P(o.buf[2]), i[0], i[1],
                                       i[2], i[3], i[4], i[5], i[6], i[7], i[8], dy, dy + i[8], i[9], i[10], i[11], st);
            break;
        case SDN_OP_CONV_WGRAD_HEAD:
            rc = sdn_conv_wgrad_head_mfma((const float*)P(o.buf[0]), (const float*)P(o.buf[1]), (float*)P(o.buf[2]), i[0], i[1],
                                          i[2], i[3], i[4], i[5], i[6], i[7], i[8], dy, dy + i[8], i[9], i[10], i[11], st);
            break;
        case SDN_OP_PACK_WEIGHTS:
            rc = sdn_conv_pack_weights((const float*)P(o.buf[0]), i[0], i[1], (long)o.l[0], (long)o.l[1],
                                       (const int32_t*)P(o.buf[1]), i[2], i[3], i[4], i[5], P(o.buf[2]), st);
            break;
        case SDN_OP_UNPACK_GRAD:
            rc = sdn_conv_unpack_grad((const float*)P(o.buf[0]), i[0], i[1], (long)o.l[0], (long)o.l[1],
                                      (const int32_t*)P(o.buf[1]), i[2], i[3], (float*)P(o.buf[2]), i[4], st);
            break;
        case SDN_OP_MEMSET:
            if (o.l[0] > 0) rc = hip_ok(hipMemsetAsync(P(o.buf[0]), 0, (size_t)o.l[0], st), "memset");
            break;
        case SDN_OP_COPY:
            if (o.l[0] > 0)
                rc = hip_ok(hipMemcpyAsync(P(o.buf[0]), P(o.buf[1]), (size_t)o.l[0], hipMemcpyDeviceToDevice, st), "copy");
            break;
        case SDN_OP_ADD: {
            const long n4 = (long)(o.l[0] >> 2);
            if (o.l[0] & 3) {
                rc = fail(SDN_EINVAL, "sdn_program_run: ADD length %ld is not a multiple of 4", (long)o.l[0]);
                break;
            }
            if (n4 > 0) {
                const unsigned blocks = (unsigned)((n4 + 1023) / 1024 < 8192 ? (n4 + 1023) / 1024 : 8192);
                hipLaunchKernelGGL(k_add, dim3(blocks ? blocks : 1), dim3(256), 0, st, (float*)P(o.buf[0]),
                                   (const float*)P(o.buf[1]), (const float*)P(o.buf[2]), n4);
                rc = check_launch("k_add");
            }
            break;
        }
        case SDN_OP_COLSUM:
            if ((i[0] & 3) || i[1] < 1 || i[1] > i[0]) {
                rc = fail(SDN_EINVAL, "sdn_program_run: COLSUM pitch %d, %d columns", i[0], i[1]);
                break;
            }
            hipLaunchKernelGGL(k_colsum, dim3((unsigned)((i[1] + 3) / 4)), dim3(256), 0, st, (const float*)P(o.buf[0]),
                               (long)o.l[0], i[0], i[1], (float*)P(o.buf[1]));
            rc = check_launch("k_colsum");
            break;
        case SDN_OP_SPLIT_PLANES:
            rc = sdn_split_planes((const float*)P(o.buf[0]), (long)o.l[0], i[0], P(o.buf[1]), (long)o.l[1], st);
            break;
        case SDN_OP_PACK_WEIGHTS_KMAJOR:
            rc = sdn_conv_pack_weights_kmajor((const float*)P(o.buf[0]), i[0], i[1], (long)o.l[0], (long)o.l[1],
                                              (const int32_t*)P(o.buf[1]), i[2], i[3], i[4], P(o.buf[2]), st);
            break;
        case SDN_OP_CONV_TILE:
            rc = sdn_conv_tile(P(o.buf[0]), (long)o.l[0], i[0], i[1], i[2], i[3], (float*)P(o.buf[1]), P(o.buf[2]), (long)o.l[1],
                               i[4], i[5], i[6], i[7], i[8], i[9], i[10], i[11], i[12], i[13], i[14], dy, dy + i[14], i[15],
                               P(o.buf[3]), i[16], (const float*)P(o.buf[4]), i[17], (double*)P(o.buf[5]), i[18], i[19], st);
            break;
        case SDN_OP_CONV_HALO:
            rc = sdn_conv_halo(P(o.buf[0]), (long)o.l[0], i[0], i[1], i[2], i[3], (float*)P(o.buf[1]), i[4], i[5], i[6], i[7], dy,
                               dy + i[7], i[8], P(o.buf[2]), i[9], (const float*)P(o.buf[3]), i[10], (double*)P(o.buf[4]), i[11], st);
            break;
        case SDN_OP_CONV_WGRAD_TILE:
            rc = sdn_conv_wgrad_tile(P(o.buf[0]), (long)o.l[0], P(o.buf[1]), (long)o.l[1], (float*)P(o.buf[2]), i[0], i[1], i[2],
                                     i[3], i[4], i[5], i[6], i[7], i[8], dy, dy + i[8], i[9], st);
            break;
        case SDN_OP_CONV_HEAD_MFMA:
            rc = sdn_conv_head_mfma((const float*)P(o.buf[0]), i[0], i[1], i[2], i[3], (float*)P(o.buf[1]), i[4], i[5], i[6], i[7],
                                    P(o.buf[2]), i[8], i[9], i[10], i[11], i[12], i[13], (const float*)P(o.buf[3]), i[14],
                                    (double*)P(o.buf[4]), st);
            break;
        case SDN_OP_CONV_GEMM_PHASES: {
            const int np = i[9];
            int32_t qh[4], qw[4], py[4], px[4], nt[4], kp[4];
            const void* wp[4];
            for (int q = 0; q < np; q++) {
                const int32_t* r = i + 16 + 6 * q;
                qh[q] = r[0]; qw[q] = r[1]; py[q] = r[2]; px[q] = r[3]; nt[q] = r[4]; kp[q] = r[5];
                wp[q] = P(o.buf[2 + q]);
            }
            rc = sdn_conv_gemm_phases((const float*)P(o.buf[0]), i[0], i[1], i[2], i[3], (float*)P(o.buf[1]), i[4], i[5], i[6],
                                      i[7], i[8], np, qh, qw, py, px, nt, dy, i[10], i[11], wp, kp, i[12],
                                      (const float*)P(o.buf[6]), i[13], (double*)P(o.buf[7]), i[14], i[15], st);
            break;
        }
        case SDN_OP_FORK:
        case SDN_OP_JOIN:
            if (two) {
                if (!ev) rc = events_for_current_device(&ev);
                if (rc != SDN_OK) break;
                const bool fork = o.code == SDN_OP_FORK;
                static const int dbg = [] { const char* e = getenv("SDN_DEBUG_FORK"); return e ? atoi(e) : 0; }();
                if (dbg == 1) {          // lab: the host waits for the recording stream -- no event semantics involved
                    rc = hip_ok(hipStreamSynchronize(streams[fork ? 0 : 1]), "debug synchronize");
                    break;
                }
                if (dbg == 2) {          // lab: a fresh event per record
                    hipEvent_t fe;
                    rc = hip_ok(hipEventCreateWithFlags(&fe, hipEventDisableTiming), "event create");
                    if (rc == SDN_OK) rc = hip_ok(hipEventRecord(fe, streams[fork ? 0 : 1]), "event record");
                    if (rc == SDN_OK) rc = hip_ok(hipStreamWaitEvent(streams[fork ? 1 : 0], fe, 0), "stream wait");
                    (void)hipEventDestroy(fe);
                    break;
                }
                hipEvent_t e = fork ? ev->fork : ev->join;
                rc = hip_ok(hipEventRecord(e, streams[fork ? 0 : 1]), "event record");
                if (rc == SDN_OK) rc = hip_ok(hipStreamWaitEvent(streams[fork ? 1 : 0], e, 0), "stream wait");
            }
            break;
        default:
            rc = fail(SDN_EINVAL, "sdn_program_run: record %zu: code %d", k, o.code);
        }
        if (declared) timing_declare_work(0.0);
        {   // lab: SDN_DEBUG_SYNC_CODES="8,23": the host waits for the device behind every record of these codes
            static const unsigned long long sync_mask = [] {
                unsigned long long m = 0;
                if (const char* e = getenv("SDN_DEBUG_SYNC_CODES"))
                    for (const char* p = e; *p;) {
                        char* end = nullptr;
                        const long v = strtol(p, &end, 10);
                        if (end == p) break;
                        if (v > 0 && v < 64) m |= 1ull << v;
                        p = *end ? end + 1 : end;
                    }
                return m;
            }();
            if (sync_mask && ((sync_mask >> o.code) & 1ull)) (void)hipDeviceSynchronize();
        }
        if (op_ms && marks[2 * k + 1]) (void)hipEventRecord(marks[2 * k + 1], st);
    }
    if (rc != SDN_OK && failed_op) *failed_op = (int)(k ? k - 1 : 0);
    if (op_ms) {
        for (int s = 0; s < 2; s++) (void)hipStreamSynchronize(streams[s]);
        for (size_t j = 0; j < prog->ops.size(); j++) {
            float ms = 0.f;
            if (marks[2 * j] && marks[2 * j + 1]) (void)hipEventElapsedTime(&ms, marks[2 * j], marks[2 * j + 1]);
            op_ms[j] = ms;
            if (marks[2 * j]) (void)hipEventDestroy(marks[2 * j]);
            if (marks[2 * j + 1]) (void)hipEventDestroy(marks[2 * j + 1]);
        }
    }
    return rc;
}
