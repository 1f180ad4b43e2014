// bf16 operand planes for the tiled MFMA kernels (conv_tile.hip, conv_wtile.hip).
//
// The textural convolutions (/root/reference/textural/models/networks.py:211-283, 412-461; cuDNN fp32 in the reference) run as
// three bf16 MFMA products of split operands x = hi + lo (conv_common.h).  k_conv_gemm / k_conv_wgrad re-split the fp32
// activations in EVERY workgroup that gathered them (x8 channel tiles, x9 taps).  A plane pair [2][elements] bf16 is the
// split done once: written by the tensor's producer (the plane outputs of sdn_in_apply / sdn_in_bwd / sdn_act_bwd and of
// sdn_conv_tile's epilogue) or, for tensors that enter a chain from outside, by sdn_split_planes.  A tensor whose consumers
// apply ReLU on load (the deferred ReLU of conv.py) is split AFTER the ReLU: the planes hold what the consumers multiply.
#include "conv_common.h"
#include "sdn_common.h"

namespace sdn {

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

__global__ __launch_bounds__(256) void k_split_planes(const float* __restrict__ x, long n4, int relu, __bf16* __restrict__ hi,
                                                      __bf16* __restrict__ lo)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    if (relu) {
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], 0.f);
    }
    const SplitBf16 a = split2(v[0], v[1]), b = split2(v[2], v[3]);
    reinterpret_cast<bf16x4*>(hi)[i] = bf16x4{a.hi[0], a.hi[1], b.hi[0], b.hi[1]};
    reinterpret_cast<bf16x4*>(lo)[i] = bf16x4{a.lo[0], a.lo[1], b.lo[0], b.lo[1]};
}

// ---- chain inputs: NCHW parts side by side -> one channels-last buffer [N, H, W, Cp], pad channels zero.
// What the caller's torch.cat + the executor's permute + pad did as one strided torch copy per part plus a zero fill of the
// whole buffer (pix2pixHD_model.py:155-166 concatenates label one-hots, pose bins, edge map, features: 48 channels for G, 18
// for D; 1.6 ms of torch launches per GAN step).  A workgroup transposes ASM_PX pixels of one image row through LDS: reads
// are coalesced along W (the parts' fast axis), writes along C (the buffer's).
constexpr int ASM_PX = 64, ASM_MAX_PARTS = 8, ASM_MAX_CP = 128;
struct AssembleParams {
    const float* part[ASM_MAX_PARTS];
    int first[ASM_MAX_PARTS + 1];   // channel offset of every part; first[nparts] = C
    int nparts, N, H, W, Cp;
    float* out;
};

__global__ __launch_bounds__(256) void k_assemble_nhwc(const AssembleParams P)
{
    __shared__ float tile[ASM_MAX_CP][ASM_PX + 1];
    const int C = P.first[P.nparts];
    // one grid axis: N * H rows x ceil(W / ASM_PX) column blocks (grid.y stops at 65535 rows: 64 images of 1024 rows)
    const unsigned nbx = (unsigned)((P.W + ASM_PX - 1) / ASM_PX);
    const long row = blockIdx.x / nbx;              // n * H + h
    const int n = (int)(row / P.H), h = (int)(row % P.H);
    const int w0 = (int)(blockIdx.x % nbx) * ASM_PX;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t hw = (size_t)P.H * P.W;
    for (int c = wave; c < C; c += 4) {
        int k = 0;
        while (c >= P.first[k + 1]) k++;            // wave-uniform
        const int cl = c - P.first[k], ck = P.first[k + 1] - P.first[k];
        const int w = w0 + lane;
        tile[c][lane] = w < P.W ? P.part[k][((size_t)n * ck + cl) * hw + (size_t)h * P.W + w] : 0.f;
    }
    __syncthreads();
    // Cp floats per pixel, consecutive threads consecutive channels
    const int total = ASM_PX * P.Cp;
    for (int i = threadIdx.x; i < total; i += 256) {
        const int px = i / P.Cp, c = i - px * P.Cp;
        const int w = w0 + px;
        if (w < P.W) P.out[((size_t)row * P.W + w) * P.Cp + c] = c < C ? tile[c][px] : 0.f;
    }
}

}  // namespace sdn

using namespace sdn;

SDN_API int sdn_assemble_nhwc(const float* const* parts, const int32_t* channels, int nparts, int N, int H, int W, int Cp,
                              float* out, sdnStream stream)
{
    if (!parts || !channels || !out || nparts < 1 || nparts > ASM_MAX_PARTS || N < 1 || H < 1 || W < 1)
        return fail(SDN_EINVAL, "sdn_assemble_nhwc: bad argument (1..%d parts)", ASM_MAX_PARTS);
    AssembleParams P;
    int c = 0;
    for (int k = 0; k < nparts; k++) {
        if (!parts[k] || channels[k] < 1) return fail(SDN_EINVAL, "sdn_assemble_nhwc: part %d is empty", k);
        P.part[k] = parts[k];
        P.first[k] = c;
        c += channels[k];
    }
    for (int k = nparts; k < ASM_MAX_PARTS; k++) {
        P.part[k] = nullptr;
        P.first[k + 1] = c;
    }
    P.first[nparts] = c;
    if (Cp < c || Cp > ASM_MAX_CP) return fail(SDN_EINVAL, "sdn_assemble_nhwc: %d channels into a buffer of %d (max %d)", c, Cp, ASM_MAX_CP);
    const long blocks = (long)N * H * cdiv(W, ASM_PX);
    if (blocks > 0x7fffffffL) return fail(SDN_EINVAL, "sdn_assemble_nhwc: too many image rows");
    P.nparts = nparts; P.N = N; P.H = H; P.W = W; P.Cp = Cp; P.out = out;
    hipLaunchKernelGGL(k_assemble_nhwc, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, P);
    return check_launch("k_assemble_nhwc");
}

SDN_API int sdn_split_planes(const float* x, long n, int relu, void* planes, long plane_stride, sdnStream stream)
{
    if (!x || !planes || n < 0 || (n & 3) || plane_stride < n || (plane_stride & 7))
        return fail(SDN_EINVAL, "sdn_split_planes: bad argument (n %ld %% 4, plane stride %ld %% 8)", n, plane_stride);
    if (n == 0) return SDN_OK;
    hipLaunchKernelGGL(k_split_planes, dim3(cdiv(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, x, n / 4, relu,
                       (__bf16*)planes, (__bf16*)planes + plane_stride);
    return check_launch("k_split_planes");
}
