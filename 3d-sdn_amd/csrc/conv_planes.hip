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

}  // namespace sdn

using namespace sdn;

SDN_API int sdn_split_planes(const float* x, long n, int relu, void* planes, long plane_stride, sdnStream stream)
{
    if (!x || !planes || n < 0 || (n & 3) || plane_stride < n || (plane_stride & 7))
        return fail(SDN_EINVAL, "sdn_split_planes: bad argument (n %ld %% 4, plane stride %ld %% 8)", n, plane_stride);
    if (n == 0) return SDN_OK;
    hipLaunchKernelGGL(k_split_planes, dim3(cdiv(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, x, n / 4, relu,
                       (__bf16*)planes, (__bf16*)planes + plane_stride);
    return check_launch("k_split_planes");
}
