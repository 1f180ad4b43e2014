// nn.AvgPool2d(3, stride=2, padding=1, count_include_pad=False): the input pyramid of the multi-scale discriminator and of
// the LocalEnhancer (/root/reference/textural/models/networks.py:190, 392, 406).  Own kernels, forward and backward, because
// torch 2.10 / ROCm 7's avg_pool2d BACKWARD returns wrong gradients for channels-last-strided inputs with
// count_include_pad=False (found by tests/test_gpu_trainstep.py against the reference's train loop: the generator's output --
// a channels-last view -- goes straight into the discriminator's pyramid; forward values were right, d loss / d image was off
// by 70 %; tests/test_gpu_textural.py::test_pyramid_pooling_gradient_for_strided_views pins it).
//   out[n,c,oh,ow] = mean of in[n,c,2oh-1..2oh+1, 2ow-1..2ow+1] over the taps inside the image.
// Tensors are addressed through element strides (n, c, h, w), so NCHW tensors and channels-last views both work without a
// copy; threads are laid out over the OUTPUT (forward) / INPUT-gradient (backward) in the order the caller says is the
// fastest (`inner_c`: channel fastest, i.e. channels-last storage).  HBM-bound: one read of the input, a quarter written.
#include <hip/hip_runtime.h>

#include "sdn_common.h"

namespace sdn {

struct PoolParams {
    int N, C, H, W, OH, OW;
    long is[4], os[4];  // element strides (n, c, h, w) of the tensor read and of the tensor written
    int inner_c;
};

__device__ __forceinline__ void pool_index(const PoolParams& P, long idx, int d2, int d3, int& n, int& c, int& y, int& x)
{
    // idx enumerates (n, y, x, c) when inner_c else (n, c, y, x); (d2, d3) are the spatial extents of the tensor enumerated
    if (P.inner_c) {
        c = (int)(idx % P.C);
        idx /= P.C;
        x = (int)(idx % d3);
        idx /= d3;
        y = (int)(idx % d2);
        n = (int)(idx / d2);
    } else {
        x = (int)(idx % d3);
        idx /= d3;
        y = (int)(idx % d2);
        idx /= d2;
        c = (int)(idx % P.C);
        n = (int)(idx / P.C);
    }
}

__global__ __launch_bounds__(256) void k_avgpool3s2_fwd(const float* __restrict__ in, float* __restrict__ out, const PoolParams P)
{
    const long total = (long)P.N * P.C * P.OH * P.OW;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    int n, c, oh, ow;
    pool_index(P, idx, P.OH, P.OW, n, c, oh, ow);
    const int y0 = max(2 * oh - 1, 0), y1 = min(2 * oh + 1, P.H - 1);
    const int x0 = max(2 * ow - 1, 0), x1 = min(2 * ow + 1, P.W - 1);
    const float* src = in + n * P.is[0] + c * P.is[1];
    float s = 0.f;
    for (int y = y0; y <= y1; y++)
        for (int x = x0; x <= x1; x++) s += src[y * P.is[2] + x * P.is[3]];
    out[n * P.os[0] + c * P.os[1] + oh * P.os[2] + ow * P.os[3]] = s / (float)((y1 - y0 + 1) * (x1 - x0 + 1));
}

// gather form of the adjoint: input pixel (y, x) lies in the windows oh in [ceil((y-1)/2), floor((y+1)/2)] (one or two)
__global__ __launch_bounds__(256) void k_avgpool3s2_bwd(const float* __restrict__ g, float* __restrict__ gin, const PoolParams P)
{
    const long total = (long)P.N * P.C * P.H * P.W;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    int n, c, y, x;
    pool_index(P, idx, P.H, P.W, n, c, y, x);
    const int oh0 = y >> 1, oh1 = min((y + 1) >> 1, P.OH - 1);
    const int ow0 = x >> 1, ow1 = min((x + 1) >> 1, P.OW - 1);
    const float* src = g + n * P.is[0] + c * P.is[1];
    float s = 0.f;
    for (int oh = oh0; oh <= oh1; oh++) {
        const int ny = min(2 * oh + 1, P.H - 1) - max(2 * oh - 1, 0) + 1;
        for (int ow = ow0; ow <= ow1; ow++) {
            const int nx = min(2 * ow + 1, P.W - 1) - max(2 * ow - 1, 0) + 1;
            s += src[oh * P.is[2] + ow * P.is[3]] / (float)(ny * nx);
        }
    }
    gin[n * P.os[0] + c * P.os[1] + y * P.os[2] + x * P.os[3]] = s;
}

}  // namespace sdn

using namespace sdn;

static int pool_params(PoolParams& P, int N, int C, int H, int W, const long* in_strides, const long* out_strides, int inner_c,
                       const char* who)
{
    if (N < 1 || C < 1 || H < 1 || W < 1 || !in_strides || !out_strides) return fail(SDN_EINVAL, "%s: bad arguments", who);
    P.N = N; P.C = C; P.H = H; P.W = W;
    P.OH = (H - 1) / 2 + 1;   // floor((H + 2 - 3) / 2) + 1
    P.OW = (W - 1) / 2 + 1;
    for (int k = 0; k < 4; k++) {
        P.is[k] = in_strides[k];
        P.os[k] = out_strides[k];
    }
    P.inner_c = inner_c != 0;
    return SDN_OK;
}

SDN_API int sdn_avgpool3x3s2_fwd(const float* in, int N, int C, int H, int W, const long* in_strides, float* out,
                                 const long* out_strides, int inner_c, sdnStream stream)
{
    if (!in || !out) return fail(SDN_EINVAL, "sdn_avgpool3x3s2_fwd: null pointer");
    PoolParams P;
    if (int rc = pool_params(P, N, C, H, W, in_strides, out_strides, inner_c, "sdn_avgpool3x3s2_fwd")) return rc;
    const long total = (long)N * C * P.OH * P.OW;
    hipLaunchKernelGGL(k_avgpool3s2_fwd, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, in, out, P);
    return check_launch("k_avgpool3s2_fwd");
}

SDN_API int sdn_avgpool3x3s2_bwd(const float* g, int N, int C, int H, int W, const long* g_strides, float* gin,
                                 const long* gin_strides, int inner_c, sdnStream stream)
{
    if (!g || !gin) return fail(SDN_EINVAL, "sdn_avgpool3x3s2_bwd: null pointer");
    PoolParams P;
    if (int rc = pool_params(P, N, C, H, W, g_strides, gin_strides, inner_c, "sdn_avgpool3x3s2_bwd")) return rc;
    const long total = (long)N * C * H * W;
    hipLaunchKernelGGL(k_avgpool3s2_bwd, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, g, gin, P);
    return check_launch("k_avgpool3s2_bwd");
}
