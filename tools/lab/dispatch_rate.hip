// Lab (GPU only): how fast does gfx950 retire workgroups that have nothing to do?  Decides whether launching four workgroups
// per raster tile (three of which usually return at once) is affordable.   hipcc --offload-arch=gfx950 -O2 -o dispatch_rate dispatch_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int LDS, int VG>
__global__ __launch_bounds__(256) void k_empty(const unsigned* __restrict__ order, unsigned* out, int work_mod)
{
    __shared__ unsigned s[LDS / 4 > 0 ? LDS / 4 : 1];
    const unsigned r = order[blockIdx.x >> 2];
    if ((blockIdx.x & 3) != 0 && work_mod) return;
    float acc[VG];
#pragma unroll
    for (int i = 0; i < VG; i++) acc[i] = (float)(r + i);
    s[threadIdx.x % (LDS / 4 > 0 ? LDS / 4 : 1)] = r;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < VG; i++) t += acc[i] * (float)s[(threadIdx.x + i) % (LDS / 4 > 0 ? LDS / 4 : 1)];
    if (t == 12345.678f) out[0] = 1;
}

template <int LDS, int VG>
static void run(const char* name, unsigned* order, unsigned* out, int blocks, int work_mod)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k_empty<LDS, VG>), dim3(blocks), dim3(256), 0, 0, order, out, work_mod);
    hipEventRecord(a, 0);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL((k_empty<LDS, VG>), dim3(blocks), dim3(256), 0, 0, order, out, work_mod);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    printf("%-28s blocks %6d  early-exit 3/4 %d : %8.1f us per launch, %6.1f ns per workgroup\n", name, blocks, work_mod, ms * 1e3 / 20,
           ms * 1e6 / 20 / blocks);
}

int main()
{
    unsigned *order, *out;
    hipMalloc(&order, 4 * 65536);
    hipMemset(order, 0, 4 * 65536);
    hipMalloc(&out, 64);
    for (int blocks : {9216, 36864}) {
        for (int wm : {0, 1}) {
            run<0, 1>("no LDS, few VGPRs", order, out, blocks, wm);
            run<25600, 1>("25.6 KB LDS, few VGPRs", order, out, blocks, wm);
            run<25600, 96>("25.6 KB LDS, ~100 VGPRs", order, out, blocks, wm);
        }
    }
    return 0;
}
