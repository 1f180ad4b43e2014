// Mask R-CNN's two custom ops for gfx950: greedy non-maximum suppression and crop_and_resize (RoIAlign's core).
//
// Reference: geometric/maskrcnn/nms/src/nms.c:4-69 (cpu_nms) with its CUDA twin nms/src/cuda/nms_kernel.cu:26-82 +
// nms_cuda.c:17-67, and geometric/maskrcnn/roialign/roi_align/src/crop_and_resize.c:7-251 with
// src/cuda/crop_and_resize_kernel.cu:10-185.  Only --source maskrcnn of geometric/scripts/main.py needs them (SURVEY.md 8f
// n4).  Results follow the C paths operation by operation (this file is built without FMA contraction), so kept indices
// and crops are bit-identical to them; the crop gradient is a scatter-add with float atomics (as the reference's CUDA
// path) and matches the serial C sum up to re-association.
//
//   k_nms_mask  one wave per 64 x 64 block of the (row box i, column box j) pairs of the score-sorted boxes: lane = row
//               box, 64 column boxes staged in LDS, bit j of mask[i][col block] = (j after i) && IoU >= thresh.  The
//               wave IS the 64-bit word: no masking of partial warps as in the 64-thread CUDA blocks.
//   k_nms_scan  ONE wave walks the boxes in score order with the running "removed" bitmap in LDS and ORs in the mask row
//               of every kept box (lanes = words).  The reference copies the mask to the host for this loop
//               (nms_cuda.c:33-58); here it stays on the device, so nms() has no host round trip.
//   k_crop_fwd / k_crop_bwd  one thread per crop element; bilinear taps exactly as CropAndResizePerBox.
#include "sdn_common.h"

namespace sdn {

// nms.c:51-58 with box i's corners first.  IoU is symmetric in (x, y), so the caller's column order does not matter.
__device__ __forceinline__ bool suppresses(const float4 bi, const float ai, const float4 bj, const float aj, const float thresh,
                                           const int strict)
{
    const float xx1 = fmaxf(bi.x, bj.x), yy1 = fmaxf(bi.y, bj.y);
    const float xx2 = fminf(bi.z, bj.z), yy2 = fminf(bi.w, bj.w);
    const float w = fmaxf(0.0f, xx2 - xx1 + 1.0f), h = fmaxf(0.0f, yy2 - yy1 + 1.0f);
    const float inter = w * h;
    const float ovr = inter / (ai + aj - inter);
    return strict ? ovr > thresh : ovr >= thresh;  // nms_kernel.cu:66 compares with `>`, cpu_nms (nms.c:59) with `>=`
}

// boxes [n,4], areas [n]: already in descending score order.  mask [n, nb] u64, nb = ceil(n / 64).
__global__ __launch_bounds__(64) void k_nms_mask(const float4* __restrict__ boxes, const float* __restrict__ areas, int n,
                                                 float thresh, unsigned long long* __restrict__ mask, int nb, int strict)
{
    __shared__ float4 cb[64];
    __shared__ float ca[64];
    const int row0 = blockIdx.y * 64, col0 = blockIdx.x * 64, lane = threadIdx.x;
    if (col0 + 63 <= row0 && blockIdx.x != blockIdx.y) {  // every column precedes every row: nothing to suppress
        if (row0 + lane < n) mask[(size_t)(row0 + lane) * nb + blockIdx.x] = 0ull;
        return;
    }
    if (col0 + lane < n) {
        cb[lane] = boxes[col0 + lane];
        ca[lane] = areas[col0 + lane];
    }
    __syncthreads();
    const int i = row0 + lane;
    if (i >= n) return;
    const float4 bi = boxes[i];
    const float ai = areas[i];
    const int cols = min(64, n - col0);
    unsigned long long t = 0ull;
    for (int c = 0; c < cols; c++) {
        const int j = col0 + c;
        if (j <= i) continue;
        const bool s = suppresses(bi, ai, cb[c], ca[c], thresh, strict);
        if (s) t |= 1ull << c;
    }
    mask[(size_t)i * nb + blockIdx.x] = t;
}

constexpr int NMS_MAX_WORDS = 2048;  // boxes <= 131072

__global__ __launch_bounds__(64) void k_nms_scan(const unsigned long long* __restrict__ mask, int n, int nb,
                                                 long long* __restrict__ keep, long long* __restrict__ count)
{
    __shared__ unsigned long long remv[NMS_MAX_WORDS];
    const int lane = threadIdx.x;
    for (int w = lane; w < nb; w += 64) remv[w] = 0ull;
    __syncthreads();
    long long kept = 0;
    for (int blk = 0; blk < nb; blk++) {
        // boxes of one word: decisions inside the word depend on each other through remv[blk] only
        const int base = blk * 64, cnt = min(64, n - base);
        for (int b = 0; b < cnt; b++) {
            const unsigned long long w = remv[blk];  // uniform
            if ((w >> b) & 1ull) continue;
            const int i = base + b;
            if (lane == 0) keep[kept] = i;
            kept++;
            const unsigned long long* row = mask + (size_t)i * nb;
            for (int j = blk + lane; j < nb; j += 64) remv[j] |= row[j];
            __syncthreads();
        }
    }
    if (lane == 0) *count = kept;
}

struct CropParams {
    const float* image;   // [B, C, H, W]
    const float* boxes;   // [n, 4] (y1, x1, y2, x2), normalised
    const int* box_index; // [n]
    float* crops;         // [n, C, ch, cw]
    const float* grads;   // backward: [n, C, ch, cw]
    float* grads_image;   // backward: [B, C, H, W]
    int B, C, H, W, n, ch, cw;
    float extrapolation;
};

// crop_and_resize.c:43-58: source coordinate of output index k along one axis (float / double mixing as written there)
__device__ __forceinline__ float axis_in(float a1, float a2, int extent, int crop, int k)
{
    if (crop > 1) {
        const float scale = (a2 - a1) * (float)(extent - 1) / (float)(crop - 1);
        return a1 * (float)(extent - 1) + (float)k * scale;
    }
    return (float)(0.5 * (double)(a1 + a2) * (double)(extent - 1));
}

__global__ __launch_bounds__(256) void k_crop_fwd(const CropParams P)
{
    const long total = (long)P.n * P.C * P.ch * P.cw;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % P.cw);
    long q = i / P.cw;
    const int y = (int)(q % P.ch);
    q /= P.ch;
    const int d = (int)(q % P.C), b = (int)(q / P.C);
    const float* box = P.boxes + 4 * (long)b;
    const int b_in = P.box_index[b];
    float out = P.extrapolation;
    const float in_y = axis_in(box[0], box[2], P.H, P.ch, y);
    const float in_x = axis_in(box[1], box[3], P.W, P.cw, x);
    const bool inside = !(in_y < 0 || in_y > (float)(P.H - 1)) && !(in_x < 0 || in_x > (float)(P.W - 1));
    if (inside && b_in >= 0 && b_in < P.B) {
        const int top = (int)floorf(in_y), bottom = (int)ceilf(in_y);
        const int left = (int)floorf(in_x), right = (int)ceilf(in_x);
        const float y_lerp = in_y - (float)top, x_lerp = in_x - (float)left;
        const float* p = P.image + ((long)b_in * P.C + d) * P.H * P.W;
        const float tl = p[(long)top * P.W + left], tr = p[(long)top * P.W + right];
        const float bl = p[(long)bottom * P.W + left], br = p[(long)bottom * P.W + right];
        const float t = tl + (tr - tl) * x_lerp;
        const float bt = bl + (br - bl) * x_lerp;
        out = t + (bt - t) * y_lerp;
    }
    P.crops[i] = out;
}

__global__ __launch_bounds__(256) void k_crop_bwd(const CropParams P)
{
    const long total = (long)P.n * P.C * P.ch * P.cw;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % P.cw);
    long q = i / P.cw;
    const int y = (int)(q % P.ch);
    q /= P.ch;
    const int d = (int)(q % P.C), b = (int)(q / P.C);
    const float* box = P.boxes + 4 * (long)b;
    const int b_in = P.box_index[b];
    if (b_in < 0 || b_in >= P.B) return;
    const float in_y = axis_in(box[0], box[2], P.H, P.ch, y);
    if (in_y < 0 || in_y > (float)(P.H - 1)) return;
    const float in_x = axis_in(box[1], box[3], P.W, P.cw, x);
    if (in_x < 0 || in_x > (float)(P.W - 1)) return;
    const int top = (int)floorf(in_y), bottom = (int)ceilf(in_y);
    const int left = (int)floorf(in_x), right = (int)ceilf(in_x);
    const float y_lerp = in_y - (float)top, x_lerp = in_x - (float)left;
    float* p = P.grads_image + ((long)b_in * P.C + d) * P.H * P.W;
    const float g = P.grads[i];
    const float dtop = (1.0f - y_lerp) * g;
    unsafeAtomicAdd(p + (long)top * P.W + left, (1.0f - x_lerp) * dtop);
    unsafeAtomicAdd(p + (long)top * P.W + right, x_lerp * dtop);
    const float dbottom = y_lerp * g;
    unsafeAtomicAdd(p + (long)bottom * P.W + left, (1.0f - x_lerp) * dbottom);
    unsafeAtomicAdd(p + (long)bottom * P.W + right, x_lerp * dbottom);
}

}  // namespace sdn

using namespace sdn;

SDN_API int sdn_nms_workspace_bytes(int n, size_t* out)
{
    if (n < 0 || !out) return fail(SDN_EINVAL, "sdn_nms_workspace_bytes: bad arguments");
    const size_t nb = ((size_t)n + 63) / 64;
    *out = (size_t)(n > 0 ? n : 1) * (nb > 0 ? nb : 1) * sizeof(unsigned long long);
    return SDN_OK;
}

SDN_API int sdn_nms(const float* boxes_sorted, const float* areas_sorted, int n, float thresh, int strict, long long* keep,
                    long long* count, void* workspace, size_t workspace_bytes, sdnStream stream)
{
    if (n < 0 || !keep || !count || (n > 0 && (!boxes_sorted || !areas_sorted || !workspace)))
        return fail(SDN_EINVAL, "sdn_nms: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        if (hipMemsetAsync(count, 0, sizeof(long long), st) != hipSuccess) return fail(SDN_ELAUNCH, "sdn_nms: memset");
        return SDN_OK;
    }
    const int nb = (n + 63) / 64;
    if (nb > NMS_MAX_WORDS) return fail(SDN_EINVAL, "sdn_nms: %d boxes > %d", n, NMS_MAX_WORDS * 64);
    if (workspace_bytes < (size_t)n * nb * sizeof(unsigned long long))
        return fail(SDN_ENOMEM, "sdn_nms: workspace %zu < %zu bytes", workspace_bytes, (size_t)n * nb * sizeof(unsigned long long));
    unsigned long long* mask = (unsigned long long*)workspace;
    hipLaunchKernelGGL(k_nms_mask, dim3(nb, nb), dim3(64), 0, st, (const float4*)boxes_sorted, areas_sorted, n, thresh, mask,
                       nb, strict);
    hipLaunchKernelGGL(k_nms_scan, dim3(1), dim3(64), 0, st, mask, n, nb, keep, count);
    return check_launch("sdn_nms");
}

SDN_API int sdn_crop_and_resize_fwd(const float* image, int B, int C, int H, int W, const float* boxes, const int32_t* box_index,
                                    int n, int crop_h, int crop_w, float extrapolation, float* crops, sdnStream stream)
{
    if (!image || !crops || B <= 0 || C <= 0 || H <= 0 || W <= 0 || n < 0 || crop_h <= 0 || crop_w <= 0 || (n > 0 && (!boxes || !box_index)))
        return fail(SDN_EINVAL, "sdn_crop_and_resize_fwd: bad arguments");
    if (n == 0) return SDN_OK;
    CropParams P{image, boxes, box_index, crops, nullptr, nullptr, B, C, H, W, n, crop_h, crop_w, extrapolation};
    const long total = (long)n * C * crop_h * crop_w;
    hipLaunchKernelGGL(k_crop_fwd, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, P);
    return check_launch("k_crop_fwd");
}

SDN_API int sdn_crop_and_resize_bwd(const float* grads, const float* boxes, const int32_t* box_index, int n, int crop_h, int crop_w,
                                    float* grads_image, int B, int C, int H, int W, sdnStream stream)
{
    if (!grads_image || B <= 0 || C <= 0 || H <= 0 || W <= 0 || n < 0 || crop_h <= 0 || crop_w <= 0 ||
        (n > 0 && (!grads || !boxes || !box_index)))
        return fail(SDN_EINVAL, "sdn_crop_and_resize_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(grads_image, 0, sizeof(float) * (size_t)B * C * H * W, st) != hipSuccess)
        return fail(SDN_ELAUNCH, "sdn_crop_and_resize_bwd: memset");
    if (n == 0) return SDN_OK;
    CropParams P{nullptr, boxes, box_index, nullptr, grads, grads_image, B, C, H, W, n, crop_h, crop_w, 0.f};
    const long total = (long)n * C * crop_h * crop_w;
    hipLaunchKernelGGL(k_crop_bwd, dim3(cdiv(total, 256)), dim3(256), 0, st, P);
    return check_launch("k_crop_bwd");
}
