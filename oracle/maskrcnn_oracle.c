/* CPU restatement of the reference's Mask R-CNN custom ops (TEST INFRASTRUCTURE ONLY: the checker of
 * 3d-sdn_amd/csrc/raster_boxes.hip; never linked into the product).
 *
 *   mrcnn_nms            geometric/maskrcnn/nms/src/nms.c:4-69 (cpu_nms): greedy suppression in score order,
 *                        IoU with the +1 pixel convention, suppress when ovr >= thresh
 *   mrcnn_crop_forward   geometric/maskrcnn/roialign/roi_align/src/crop_and_resize.c:7-114 (CropAndResizePerBox)
 *   mrcnn_crop_backward  crop_and_resize.c:160-251
 * Float / double mixing follows the C source: `0.5 * (y1 + y2) * (H - 1)` is evaluated in double and rounded once,
 * `fmaxf(0.0, x)` takes the float of the double literal; built with -ffp-contract=off.  tests/test_maskrcnn_oracle.py
 * demands bit equality with oracle/_ref/libmaskrcnn_ref.so (the reference's files compiled unmodified). */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define API __attribute__((visibility("default")))

/* boxes [n,4] in the order the CALLER of cpu_nms passes them (nms.c reads columns 0..3 as x1 y1 x2 y2 of `dets`: with the
 * reference's pth_nms that is (y1, x1, y2, x2) of its detections -- IoU is symmetric in the two axes), order [n] = indices
 * by descending score, areas [n].  keep [n] receives the kept indices; returns their number. */
API long mrcnn_nms(const float* boxes, long n, long dim, const long* order, const float* areas, float thresh, long* keep)
{
    unsigned char* suppressed = (unsigned char*)calloc(n > 0 ? n : 1, 1);
    long num = 0;
    for (long _i = 0; _i < n; ++_i) {
        const long i = order[_i];
        if (suppressed[i] == 1) continue;
        keep[num++] = i;
        const float ix1 = boxes[i * dim], iy1 = boxes[i * dim + 1], ix2 = boxes[i * dim + 2], iy2 = boxes[i * dim + 3];
        const float iarea = areas[i];
        for (long _j = _i + 1; _j < n; ++_j) {
            const long j = order[_j];
            if (suppressed[j] == 1) continue;
            const float xx1 = fmaxf(ix1, boxes[j * dim]);
            const float yy1 = fmaxf(iy1, boxes[j * dim + 1]);
            const float xx2 = fminf(ix2, boxes[j * dim + 2]);
            const float yy2 = fminf(iy2, boxes[j * dim + 3]);
            const float w = fmaxf(0.0, xx2 - xx1 + 1);
            const float h = fmaxf(0.0, yy2 - yy1 + 1);
            const float inter = w * h;
            const float ovr = inter / (iarea + areas[j] - inter);
            if (ovr >= thresh) suppressed[j] = 1;
        }
    }
    free(suppressed);
    return num;
}

/* source coordinate of output index `k` along one axis (crop_and_resize.c:43-58, 85-87) and whether it is inside */
static inline int axis_coord(float a1, float a2, int extent, int crop, int k, float* in)
{
    const float scale = (crop > 1) ? (a2 - a1) * (extent - 1) / (crop - 1) : 0;
    *in = (crop > 1) ? a1 * (extent - 1) + k * scale : 0.5 * (a1 + a2) * (extent - 1);
    return !(*in < 0 || *in > extent - 1);
}

API void mrcnn_crop_forward(const float* image, int batch, int depth, int H, int W, const float* boxes, const int* box_index,
                            int nboxes, float* crops, int ch, int cw, float extrapolation)
{
    const long img_c = (long)H * W, img_e = depth * img_c, crop_c = (long)ch * cw, crop_e = depth * crop_c;
    for (int b = 0; b < nboxes; ++b) {
        const float y1 = boxes[4 * b], x1 = boxes[4 * b + 1], y2 = boxes[4 * b + 2], x2 = boxes[4 * b + 3];
        const int b_in = box_index[b];
        for (int y = 0; y < ch; ++y) {
            float in_y;
            const int y_ok = axis_coord(y1, y2, H, ch, y, &in_y);
            const int top = floorf(in_y), bottom = ceilf(in_y);
            const float y_lerp = in_y - top;
            for (int x = 0; x < cw; ++x) {
                float in_x;
                const int x_ok = axis_coord(x1, x2, W, cw, x, &in_x);
                const int left = floorf(in_x), right = ceilf(in_x);
                const float x_lerp = in_x - left;
                for (int d = 0; d < depth; ++d) {
                    float* out = crops + crop_e * b + crop_c * d + (long)y * cw + x;
                    if (!y_ok || !x_ok) {
                        *out = extrapolation;
                        continue;
                    }
                    const float* p = image + b_in * img_e + d * img_c;
                    const float tl = p[(long)top * W + left], tr = p[(long)top * W + right];
                    const float bl = p[(long)bottom * W + left], br = p[(long)bottom * W + right];
                    const float t = tl + (tr - tl) * x_lerp;
                    const float bt = bl + (br - bl) * x_lerp;
                    *out = t + (bt - t) * y_lerp;
                }
            }
        }
    }
}

API void mrcnn_crop_backward(const float* grads, const float* boxes, const int* box_index, int nboxes, int ch, int cw,
                             float* grads_image, int batch, int depth, int H, int W)
{
    const long img_c = (long)H * W, img_e = depth * img_c, crop_c = (long)ch * cw, crop_e = depth * crop_c;
    memset(grads_image, 0, sizeof(float) * (size_t)batch * img_e);
    for (int b = 0; b < nboxes; ++b) {
        const float y1 = boxes[4 * b], x1 = boxes[4 * b + 1], y2 = boxes[4 * b + 2], x2 = boxes[4 * b + 3];
        const int b_in = box_index[b];
        for (int y = 0; y < ch; ++y) {
            float in_y;
            if (!axis_coord(y1, y2, H, ch, y, &in_y)) continue;
            const int top = floorf(in_y), bottom = ceilf(in_y);
            const float y_lerp = in_y - top;
            for (int x = 0; x < cw; ++x) {
                float in_x;
                if (!axis_coord(x1, x2, W, cw, x, &in_x)) continue;
                const int left = floorf(in_x), right = ceilf(in_x);
                const float x_lerp = in_x - left;
                for (int d = 0; d < depth; ++d) {
                    float* p = grads_image + b_in * img_e + d * img_c;
                    const float g = grads[crop_e * b + crop_c * d + (long)y * cw + x];
                    const float dtop = (1 - y_lerp) * g;
                    p[(long)top * W + left] += (1 - x_lerp) * dtop;
                    p[(long)top * W + right] += x_lerp * dtop;
                    const float dbottom = y_lerp * g;
                    p[(long)bottom * W + left] += (1 - x_lerp) * dbottom;
                    p[(long)bottom * W + right] += x_lerp * dbottom;
                }
            }
        }
    }
}
