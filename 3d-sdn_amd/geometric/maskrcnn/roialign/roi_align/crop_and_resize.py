"""crop_and_resize (geometric/maskrcnn/roialign/roi_align/crop_and_resize.py:10-66) on csrc/raster_boxes.hip.

`CropAndResizeFunction(crop_height, crop_width, extrapolation_value)(image, boxes, box_ind)` keeps the reference's call
shape (an object configured with the crop size, then called): the old-style autograd.Function it was there cannot exist in
current torch, so the object dispatches to a static Function.  image [B,C,H,W] float32, boxes [n,4] = (y1,x1,y2,x2)
normalised, box_ind [n] int32 -> crops [n,C,crop_height,crop_width]; differentiable wrt image (as the reference).
(The reference's forward allocates `torch.zeros_like(image)` and lets the C code resize it; the result is the same tensor.)"""
import os

import torch
import torch.nn as nn

from sdn_hip import check, lib, ptr, stream, want


class _CropAndResize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, boxes, box_ind, crop_height, crop_width, extrapolation_value):
        image = want(image, torch.float32, 'image')
        boxes = want(boxes, torch.float32, 'boxes')
        box_ind = want(box_ind, torch.int32, 'box_ind')
        if image.dim() != 4 or boxes.dim() != 2 or boxes.shape[1] != 4 or box_ind.numel() != boxes.shape[0]:
            raise ValueError('image [B,C,H,W], boxes [n,4], box_ind [n]')
        B, C, H, W = image.shape
        n = boxes.shape[0]
        # An out-of-range box index: the reference's C code prints and exits (crop_and_resize.c:39-42); the kernels here are
        # safe without a check (forward writes the extrapolation value for such a box, backward skips it).  The check costs
        # two host synchronisations per call -- RoIAlign runs once per FPN level for two heads, ~8 stalls per image -- so it
        # is a debug switch: SDN_DEBUG_CHECKS=1.
        if n and os.environ.get('SDN_DEBUG_CHECKS') == '1' and (int(box_ind.min()) < 0 or int(box_ind.max()) >= B):
            raise IndexError('box_ind out of range [0, %d)' % B)
        crops = torch.empty(n, C, crop_height, crop_width, dtype=torch.float32, device=image.device)
        check(lib().sdn_crop_and_resize_fwd(ptr(image), B, C, H, W, ptr(boxes), ptr(box_ind), n, int(crop_height),
                                            int(crop_width), float(extrapolation_value), ptr(crops), stream()))
        ctx.save_for_backward(boxes, box_ind)
        ctx.im_size = (B, C, H, W)
        return crops

    @staticmethod
    def backward(ctx, grad_outputs):
        boxes, box_ind = ctx.saved_tensors
        B, C, H, W = ctx.im_size
        g = grad_outputs.contiguous()
        n, _, ch, cw = g.shape
        grad_image = torch.empty(B, C, H, W, dtype=torch.float32, device=g.device)
        check(lib().sdn_crop_and_resize_bwd(ptr(g), ptr(boxes), ptr(box_ind), n, ch, cw, ptr(grad_image), B, C, H, W,
                                            stream()))
        return grad_image, None, None, None, None, None


class CropAndResizeFunction(object):
    def __init__(self, crop_height, crop_width, extrapolation_value=0):
        self.crop_height = crop_height
        self.crop_width = crop_width
        self.extrapolation_value = extrapolation_value

    def __call__(self, image, boxes, box_ind):
        return _CropAndResize.apply(image, boxes, box_ind, self.crop_height, self.crop_width, self.extrapolation_value)

    forward = __call__


class CropAndResize(nn.Module):
    """Crop and resize as tf.image.crop_and_resize (reference: crop_and_resize.py:52-66)."""

    def __init__(self, crop_height, crop_width, extrapolation_value=0):
        super(CropAndResize, self).__init__()
        self.crop_height = crop_height
        self.crop_width = crop_width
        self.extrapolation_value = extrapolation_value

    def forward(self, image, boxes, box_ind):
        return CropAndResizeFunction(self.crop_height, self.crop_width, self.extrapolation_value)(image, boxes, box_ind)
