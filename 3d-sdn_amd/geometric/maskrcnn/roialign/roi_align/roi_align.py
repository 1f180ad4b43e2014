"""RoIAlign on top of crop_and_resize (reference: geometric/maskrcnn/roialign/roi_align/roi_align.py:7-48)."""
import torch
from torch import nn

from .crop_and_resize import CropAndResize, CropAndResizeFunction  # noqa: F401


class RoIAlign(nn.Module):
    def __init__(self, crop_height, crop_width, extrapolation_value=0, transform_fpcoor=True):
        super(RoIAlign, self).__init__()
        self.crop_height = crop_height
        self.crop_width = crop_width
        self.extrapolation_value = extrapolation_value
        self.transform_fpcoor = transform_fpcoor

    def forward(self, featuremap, boxes, box_ind):
        """featuremap [N,C,H,W]; boxes [M,4] = (x1, y1, x2, y2) in pixels; box_ind [M] -> [M,C,crop_height,crop_width]."""
        x1, y1, x2, y2 = torch.split(boxes, 1, dim=1)
        image_height, image_width = featuremap.size()[2:4]
        if self.transform_fpcoor:   # sample at bin centres (roi_align.py:27-36)
            spacing_w = (x2 - x1) / float(self.crop_width)
            spacing_h = (y2 - y1) / float(self.crop_height)
            nx0 = (x1 + spacing_w / 2 - 0.5) / float(image_width - 1)
            ny0 = (y1 + spacing_h / 2 - 0.5) / float(image_height - 1)
            nw = spacing_w * float(self.crop_width - 1) / float(image_width - 1)
            nh = spacing_h * float(self.crop_height - 1) / float(image_height - 1)
            boxes = torch.cat((ny0, nx0, ny0 + nh, nx0 + nw), 1)
        else:
            boxes = torch.cat((y1 / float(image_height - 1), x1 / float(image_width - 1), y2 / float(image_height - 1),
                               x2 / float(image_width - 1)), 1)
        return CropAndResizeFunction(self.crop_height, self.crop_width, self.extrapolation_value)(
            featuremap, boxes.detach().contiguous(), box_ind.detach())
