"""geometric/maskrcnn/nms/nms_wrapper.py: `nms(dets, thresh)`."""
from .pth_nms import pth_nms


def nms(dets, thresh):
    """Dispatch as the reference does (nms_wrapper.py:13-16); dets [n,5] = (y1, x1, y2, x2, score) tensor."""
    return pth_nms(dets, thresh)
