"""numpy front end of the Mask R-CNN op checkers (TEST INFRASTRUCTURE ONLY):
  impl='oracle'  oracle/libmaskrcnn_oracle.so  -- our C restatement (oracle/maskrcnn_oracle.c)
  impl='ref'     oracle/_ref/libmaskrcnn_ref.so -- the reference's nms.c / crop_and_resize.c compiled unmodified against
                 oracle/th_shim (build container only; the built .so travels to the GPU box)
plus `pth_nms` = the CPU branch of geometric/maskrcnn/nms/pth_nms.py:10-26 on top of either."""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_libs = {}


class _TH(ctypes.Structure):
    _fields_ = [('data', ctypes.c_void_p), ('size', ctypes.c_long * 4), ('ndim', ctypes.c_int), ('capacity', ctypes.c_long)]


def _th(a):
    t = _TH()
    t.data = a.ctypes.data
    for i, s in enumerate(a.shape):
        t.size[i] = s
    t.ndim = a.ndim
    t.capacity = a.size
    return t


def have(impl):
    path = os.path.join(HERE, 'libmaskrcnn_oracle.so') if impl == 'oracle' else os.path.join(HERE, '_ref', 'libmaskrcnn_ref.so')
    return os.path.exists(path)


def _lib(impl):
    if impl not in _libs:
        path = os.path.join(HERE, 'libmaskrcnn_oracle.so') if impl == 'oracle' else \
            os.path.join(HERE, '_ref', 'libmaskrcnn_ref.so')
        _libs[impl] = ctypes.CDLL(path)
        if impl == 'oracle':
            _libs[impl].mrcnn_nms.restype = ctypes.c_long
    return _libs[impl]


def cpu_nms(boxes, order, areas, thresh, impl='oracle'):
    """nms.c:4-69.  boxes float32 [n, dim >= 4], order int64 [n], areas float32 [n] -> kept indices (int64)."""
    boxes = np.ascontiguousarray(boxes, np.float32)
    order = np.ascontiguousarray(order, np.int64)
    areas = np.ascontiguousarray(areas, np.float32)
    n = boxes.shape[0]
    keep = np.zeros(max(n, 1), np.int64)
    if impl == 'oracle':
        num = _lib(impl).mrcnn_nms(boxes.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(n), ctypes.c_long(boxes.shape[1]),
                                   order.ctypes.data_as(ctypes.c_void_p), areas.ctypes.data_as(ctypes.c_void_p),
                                   ctypes.c_float(thresh), keep.ctypes.data_as(ctypes.c_void_p))
    else:
        num_out = np.zeros(1, np.int64)
        tk, tn, tb, to, ta = _th(keep), _th(num_out), _th(boxes), _th(order), _th(areas)
        _lib(impl).cpu_nms(ctypes.byref(tk), ctypes.byref(tn), ctypes.byref(tb), ctypes.byref(to), ctypes.byref(ta),
                           ctypes.c_float(thresh))
        num = int(num_out[0])
    return keep[:num].copy()


def pth_nms(dets, thresh, impl='oracle'):
    """pth_nms.py:10-26 (CPU branch): dets [n,5] = (y1, x1, y2, x2, score); areas with the +1 convention; stable
    descending score order (torch's CPU sort of distinct scores)."""
    dets = np.ascontiguousarray(dets, np.float32)
    x1, y1, x2, y2 = dets[:, 1], dets[:, 0], dets[:, 3], dets[:, 2]
    areas = (x2 - x1 + np.float32(1)) * (y2 - y1 + np.float32(1))
    order = np.argsort(-dets[:, 4], kind='stable')
    return cpu_nms(dets, order, areas, thresh, impl)


def crop_forward(image, boxes, box_index, ch, cw, extrapolation=0.0, impl='oracle'):
    """crop_and_resize.c:116-158.  image [B,C,H,W], boxes [n,4] = (y1,x1,y2,x2) normalised, box_index int32 [n]."""
    image = np.ascontiguousarray(image, np.float32)
    boxes = np.ascontiguousarray(boxes, np.float32)
    box_index = np.ascontiguousarray(box_index, np.int32)
    B, C, H, W = image.shape
    n = boxes.shape[0]
    crops = np.zeros((n, C, ch, cw), np.float32)
    if impl == 'oracle':
        _lib(impl).mrcnn_crop_forward(image.ctypes.data_as(ctypes.c_void_p), B, C, H, W, boxes.ctypes.data_as(ctypes.c_void_p),
                                      box_index.ctypes.data_as(ctypes.c_void_p), n, crops.ctypes.data_as(ctypes.c_void_p),
                                      ch, cw, ctypes.c_float(extrapolation))
    else:
        ti, tb, tx, tc = _th(image), _th(boxes), _th(box_index), _th(crops)
        _lib(impl).crop_and_resize_forward(ctypes.byref(ti), ctypes.byref(tb), ctypes.byref(tx), ctypes.c_float(extrapolation),
                                           ch, cw, ctypes.byref(tc))
    return crops


def crop_backward(grads, boxes, box_index, image_shape, impl='oracle'):
    """crop_and_resize.c:160-251.  grads [n,C,ch,cw] -> grads_image [B,C,H,W] (serial accumulation order)."""
    grads = np.ascontiguousarray(grads, np.float32)
    boxes = np.ascontiguousarray(boxes, np.float32)
    box_index = np.ascontiguousarray(box_index, np.int32)
    B, C, H, W = image_shape
    n, _, ch, cw = grads.shape
    gi = np.zeros((B, C, H, W), np.float32)
    if impl == 'oracle':
        _lib(impl).mrcnn_crop_backward(grads.ctypes.data_as(ctypes.c_void_p), boxes.ctypes.data_as(ctypes.c_void_p),
                                       box_index.ctypes.data_as(ctypes.c_void_p), n, ch, cw,
                                       gi.ctypes.data_as(ctypes.c_void_p), B, C, H, W)
    else:
        tg, tb, tx, ti = _th(grads), _th(boxes), _th(box_index), _th(gi)
        _lib(impl).crop_and_resize_backward(ctypes.byref(tg), ctypes.byref(tb), ctypes.byref(tx), ctypes.byref(ti))
    return gi
