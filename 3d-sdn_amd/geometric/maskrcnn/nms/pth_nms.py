"""pth_nms (geometric/maskrcnn/nms/pth_nms.py:6-54) on the HIP kernels of csrc/raster_boxes.hip.

dets [n, 5] = (y1, x1, y2, x2, score) float32 CUDA tensor -> LongTensor of kept row indices, best score first.
Boxes are visited in descending score order; a box is dropped when its IoU (+1 pixel convention) with an already kept box
exceeds thresh.  Which comparison: only CUDA tensors reach this op, and on CUDA the reference runs nms_kernel.cu, which
suppresses on IoU > thresh (cuda/nms_kernel.cu:63-66); both reference callers (maskrcnn/model.py:394, :815) pass
score-sorted dets, so the reference's reachable rule is `>` in score order: `strict=True`, the default.  `strict=False`
selects the CPU path's IoU >= thresh (cpu_nms, nms/src/nms.c:59); the two differ only on exact ties
(tests/test_gpu_maskrcnn_ops.py has a tie case).  NOTE on the reference's GPU branch (pth_nms.py:27-52): it hands the
kernel `dets_temp`, a copy made BEFORE the score sort, so for UNSORTED input its suppression runs in input order and
`order[keep]` then indexes a different permutation -- a defect that its callers never trigger, not a contract.
The whole pass (score sort, pair mask, greedy scan) stays on the device: one host read of the kept count at the end, where
the reference copies the n x n/64 mask to the host.  CPU tensors raise NotImplementedError (no fallback)."""
import ctypes

import torch

from sdn_hip import check, lib, ptr, stream, want


def pth_nms(dets, thresh, strict=True):
    dets = want(dets, torch.float32, 'dets')
    if dets.dim() != 2 or dets.shape[1] < 5:
        raise ValueError('dets must be [n, 5] = (y1, x1, y2, x2, score)')
    n = dets.shape[0]
    if n == 0:
        return torch.empty(0, dtype=torch.int64, device=dets.device)
    scores = dets[:, 4]
    order = scores.sort(dim=0, descending=True, stable=True)[1]       # pth_nms.py:19
    boxes = dets[order, :4].contiguous()
    # areas as pth_nms.py:18 computes them: (x2 - x1 + 1) * (y2 - y1 + 1), float32
    areas = ((boxes[:, 3] - boxes[:, 1] + 1) * (boxes[:, 2] - boxes[:, 0] + 1)).contiguous()
    keep = torch.empty(n, dtype=torch.int64, device=dets.device)
    count = torch.empty(1, dtype=torch.int64, device=dets.device)
    nbytes = ctypes.c_size_t(0)
    check(lib().sdn_nms_workspace_bytes(n, ctypes.byref(nbytes)))
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dets.device)
    check(lib().sdn_nms(ptr(boxes), ptr(areas), n, float(thresh), int(bool(strict)), ptr(keep), ptr(count), ptr(ws),
                        ws.numel(), stream()))
    return order[keep[:int(count.item())]].contiguous()
