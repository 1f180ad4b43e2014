"""Mask R-CNN's custom ops on the HIP kernels against the oracle (oracle/maskrcnn_oracle.c, pinned bit
for bit to the reference's nms.c / crop_and_resize.c by tests/test_maskrcnn_oracle.py): kept indices identical, crops
bit-identical, crop gradient (atomic scatter-add) 1e-6 relative."""
import numpy as np
import pytest
import torch

from oracle import maskrcnn_np as mn
from test_maskrcnn_oracle import crop_case, random_dets
from util import biteq

pytestmark = pytest.mark.gpu
MR = 'maskrcnn'


def _ops():
    import os
    import sys
    p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), '3d-sdn_amd', 'geometric', MR)
    if p not in sys.path:
        sys.path.insert(0, p)       # the reference imports `nms.nms_wrapper`, `roialign.roi_align...` from this directory
    from nms.nms_wrapper import nms
    from roialign.roi_align.crop_and_resize import CropAndResize, CropAndResizeFunction
    from roialign.roi_align.roi_align import RoIAlign
    return nms, CropAndResizeFunction, CropAndResize, RoIAlign


@pytest.mark.parametrize('n,thresh', [(1, 0.5), (17, 0.3), (64, 0.5), (65, 0.5), (300, 0.5), (1500, 0.7), (6000, 0.7), (600, 0.0)])
def test_nms_keeps_the_same_boxes(n, thresh):
    nms = _ops()[0]
    dets = random_dets(np.random.default_rng(n), n)
    ref = mn.pth_nms(dets, thresh)      # the oracle = the reference's C path: IoU >= thresh
    from nms.pth_nms import pth_nms
    got = pth_nms(torch.tensor(dets).cuda(), thresh, strict=False)
    assert got.dtype == torch.int64 and got.is_cuda
    assert np.array_equal(got.cpu().numpy(), ref)
    if thresh > 0:   # random boxes have no exact ties: the default rule (IoU > thresh, the reference's CUDA kernel) keeps the same
        assert np.array_equal(nms(torch.tensor(dets).cuda(), thresh).cpu().numpy(), ref)


def test_nms_edge_cases():
    nms = _ops()[0]
    assert nms(torch.zeros(0, 5).cuda(), 0.5).numel() == 0
    dets = np.array([[0, 0, 9, 9, 0.9], [0, 0, 9, 9, 0.8], [0, 5, 9, 14, 0.7], [20, 20, 29, 29, 0.6]], np.float32)
    assert nms(torch.tensor(dets).cuda(), 0.5).tolist() == [0, 2, 3]
    assert nms(torch.tensor(dets).cuda(), 0.3).tolist() == [0, 3]
    with pytest.raises(NotImplementedError):
        nms(torch.tensor(dets), 0.5)
    # an exact tie: the second box lies inside the first with half its area, IoU = 50 / 100 = 0.5 == thresh.  The default is
    # the rule the reference reaches on CUDA tensors (nms_kernel.cu:63-66: suppress on IoU > thresh): the box stays;
    # strict=False is cpu_nms's IoU >= thresh (nms.c:59): it goes.
    from nms.pth_nms import pth_nms
    tie = np.array([[0, 0, 9, 9, 0.9], [0, 0, 9, 4, 0.8], [30, 30, 39, 39, 0.7]], np.float32)
    assert nms(torch.tensor(tie).cuda(), 0.5).tolist() == [0, 1, 2]
    assert pth_nms(torch.tensor(tie).cuda(), 0.5, strict=False).tolist() == [0, 2]
    assert mn.pth_nms(tie, 0.5).tolist() == [0, 2]


@pytest.mark.parametrize('shape', [(2, 3, 17, 23, 9, 7, 7), (1, 8, 32, 32, 30, 14, 14), (3, 2, 5, 9, 6, 1, 4), (1, 1, 8, 8, 4, 1, 1),
                                   (2, 256, 48, 156, 200, 7, 7)])
def test_crop_and_resize_matches_the_c_path(shape):
    _, CropAndResizeFunction, CropAndResize, _ = _ops()
    B, C, H, W, n, ch, cw = shape
    rng = np.random.default_rng(sum(shape))
    image, boxes, idx = crop_case(rng, B, C, H, W, n, ch, cw)
    ref = mn.crop_forward(image, boxes, idx, ch, cw, -1.5)
    im = torch.tensor(image).cuda().requires_grad_(True)
    out = CropAndResizeFunction(ch, cw, -1.5)(im, torch.tensor(boxes).cuda(), torch.tensor(idx).cuda())
    assert biteq(out.detach().cpu().numpy(), ref)
    g = rng.normal(size=ref.shape).astype(np.float32)
    out.backward(torch.tensor(g).cuda())
    gref = mn.crop_backward(g, boxes, idx, image.shape)
    rel = np.linalg.norm(im.grad.cpu().numpy().astype(np.float64) - gref) / max(np.linalg.norm(gref), 1e-30)
    assert rel <= 1e-6, rel
    mod = CropAndResize(ch, cw, -1.5)(torch.tensor(image).cuda(), torch.tensor(boxes).cuda(), torch.tensor(idx).cuda())
    assert torch.equal(mod, out.detach())


def test_roi_align_identity_grid():
    RoIAlign = _ops()[3]
    fm = torch.arange(2 * 6 * 8, dtype=torch.float32).reshape(1, 2, 6, 8).cuda()
    # one bin per pixel over the whole map (pixel i spans [i, i + 1)): bin centres fall on the sample grid
    out = RoIAlign(6, 8)(fm, torch.tensor([[0.0, 0.0, 8.0, 6.0]]).cuda(), torch.zeros(1, dtype=torch.int32).cuda())
    assert torch.allclose(out, fm, atol=1e-5)
