"""The Mask R-CNN op oracle (oracle/maskrcnn_oracle.c) against the reference's own nms.c / crop_and_resize.c compiled
unmodified (oracle/_ref/libmaskrcnn_ref.so): bit equality on seeded inputs, plus known answers."""
import numpy as np
import pytest

from oracle import maskrcnn_np as mn
from util import biteq

needs_ref = pytest.mark.skipif(not mn.have('ref'), reason='oracle/_ref/libmaskrcnn_ref.so not built (no reference checkout)')


def random_dets(rng, n, extent=200.0):
    y1, x1 = rng.uniform(0, extent, n), rng.uniform(0, extent, n)
    h, w = rng.uniform(2, 60, n), rng.uniform(2, 60, n)
    return np.stack([y1, x1, y1 + h, x1 + w, rng.permutation(n) / n + rng.uniform(0, 1e-3, n)], 1).astype(np.float32)


@needs_ref
@pytest.mark.parametrize('n,thresh', [(1, 0.5), (17, 0.3), (300, 0.5), (1500, 0.7), (600, 0.0)])
def test_nms_oracle_equals_reference(n, thresh):
    dets = random_dets(np.random.default_rng(n), n)
    a, b = mn.pth_nms(dets, thresh, 'oracle'), mn.pth_nms(dets, thresh, 'ref')
    assert a.dtype == np.int64 and np.array_equal(a, b)
    assert 1 <= len(a) <= n


def test_nms_known_answers():
    dets = np.array([[0, 0, 9, 9, 0.9], [0, 0, 9, 9, 0.8], [0, 5, 9, 14, 0.7], [20, 20, 29, 29, 0.6]], np.float32)
    # identical box suppressed; the shifted box overlaps 50 / 150 = 1/3; the far box survives
    assert mn.pth_nms(dets, 0.5).tolist() == [0, 2, 3]
    assert mn.pth_nms(dets, 0.3).tolist() == [0, 3]
    assert mn.pth_nms(dets, 1.0 / 3.0).tolist() in ([0, 3], [0, 2, 3])   # ovr >= thresh at the rounding boundary
    assert mn.pth_nms(dets[:0], 0.5).tolist() == []


def crop_case(rng, B, C, H, W, n, ch, cw):
    image = rng.normal(size=(B, C, H, W)).astype(np.float32)
    y1, x1 = rng.uniform(-0.2, 0.9, n), rng.uniform(-0.2, 0.9, n)
    boxes = np.stack([y1, x1, y1 + rng.uniform(0.05, 0.6, n), x1 + rng.uniform(0.05, 0.6, n)], 1).astype(np.float32)
    boxes[0] = (0.0, 0.0, 1.0, 1.0)                       # the whole image: corner samples hit integer coordinates
    if n > 1:
        boxes[1] = (0.25, 0.5, 0.25, 0.5)                 # degenerate box
    return image, boxes, rng.integers(0, B, n).astype(np.int32)


@needs_ref
@pytest.mark.parametrize('shape', [(2, 3, 17, 23, 9, 7, 7), (1, 8, 32, 32, 30, 14, 14), (3, 2, 5, 9, 6, 1, 4), (1, 1, 8, 8, 4, 1, 1)])
def test_crop_and_resize_oracle_equals_reference(shape):
    B, C, H, W, n, ch, cw = shape
    rng = np.random.default_rng(sum(shape))
    image, boxes, idx = crop_case(rng, B, C, H, W, n, ch, cw)
    a = mn.crop_forward(image, boxes, idx, ch, cw, -1.5, 'oracle')
    b = mn.crop_forward(image, boxes, idx, ch, cw, -1.5, 'ref')
    assert biteq(a, b)
    g = rng.normal(size=a.shape).astype(np.float32)
    assert biteq(mn.crop_backward(g, boxes, idx, image.shape, 'oracle'), mn.crop_backward(g, boxes, idx, image.shape, 'ref'))


def test_crop_and_resize_known_answers():
    image = np.arange(2 * 4 * 5, dtype=np.float32).reshape(1, 2, 4, 5)
    full = np.array([[0, 0, 1, 1]], np.float32)
    out = mn.crop_forward(image, full, np.zeros(1, np.int32), 4, 5)
    assert biteq(out, image)                               # identity sampling grid
    half = mn.crop_forward(image, full, np.zeros(1, np.int32), 2, 2)
    assert half[0, 0].tolist() == [[0.0, 4.0], [15.0, 19.0]]   # the four corners
    outside = mn.crop_forward(image, np.array([[1.5, 1.5, 2.0, 2.0]], np.float32), np.zeros(1, np.int32), 3, 3, 7.0)
    assert (outside == 7.0).all()
    g = np.ones((1, 2, 4, 5), np.float32)
    assert biteq(mn.crop_backward(g, full, np.zeros(1, np.int32), image.shape), g)
