"""Textural input assembly on the device: what the reference's VKITTI loader does per item with PIL
and numpy on the host -- textural/data/vkitti_dataset.py:44-142 on top of base_dataset.py:21-110 -- as tensor
operations that run wherever the input tensors live, so that the geometric branch's maps (or the decoded PNGs of its
wire format) become the generator's inputs without a host round trip.

The one non-trivial piece is PIL's resizing.  `resize_u8` reproduces it bit for bit:
  * BICUBIC / BILINEAR (images, normal maps): Pillow's ImagingResample -- per output pixel a source window and weights
    (precompute_coeffs), 22-bit fixed point, horizontal pass rounded to uint8, then vertical;
  * NEAREST (label / instance / pose maps): Pillow's ImagingScaleAffine -- source index = (int) of a running double sum.
The tables are prepared on the host (numpy, cached per size) and the passes are integer gathers and sums on the tensor's
device.  tests/test_assemble.py pins every step against the real PIL of this image; tests/test_gpu_assemble.py runs
the same comparison with CUDA tensors.

`--feat_depth` (the 16-bit PNG branch, vkitti_dataset.py:131-137) is `depth_feature`.  Every division the loader performs
(ToTensor's / 255, the depth branch's / 65535) is a look-up in a table computed on the HOST: torch's device kernels turn
a division by a scalar into a multiplication by its reciprocal, which is not the same rounding.
Not covered: colour jitter, file discovery -- host-side data loading proper stays the caller's business.
"""
import functools
import os
from math import cos, pi, sin

import numpy as np
import torch

PRECISION_BITS = 32 - 8 - 2  # Pillow, Resample.c


def _bicubic(x):
    a = -0.5
    x = np.abs(x)
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1,
                    np.where(x < 2.0, (((x - 5) * x + 8) * x - 4) * a, 0.0))


def _bilinear(x):
    x = np.abs(x)
    return np.where(x < 1.0, 1.0 - x, 0.0)


_FILTERS = {'bicubic': (_bicubic, 2.0), 'bilinear': (_bilinear, 1.0)}


@functools.lru_cache(maxsize=64)
def _resample_table(in_size, out_size, method):
    """Pillow precompute_coeffs + normalize_coeffs_8bpc.  Returns (idx int64 [out, ksize] clamped source indices,
    k8 int64 [out, ksize] fixed-point weights, zero beyond each window)."""
    filt, support0 = _FILTERS[method]
    scale = np.float64(np.float32(in_size) - np.float32(0.0)) / out_size
    filterscale = max(scale, 1.0)
    support = support0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    xx = np.arange(out_size, dtype=np.float64)
    center = 0.0 + (xx + 0.5) * scale
    ss = 1.0 / filterscale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size)
    cnt = xmax - xmin
    kk = np.zeros((out_size, ksize), dtype=np.float64)
    ww = np.zeros(out_size, dtype=np.float64)
    for x in range(ksize):
        w = np.where(x < cnt, filt((x + xmin - center + 0.5) * ss), 0.0)
        kk[:, x] = w
        ww = ww + w
    nz = ww != 0.0
    kk[nz] = kk[nz] / ww[nz, None]
    v = kk * float(1 << PRECISION_BITS)
    k8 = np.where(kk < 0, np.trunc(-0.5 + v), np.trunc(0.5 + v)).astype(np.int64)
    idx = np.minimum(xmin[:, None] + np.arange(ksize)[None, :], in_size - 1)
    return torch.from_numpy(idx), torch.from_numpy(k8)


@functools.lru_cache(maxsize=64)
def _nearest_table(in_size, out_size):
    """Pillow ImagingScaleAffine: the source index of output x is (int) of xo, xo = a/2, a/2 + a, ... summed in double."""
    a = np.float64(in_size) / out_size
    xo = a * 0.5
    idx = np.zeros(out_size, dtype=np.int64)
    for x in range(out_size):
        idx[x] = int(xo)
        xo += a
    return torch.from_numpy(np.minimum(idx, in_size - 1))


_TABLES_ON = {}


def _on(dev, fn, *key):
    """the (host-built, lru-cached) index / weight table fn(*key) as tensors on `dev`, uploaded ONCE per device: `.to(dev)` of a
    pageable host tensor is a synchronous copy queued behind the stream's kernels, i.e. one device synchronisation per resize
    axis -- 14 of them per frame made configs[4] host-paced (208 ms of a 64-frame pass, profiles/r05zb_host_profile_pipe.txt)"""
    if os.environ.get('SDN_ASSEMBLE_UPLOAD_TABLES') == '1':   # A/B switch: the upload per call of r01-r04
        r = fn(*key)
        return tuple(x.to(dev) for x in r) if isinstance(r, tuple) else r.to(dev)
    k = (fn.__name__, str(dev)) + key
    t = _TABLES_ON.get(k)
    if t is None:
        if len(_TABLES_ON) > 256:
            _TABLES_ON.clear()
        r = fn(*key)
        t = _TABLES_ON[k] = tuple(x.to(dev) for x in r) if isinstance(r, tuple) else r.to(dev)
    return t


def resize_nearest(img, oh, ow):
    """PIL NEAREST for any pixel type ([C, H, W]): pure index selection."""
    C, H, W = img.shape
    if (oh, ow) == (H, W):
        return img.clone()
    dev = img.device
    return img[:, _on(dev, _nearest_table, H, oh)][:, :, _on(dev, _nearest_table, W, ow)]


def _resize(img, oh, ow, method):
    return resize_nearest(img, oh, ow) if method == 'nearest' else resize_u8(img, oh, ow, method)


def resize_u8(img, oh, ow, method):
    """img uint8 [C, H, W] on any device -> uint8 [C, oh, ow], bit-identical to PIL.Image.resize((ow, oh), method)."""
    if img.dtype != torch.uint8 or img.dim() != 3:
        raise TypeError('resize_u8 expects a uint8 [C, H, W] tensor')
    C, H, W = img.shape
    if (oh, ow) == (H, W):
        return img.clone()
    dev = img.device
    if method == 'nearest':
        return resize_nearest(img, oh, ow)
    half = 1 << (PRECISION_BITS - 1)
    cur = img.to(torch.int64)
    if ow != W:  # horizontal pass, rounded and clipped to the pixel type
        idx, k8 = _on(dev, _resample_table, W, ow, method)
        acc = (cur[:, :, idx] * k8).sum(dim=3) + half
        cur = (acc >> PRECISION_BITS).clamp_(0, 255)
    if oh != H:
        idx, k8 = _on(dev, _resample_table, H, oh, method)
        acc = (cur[:, idx] * k8[None, :, :, None]).sum(dim=2) + half
        cur = (acc >> PRECISION_BITS).clamp_(0, 255)
    return cur.to(torch.uint8)


@functools.lru_cache(maxsize=1)
def _to_tensor_lut():
    """float32(k) / float32(255), correctly rounded, for k = 0..255 -- computed on the HOST.  torch's device kernels
    evaluate `x.div(255)` as `x * (1 / 255)`, which differs from the division ToTensor performs on the CPU for many k
    (and then k / 255 * 255 != k: a label of 6.9999995 truncates to class 6 in Pix2PixHDModel's one-hot encoding)."""
    return torch.from_numpy((np.arange(256, dtype=np.float32) / np.float32(255.0)).astype(np.float32))


@functools.lru_cache(maxsize=2)
def _depth_lut(wrap_int16):
    """1.0 - float32(d) / 65535.0 for the 65536 values of a 16-bit depth PNG (vkitti_dataset.py:131-137), computed on
    the host with the CPU loader's own operations.  wrap_int16: torchvision 0.2.x's ToTensor reads a mode-'I;16' image
    through np.int16 (values >= 32768 come out negative); PIL versions that open 16-bit PNGs as mode 'I' do not."""
    d = np.arange(65536, dtype=np.int64)
    if wrap_int16:
        d = d.astype(np.uint16).view(np.int16).astype(np.int64)
    t = torch.from_numpy(d).float() / 65535.0
    return 1.0 - t


_LUT_ON = {}


def to_tensor_u8(img):
    """torchvision's ToTensor arithmetic (uint8 -> float32, / 255) for a uint8 tensor on any device, bit-identical to the
    CPU result: a 256-entry table look-up instead of a device-side division."""
    if img.dtype != torch.uint8:
        raise TypeError('to_tensor_u8 expects uint8')
    lut = _LUT_ON.get(img.device)
    if lut is None:
        lut = _LUT_ON[img.device] = _to_tensor_lut().to(img.device)
    return lut[img.long()]


# ---------------------------------------------------------------------------------------------------------------------
# base_dataset.py: get_transform as a function of (options, params) on a uint8 [C, H, W] tensor
def load_size_after_scaling(opt, h, w):
    """(h, w) after the resize step of get_transform (base_dataset.py:45-49, 82-91)."""
    roc = opt.resize_or_crop
    if 'resize' in roc:
        return opt.loadSize, opt.loadSize
    if 'scale_width' in roc:
        if w == opt.loadSize:
            return h, w
        nh = int(opt.loadSize * h / w)
        return (192 if nh == 188 else nh), opt.loadSize   # the reference's hack for 375 x 1242 -> 192 x 624
    return h, w


def _geometry(img, opt, params, method):
    """The PIL-image part of get_transform (base_dataset.py:41-61): resize / scale width, crop, make_power_2, flip, on an
    integer [C, H, W] tensor (uint8 for every method; any integer type for 'nearest', which only moves pixels)."""
    C, H, W = img.shape
    roc = opt.resize_or_crop
    oh, ow = load_size_after_scaling(opt, H, W)
    img = _resize(img, oh, ow, method)
    if 'crop' in roc:
        x1, y1 = params['crop_pos']
        tw, th = opt.fineWidth, opt.fineHeight
        if ow > tw or oh > th:
            out = torch.zeros(C, th, tw, dtype=img.dtype, device=img.device)   # PIL pads a crop box beyond the image
            ys, xs = max(0, min(th, oh - y1)), max(0, min(tw, ow - x1))
            out[:, :ys, :xs] = img[:, y1:y1 + ys, x1:x1 + xs]
            img = out
    if roc == 'none':
        base = float(2 ** opt.n_downsample_global)
        if opt.netG == 'local':
            base *= (2 ** opt.n_local_enhancers)
        h2, w2 = int(round(img.shape[1] / base) * base), int(round(img.shape[2] / base) * base)
        img = _resize(img, h2, w2, method)
    if opt.isTrain and not opt.no_flip and params['flip']:
        img = torch.flip(img, dims=(2,))
    return img


def transform(img, opt, params, method='bicubic', normalize=True):
    """get_transform(opt, params, method, normalize)(PIL image) for a uint8 [C, H, W] tensor (base_dataset.py:41-66):
    the image geometry above, then ToTensor (/ 255) and Normalize((x - 0.5) / 0.5)."""
    t = to_tensor_u8(_geometry(img, opt, params, method))
    if normalize:
        t = (t - 0.5) / 0.5   # exact on every device: x - 0.5 is one IEEE subtraction, / 0.5 is a multiplication by 2
    return t


def pose_bins(num_bins):
    return np.array(list(range(-180, 181, 360 // num_bins))) / 180


def depth_feature(depth, opt, params, wrap_int16=False):
    """The `--feat_depth` branch (vkitti_dataset.py:131-137): the geometric branch's NNNNN-depth.png (16-bit, values
    0..65535, integer [1, H, W] tensor of any integer type wide enough) through transform_A's geometry (NEAREST), then
    1.0 - float(d) / 65535.0.  The division is a host-built table look-up, for the reason given at to_tensor_u8."""
    if depth.dtype in (torch.float16, torch.float32, torch.float64, torch.bfloat16):
        raise TypeError('depth_feature expects the integer pixel values of the 16-bit PNG')
    g = _geometry(depth, opt, params, 'nearest').long()
    if int(g.min()) < 0 or int(g.max()) > 65535:
        raise ValueError('depth values outside 0..65535')
    key = ('depth', bool(wrap_int16), g.device)
    lut = _LUT_ON.get(key)
    if lut is None:
        lut = _LUT_ON[key] = _depth_lut(bool(wrap_int16)).to(g.device)
    return lut[g]


def assemble_item(opt, params, segm, image, inst=None, pose_inst=None, pose_json=None, normal=None, depth=None,
                  depth_wrap_int16=False):
    """vkitti_dataset.__getitem__ (:44-142) from already decoded maps, all uint8 [C, H, W] tensors on one device:
    segm [1,H,W] label ids, image [3,H,W] RGB, inst [1,H,W] instance ids, pose_inst [1,H,W] + pose_json (the geometric
    branch's NNNNN.png / NNNNN.json), normal [3,H,W] (NNNNN-normal.png).  Returns the loader's dict entries
    label / inst / image / pose / normal (0 where the loader leaves its default)."""
    out = {'label': 0, 'inst': 0, 'image': 0, 'pose': 0, 'normal': 0, 'feat': 0, 'depth': 0}
    if opt.label_nc == 0:
        A = transform(segm.expand(3, -1, -1) if segm.shape[0] == 1 else segm, opt, params)
    else:
        A = transform(segm, opt, params, method='nearest', normalize=False) * 255.0
    if opt.segm_precomputed_path:
        A = A + 1
    out['image'] = transform(image, opt, params)
    if not opt.no_instance:
        if inst is None:
            inst_t = A
        else:
            inst_t = transform(inst, opt, params, method='nearest', normalize=False)
            if opt.inst_precomputed_path:
                inst_t = inst_t * 255.0
                inst_t = inst_t * 1000
                if opt.segm_precomputed_path:   # car labels without an instance become "misc" (:75-78)
                    A = A.clone()
                    A[(inst_t == 0) & (A == 2)] = 5
                    A[(inst_t == 0) & (A == 12)] = 5
                inst_t = torch.where(inst_t == 0, A, inst_t)
        out['inst'] = inst_t
    out['label'] = A
    if opt.feat_pose:
        nb = opt.feat_pose_num_bins
        H, W = A.shape[1], A.shape[2]
        if nb > 0:
            pose = torch.zeros(1, H, W, dtype=torch.float64, device=A.device)
        else:
            pose = torch.zeros(2, H, W, dtype=torch.float64, device=A.device)
        if pose_inst is not None and pose_json is not None:
            inst_map = (transform(pose_inst, opt, params, method='nearest', normalize=False) * 255.0)[0]
            bins = pose_bins(nb) if nb else None
            for key, rec in pose_json.items():      # the loader walks np.unique(inst_map); ids absent from the map
                k = int(key)                        # give empty masks here
                if k == 0:
                    continue
                m = inst_map == float(k)
                alpha = rec['alpha']
                if nb > 0:
                    pose[0][m] = float(np.digitize(alpha / pi, bins))
                else:
                    pose[0][m] = cos(alpha)
                    pose[1][m] = sin(alpha)
        out['pose'] = pose.int() if nb else pose.float()
    if opt.feat_normal:
        if normal is not None:
            out['normal'] = transform(normal, opt, params) + 1 / 255   # "bias caused by 0..256 instead of 0..255" (:125)
        else:
            out['normal'] = torch.zeros_like(out['image'])
    if getattr(opt, 'feat_depth', None):
        out['depth'] = depth_feature(depth, opt, params, depth_wrap_int16) if depth is not None else torch.zeros_like(A)
    return out
