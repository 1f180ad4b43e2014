"""numpy/ctypes front-end of the CPU rasterizer oracle.  TEST INFRASTRUCTURE ONLY.

Two interchangeable back-ends, same call signatures:
  impl='oracle' : oracle/libraster_oracle.so  -- our restatement (raster_oracle.c)
  impl='ref'    : oracle/_ref/libnr_ref.so    -- the reference's own kernel strings compiled for
                  the CPU by oracle/build_ref.py (present only after that script ran where
                  /root/reference exists; the built .so travels with the repo snapshot).

The host code below restates Rasterize.forward_gpu / backward_gpu
(/root/reference/geometric/neural_renderer/rasterize.py:464-510, 846-886): buffer
initialisation, kernel order, which maps exist for which return_* flags.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_F = ctypes.POINTER(ctypes.c_float)
_I = ctypes.POINTER(ctypes.c_int32)
_LIB = None
_REF = None


def _fp(a):
    return a.ctypes.data_as(_F) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(_I) if a is not None else None


def build(force=False):
    """Compile the C restatement (and _ref when the reference tree is present)."""
    so = os.path.join(_HERE, 'libraster_oracle.so')
    src = os.path.join(_HERE, 'raster_oracle.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, 'libraster_oracle.so'], stdout=subprocess.DEVNULL)
    ref_so = os.path.join(_HERE, '_ref', 'libnr_ref.so')
    if os.path.exists('/root/reference') and (force or not os.path.exists(ref_so)):
        subprocess.check_call(['python3', os.path.join(_HERE, 'build_ref.py')], stdout=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is None:
        build()
        _LIB = ctypes.CDLL(os.path.join(_HERE, 'libraster_oracle.so'))
        _LIB.orc_num_threads.restype = ctypes.c_int
    return _LIB


def have_ref():
    return os.path.exists(os.path.join(_HERE, '_ref', 'libnr_ref.so'))


def ref():
    global _REF
    if _REF is None:
        _REF = ctypes.CDLL(os.path.join(_HERE, '_ref', 'libnr_ref.so'))
    return _REF


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


def _ref_params(nf, is_, ts, rr, ra, rd, near, far, eps):
    # order fixed by build_ref.PARAM_TYPES
    return (ctypes.c_int(nf), ctypes.c_int(is_), ctypes.c_int(ts), ctypes.c_int(rr), ctypes.c_int(ra),
            ctypes.c_int(rd), ctypes.c_double(near), ctypes.c_double(far), ctypes.c_double(eps))


class RasterState(object):
    """Everything Rasterize keeps on `self` between forward and backward."""
    pass


def forward(faces, textures=None, image_size=256, near=0.1, far=100, eps=1e-4, background_color=(0, 0, 0),
            return_rgb=False, return_alpha=False, return_depth=False, unsafe=False, impl='oracle'):
    """Rasterize.forward_gpu (rasterize.py:464-510).  faces [bs,nf,3,3] f32 -> RasterState."""
    if not any((return_rgb, return_alpha, return_depth)):
        raise Exception  # rasterize.py:25-27
    faces = np.ascontiguousarray(faces, dtype=np.float32).copy()
    bs, nf = faces.shape[:2]
    is_ = int(image_size)
    st = RasterState()
    st.faces, st.bs, st.nf, st.image_size = faces, bs, nf, is_
    st.near, st.far, st.eps = float(near), float(far), float(eps)
    st.return_rgb, st.return_alpha, st.return_depth = bool(return_rgb), bool(return_alpha), bool(return_depth)
    st.impl = impl
    st.face_index_map = -np.ones((bs, is_, is_), np.int32)
    st.weight_map = np.zeros((bs, is_, is_, 3), np.float32)
    st.depth_map = np.zeros((bs, is_, is_), np.float32) + np.float32(far)
    ts = 0
    if return_rgb:
        st.textures = np.ascontiguousarray(textures, dtype=np.float32)
        ts = st.textures.shape[2]
        st.rgb_map = np.zeros((bs, is_, is_, 3), np.float32)
        st.sampling_index_map = np.zeros((bs, is_, is_, 8), np.int32)
        st.sampling_weight_map = np.zeros((bs, is_, is_, 8), np.float32)
    else:
        st.textures = None
        st.rgb_map = None
        st.sampling_index_map = None
        st.sampling_weight_map = None
    st.texture_size = ts
    st.alpha_map = np.zeros((bs, is_, is_), np.float32) if return_alpha else None
    # face_inv_map is only kept when depth is requested (rasterize.py:490-493)
    st.face_inv_map = np.zeros((bs, is_, is_, 3, 3), np.float32) if return_depth else np.zeros(1, np.float32)
    npx = bs * is_ * is_

    if impl == 'oracle':
        L = lib()
        if unsafe:
            L.orc_raster_unsafe(_fp(faces), bs, nf, is_, ctypes.c_double(near), ctypes.c_double(far),
                                int(return_depth), _ip(st.face_index_map), _fp(st.weight_map),
                                _fp(st.depth_map), _fp(st.face_inv_map))
        else:
            faces_inv = np.zeros_like(faces)
            L.orc_face_setup_safe(_fp(faces), bs * nf, is_, _fp(faces_inv))
            L.orc_raster_safe(_fp(faces), _fp(faces_inv), bs, nf, is_, ctypes.c_double(near),
                              ctypes.c_double(far), int(return_depth), _ip(st.face_index_map),
                              _fp(st.weight_map), _fp(st.depth_map), _fp(st.face_inv_map))
            st.faces_inv = faces_inv
        if return_rgb:
            L.orc_texture_sampling(_fp(faces), _fp(st.textures), _ip(st.face_index_map), _fp(st.weight_map),
                                   _fp(st.depth_map), bs, nf, is_, ts, ctypes.c_double(eps), _fp(st.rgb_map),
                                   _ip(st.sampling_index_map), _fp(st.sampling_weight_map))
    else:
        R = ref()
        P = _ref_params(nf, is_, ts, int(return_rgb), int(return_alpha), int(return_depth), near, far, eps)
        if unsafe:
            lock = np.zeros((bs, is_, is_), np.int32)
            R.ref_k1_unsafe_forward(ctypes.c_long(bs * nf), _fp(faces), _ip(st.face_index_map),
                                    _fp(st.weight_map), _fp(st.depth_map), _fp(st.face_inv_map), _ip(lock), *P)
        else:
            faces_inv = np.zeros_like(faces)
            R.ref_k2_face_setup(ctypes.c_long(bs * nf), _fp(faces), _fp(faces_inv), *P)
            R.ref_k3_safe_forward(ctypes.c_long(npx), _fp(faces), _fp(faces_inv), _ip(st.face_index_map),
                                  _fp(st.weight_map), _fp(st.depth_map), _fp(st.face_inv_map), *P)
            st.faces_inv = faces_inv
        if return_rgb:
            R.ref_k4_texture_sampling(ctypes.c_long(npx), _fp(faces), _fp(st.textures), _ip(st.face_index_map),
                                      _fp(st.weight_map), _fp(st.depth_map), _fp(st.rgb_map),
                                      _ip(st.sampling_index_map), _fp(st.sampling_weight_map), *P)

    # background + alpha (rasterize.py:437-462) -- host array ops in the reference
    if return_rgb:
        bg = np.asarray(background_color, np.float32)
        mask = (0 <= st.face_index_map).astype(np.float32)[:, :, :, None]
        if bg.ndim == 1:
            st.rgb_map = st.rgb_map * mask + (1 - mask) * bg[None, None, None, :]
        else:
            st.rgb_map = st.rgb_map * mask + (1 - mask) * bg[:, None, None, :]
        st.rgb_map = np.ascontiguousarray(st.rgb_map, np.float32)
    if return_alpha:
        st.alpha_map[0 <= st.face_index_map] = 1
    return st


def backward(st, grad_rgb=None, grad_alpha=None, grad_depth=None):
    """Rasterize.backward_gpu (rasterize.py:846-886) -> (grad_faces, grad_textures or None)."""
    bs, nf, is_, ts = st.bs, st.nf, st.image_size, st.texture_size
    npx = bs * is_ * is_
    grad_faces = np.zeros_like(st.faces)
    grad_textures = np.zeros_like(st.textures) if st.return_rgb else None
    one = np.zeros(1, np.float32)

    def prep(g, like, flag):
        if not flag:
            return one
        if g is None:
            return np.zeros_like(like)
        return np.ascontiguousarray(g, np.float32)

    g_rgb = prep(grad_rgb, st.rgb_map, st.return_rgb)
    g_alpha = prep(grad_alpha, st.alpha_map, st.return_alpha)
    g_depth = prep(grad_depth, st.depth_map, st.return_depth)
    rgb_map = st.rgb_map if st.return_rgb else one
    alpha_map = st.alpha_map if st.return_alpha else one

    if st.impl == 'oracle':
        L = lib()
        if st.return_rgb or st.return_alpha:
            L.orc_backward_pixel_map(_fp(st.faces), _ip(st.face_index_map), _fp(rgb_map), _fp(alpha_map),
                                     _fp(g_rgb), _fp(g_alpha), bs, nf, is_, ctypes.c_double(st.eps),
                                     int(st.return_rgb), int(st.return_alpha), _fp(grad_faces))
        if st.return_rgb:
            L.orc_backward_textures(_ip(st.face_index_map), _fp(st.sampling_weight_map),
                                    _ip(st.sampling_index_map), _fp(g_rgb), bs, nf, is_, ts, _fp(grad_textures))
        if st.return_depth:
            L.orc_backward_depth(_fp(st.faces), _fp(st.depth_map), _ip(st.face_index_map), _fp(st.face_inv_map),
                                 _fp(st.weight_map), _fp(g_depth), bs, nf, is_, _fp(grad_faces))
    else:
        R = ref()
        P = _ref_params(nf, is_, ts, int(st.return_rgb), int(st.return_alpha), int(st.return_depth),
                        st.near, st.far, st.eps)
        if st.return_rgb or st.return_alpha:
            R.ref_k5_backward_pixel_map(ctypes.c_long(bs * nf), _fp(st.faces), _ip(st.face_index_map), _fp(rgb_map),
                                        _fp(alpha_map), _fp(g_rgb), _fp(g_alpha), _fp(grad_faces), *P)
        if st.return_rgb:
            R.ref_k6_backward_textures(ctypes.c_long(npx), _ip(st.face_index_map), _fp(st.sampling_weight_map),
                                       _ip(st.sampling_index_map), _fp(g_rgb), _fp(grad_textures), *P)
        if st.return_depth:
            R.ref_k7_backward_depth(ctypes.c_long(npx), _fp(st.faces), _fp(st.depth_map), _ip(st.face_index_map),
                                    _fp(st.face_inv_map), _fp(st.weight_map), _fp(g_depth), _fp(grad_faces), *P)
    return grad_faces, grad_textures
