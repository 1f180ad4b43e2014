"""ctypes binding of libsdn_hip.so (include/sdn_hip.h) for torch tensors on an MI355X.

This is plumbing only: it checks dtype / device / contiguity, hands `tensor.data_ptr()` and the
current HIP stream to the C ABI and turns error codes into exceptions.  There is NO fallback: if
the shared library is missing or a tensor is not on the GPU the call raises (the reference's
CPU path raises NotImplementedError too, neural_renderer/rasterize.py:890-894).
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.normpath(os.path.join(_HERE, '..', 'lib', 'libsdn_hip.so'))

# flags (include/sdn_hip.h)
RGB, ALPHA, DEPTH, AA, FACE_COLOR, SAVE_MAPS, ACCUMULATE, SERIAL_EDGES, STREAM_FACES, COUNT_WORK = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512
K1_COVERAGE = 4096   # SDN_K1_COVERAGE: the reference's default ("unsafe") forward kernel's coverage rule, deterministic ties

ABI_VERSION = 8   # include/sdn_hip.h: SDN_ABI_VERSION this binding was written against (buffer sizes, argument lists)

_lib = None
_lock = threading.Lock()
_vp = ctypes.c_void_p
_ci = ctypes.c_int
_cl = ctypes.c_long
_cd = ctypes.c_double
_sz = ctypes.c_size_t


class SdnHipError(RuntimeError):
    pass


def _declare(L):
    L.sdn_last_error.restype = ctypes.c_char_p
    L.sdn_version.restype = _ci
    sig = {
        'sdn_project_vertices': [_vp, _ci, _ci, _ci, _vp, _vp, _vp, _vp, _ci, _vp, _vp],
        'sdn_project_vertices_bwd': [_vp, _ci, _ci, _ci, _vp, _vp, _vp, _vp, _ci, _vp, _vp, _vp],
        'sdn_gather_faces': [_vp, _vp, _ci, _ci, _ci, _cl, _ci, _vp, _vp],
        'sdn_gather_faces_bwd': [_vp, _vp, _ci, _ci, _ci, _cl, _ci, _vp, _vp],
        'sdn_face_normals': [_vp, _cl, _vp, _vp],
        'sdn_face_normals_bwd': [_vp, _vp, _cl, _vp, _vp],
        'sdn_raster_workspace_bytes': [_ci, _ci, _ci, ctypes.POINTER(_sz)],
        'sdn_rasterize_fwd': [_vp, _vp, _ci, _ci, _ci, _ci, _cd, _cd, _cd, _vp, _ci, _ci, _vp, _vp, _vp, _vp, _vp,
                              _vp, _vp, _vp, _vp, _sz, _vp],
        'sdn_raster_bwd_workspace_bytes': [_ci, _ci, _ci, ctypes.POINTER(_sz)],
        'sdn_rasterize_bwd': [_vp, _vp, _ci, _ci, _ci, _ci, _cd, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                              _vp, _vp, _sz, _vp],
    }
    sig['sdn_raster_work_counters'] = [_vp, _ci, _ci, _ci, ctypes.POINTER(ctypes.c_ulonglong), _vp]
    sig['sdn_ffd_decode'] = [_vp, _vp, _vp, _ci, _ci, _ci, _vp, _vp]
    sig['sdn_ffd_decode_bwd'] = [_vp, _vp, _vp, _ci, _ci, _ci, _vp, _vp]
    _i8p = ctypes.POINTER(ctypes.c_int8)
    _cf = ctypes.c_float
    sig['sdn_conv_gemm_workspace_bytes'] = [_ci, _ci, _ci, _ci, ctypes.POINTER(_sz)]
    sig['sdn_conv_gemm'] = [_vp, _ci, _ci, _ci, _ci, _vp, _ci, _ci, _ci, _ci, _ci, _ci, _ci, _ci, _ci, _ci, _i8p, _i8p,
                            _ci, _ci, _vp, _ci, _ci, _vp, _ci, _vp, _ci, _ci, _vp, _sz, _vp]
    _i32p = ctypes.POINTER(ctypes.c_int32)
    sig['sdn_conv_gemm_phases'] = [_vp, _ci, _ci, _ci, _ci, _vp, _ci, _ci, _ci, _ci, _ci, _ci, _i32p, _i32p, _i32p, _i32p, _i32p, _i8p,
                                   _ci, _ci, ctypes.POINTER(_vp), _i32p, _ci, _vp, _ci, _vp, _ci, _ci, _vp]
    sig['sdn_conv_wgrad'] = [_vp, _vp, _vp, _ci, _ci, _ci, _ci, _ci, _ci, _ci, _ci, _ci, _i8p, _i8p, _ci, _ci, _ci, _ci,
                             _ci, _vp, _sz, _vp]
    sig['sdn_conv_wgrad_narrow'] = [_vp, _vp, _vp, _ci, _ci, _ci, _ci, _ci, _ci, _ci, _ci, _ci, _i8p, _i8p, _ci, _ci, _ci,
                                    _vp]
    sig['sdn_conv_wgrad_head_mfma'] = [_vp, _vp, _vp, _ci, _ci, _ci, _ci, _ci, _ci, _ci, _ci, _ci, _i8p, _i8p, _ci, _ci, _ci,
                                    _vp]
    sig['sdn_conv_narrow_fwd'] = [_vp, _ci, _ci, _ci, _ci, _vp, _ci, _ci, _ci, _ci, _vp, _ci, _ci, _ci, _ci, _ci, _ci,
                                  _vp, _ci, _vp]
    sig['sdn_conv_head_steps'] = [_ci, _ci, _ci, ctypes.POINTER(_ci)]
    sig['sdn_conv_head_mfma'] = [_vp, _ci, _ci, _ci, _ci, _vp, _ci, _ci, _ci, _ci, _vp, _ci, _ci, _ci, _ci, _ci, _ci, _vp, _ci, _vp, _vp]
    sig['sdn_in_apply'] = [_vp, _vp, _vp, _vp, _vp, _ci, _ci, _ci, _ci, _cf, _ci, _ci, _cf, _vp, _vp, _vp, _cl, _ci, _vp]
    sig['sdn_in_bwd'] = [_vp, _vp, _vp, _vp, _ci, _ci, _ci, _ci, _vp, _cl, _vp]
    sig['sdn_act_bwd'] = [_vp, _vp, _vp, _cl, _ci, _ci, _vp, _cl, _vp]
    sig['sdn_reflect_fold'] = [_vp, _vp, _ci, _ci, _ci, _ci, _ci, _ci, _vp]
    sig['sdn_conv_pack_weights'] = [_vp, _ci, _ci, _cl, _cl, _vp, _ci, _ci, _ci, _ci, _vp, _vp]
    sig['sdn_conv_unpack_grad'] = [_vp, _ci, _ci, _cl, _cl, _vp, _ci, _ci, _vp, _ci, _vp]
    sig['sdn_split_planes'] = [_vp, _cl, _ci, _vp, _cl, _vp]
    sig['sdn_assemble_nhwc'] = [_vp, _vp, _ci, _ci, _ci, _ci, _ci, _vp, _vp]
    sig['sdn_conv_pack_weights_kmajor'] = [_vp, _ci, _ci, _cl, _cl, _vp, _ci, _ci, _ci, _vp, _vp]
    sig['sdn_conv_tile'] = [_vp, _cl, _ci, _ci, _ci, _ci, _vp, _vp, _cl, _ci, _ci, _ci, _ci, _ci, _ci, _ci, _ci, _ci, _ci, _ci,
                            _i8p, _i8p, _ci, _vp, _ci, _vp, _ci, _vp, _ci, _ci, _vp]
    sig['sdn_conv_halo_blocks'] = [_ci, _ci, _ci, _ci, _ci, _ci, ctypes.POINTER(_ci), ctypes.POINTER(_ci), ctypes.POINTER(_cl)]
    sig['sdn_conv_halo'] = [_vp, _cl, _ci, _ci, _ci, _ci, _vp, _ci, _ci, _ci, _ci, _i8p, _i8p, _ci, _vp, _ci, _vp, _ci, _vp, _ci, _vp]
    sig['sdn_conv_wgrad_tile'] = [_vp, _cl, _vp, _cl, _vp, _ci, _ci, _ci, _ci, _ci, _ci, _ci, _ci, _ci, _i8p, _i8p, _ci, _vp]
    sig['sdn_segment_mean'] = [_vp, _vp, _ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp]
    sig['sdn_l1_loss_fwd'] = [_vp, _vp, _cl, _vp, _vp, _vp]
    sig['sdn_l1_loss_bwd'] = [_vp, _vp, _cl, _vp, _vp, _vp, _vp]
    sig['sdn_silhouette_loss_fwd'] = [_vp, _vp, _vp, _cl, _vp, _cl, _vp, _vp, _vp]
    sig['sdn_silhouette_loss_bwd'] = [_vp, _vp, _vp, _cl, _vp, _cl, _vp, _vp, _vp, _vp, _vp]
    sig['sdn_pose_params'] = [_vp, _vp, _ci, _vp, _vp, _vp]
    sig['sdn_pose_params_bwd'] = [_vp, _vp, _vp, _vp, _ci, _vp, _vp, _vp]
    sig['sdn_pose_algebra'] = [_vp] * 7 + [_ci, _ci, ctypes.c_float, ctypes.c_float] + [_vp] * 9 + [_vp]
    sig['sdn_pose_algebra_bwd'] = [_vp] * 8 + [_ci, _ci] + [_vp] * 8 + [_vp] * 4 + [_vp]
    sig['sdn_composite_frame'] = [_vp, _vp, _vp, _vp, _ci, _ci, _vp, _ci, _vp, _vp, _vp, _ci, _ci, _vp, _vp, _vp, _vp]
    sig['sdn_perspective_transform_scratch'] = [_ci, _ci, ctypes.POINTER(_sz), ctypes.POINTER(_sz)]
    sig['sdn_perspective_transform'] = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _ci, _ci, _vp, _vp, _vp, _vp]
    sig['sdn_perspective_transform_bwd'] = [_vp, _vp, _vp, _vp, _vp, _vp, _ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                            _vp, _vp, _vp, _vp]
    _i8pp = _vp
    sig['sdn_bn_forward'] = [_vp, _cl, _ci, _vp, _vp, _vp, _vp, _cf, _cf, _ci, _vp, _ci, _vp, _vp, _vp, _vp, _vp]
    sig['sdn_bn_backward'] = [_vp, _vp, _vp, _vp, _vp, _cl, _ci, _ci, _ci, _vp, _vp, _vp, _vp]
    sig['sdn_maxpool3x3s2_fwd'] = [_vp, _ci, _ci, _ci, _ci, _vp, _i8pp, _vp]
    sig['sdn_maxpool3x3s2_bwd'] = [_vp, _i8pp, _ci, _ci, _ci, _ci, _vp, _vp]
    sig['sdn_avgpool_global'] = [_vp, _ci, _ci, _ci, _vp, _ci, _vp]
    sig['sdn_nms_workspace_bytes'] = [_ci, ctypes.POINTER(_sz)]
    sig['sdn_nms'] = [_vp, _vp, _ci, _cf, _ci, _vp, _vp, _vp, _sz, _vp]
    sig['sdn_crop_and_resize_fwd'] = [_vp, _ci, _ci, _ci, _ci, _vp, _vp, _ci, _ci, _ci, _cf, _vp, _vp]
    sig['sdn_crop_and_resize_bwd'] = [_vp, _vp, _vp, _ci, _ci, _ci, _vp, _ci, _ci, _ci, _ci, _vp]
    sig['sdn_timing_enable'] = [_ci]
    sig['sdn_timing_declare_work'] = [_cd]
    sig['sdn_timing_read'] = [ctypes.POINTER(_cd), ctypes.POINTER(_cl)]
    sig['sdn_timing_read_slot'] = [_ci, ctypes.POINTER(_cd), ctypes.POINTER(_cl), ctypes.POINTER(_cd)]
    sig['sdn_ffd_coefficients'] = [_vp, _vp, _vp, _ci, _ci, _ci, _vp, _vp]
    sig['sdn_raster_phase_clocks'] = [_vp, _ci, _ci, _ci, ctypes.POINTER(ctypes.c_ulonglong), _vp]
    sig['sdn_render_maps_bytes'] = [_ci, _ci, _ci, _ci, _ci, _ci, ctypes.POINTER(_sz), ctypes.POINTER(_sz), ctypes.POINTER(_sz)]
    sig['sdn_render_maps_fwd'] = [_vp, _ci, _ci, _vp, _ci, _cl, _ci, _ci, _vp, _vp, _vp, _vp, _ci, _ci, _ci, _cd, _cd, _cd, _vp,
                                  _vp, _vp, _vp, _vp, _sz, _vp, _sz, _vp]
    sig['sdn_render_maps_bwd'] = [_vp, _ci, _ci, _vp, _ci, _cl, _ci, _ci, _vp, _vp, _vp, _vp, _ci, _ci, _ci, _cd, _cd, _vp, _vp, _vp,
                                  _vp, _vp, _vp, _sz, _vp, _sz, _vp]
    _lp = ctypes.POINTER(_cl)
    sig['sdn_avgpool3x3s2_fwd'] = [_vp, _ci, _ci, _ci, _ci, _lp, _vp, _lp, _ci, _vp]
    sig['sdn_avgpool3x3s2_bwd'] = [_vp, _ci, _ci, _ci, _ci, _lp, _vp, _lp, _ci, _vp]
    sig['sdn_program_create'] = [_vp, _ci, _vp, _sz, _ci, ctypes.POINTER(_vp)]
    sig['sdn_program_run'] = [_vp, ctypes.POINTER(_vp), _ci, _vp, _vp, ctypes.POINTER(_cf), ctypes.POINTER(_ci)]
    sig['sdn_program_destroy'] = [_vp]
    for name, argtypes in sig.items():
        fn = getattr(L, name)
        fn.argtypes = argtypes
        fn.restype = _ci
    # optional families (present once the corresponding .hip files are linked in)
    return sig


def lib():
    """Load the HIP library once; fail loudly when it is absent."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise SdnHipError(
                        'libsdn_hip.so not found at %s -- run `python __graft_entry__.py build` (hipcc, gfx950); '
                        'there is no CPU or PyTorch fallback for this path' % LIB_PATH)
                L = ctypes.CDLL(LIB_PATH)
                L.sdn_version.restype = _ci
                have = L.sdn_version()
                if have != ABI_VERSION:
                    # signatures are plain C: a stale library would take buffers of the wrong size without a word
                    raise SdnHipError('%s reports ABI version %d, this binding needs %d -- rebuild it '
                                      '(`python __graft_entry__.py build`)' % (LIB_PATH, have, ABI_VERSION))
                _declare(L)
                _lib = L
    return _lib


def exported_symbols():
    """Names declared in include/sdn_hip.h that this binding expects."""
    return ['sdn_last_error', 'sdn_version', 'sdn_project_vertices', 'sdn_project_vertices_bwd', 'sdn_gather_faces',
            'sdn_gather_faces_bwd', 'sdn_face_normals', 'sdn_face_normals_bwd', 'sdn_raster_workspace_bytes',
            'sdn_rasterize_fwd', 'sdn_raster_work_counters', 'sdn_raster_bwd_workspace_bytes', 'sdn_rasterize_bwd', 'sdn_ffd_decode', 'sdn_ffd_decode_bwd', 'sdn_ffd_coefficients',
            'sdn_timing_enable', 'sdn_timing_declare_work', 'sdn_timing_read', 'sdn_timing_read_slot', 'sdn_conv_gemm', 'sdn_conv_gemm_phases', 'sdn_conv_gemm_workspace_bytes', 'sdn_conv_wgrad', 'sdn_conv_wgrad_narrow', 'sdn_conv_wgrad_head_mfma', 'sdn_conv_narrow_fwd', 'sdn_conv_head_steps', 'sdn_conv_head_mfma', 'sdn_in_apply', 'sdn_in_bwd',
            'sdn_act_bwd', 'sdn_reflect_fold', 'sdn_conv_pack_weights', 'sdn_conv_unpack_grad', 'sdn_split_planes', 'sdn_conv_pack_weights_kmajor',
            'sdn_conv_tile', 'sdn_conv_wgrad_tile', 'sdn_conv_halo', 'sdn_conv_halo_blocks', 'sdn_segment_mean', 'sdn_l1_loss_fwd', 'sdn_l1_loss_bwd', 'sdn_silhouette_loss_fwd', 'sdn_silhouette_loss_bwd', 'sdn_assemble_nhwc', 'sdn_pose_params', 'sdn_pose_algebra', 'sdn_pose_algebra_bwd',
            'sdn_pose_params_bwd', 'sdn_composite_frame',
            'sdn_perspective_transform_scratch', 'sdn_perspective_transform', 'sdn_perspective_transform_bwd', 'sdn_bn_forward', 'sdn_bn_backward',
            'sdn_maxpool3x3s2_fwd', 'sdn_maxpool3x3s2_bwd', 'sdn_avgpool_global', 'sdn_nms_workspace_bytes', 'sdn_nms',
            'sdn_crop_and_resize_fwd', 'sdn_crop_and_resize_bwd', 'sdn_avgpool3x3s2_fwd', 'sdn_avgpool3x3s2_bwd', 'sdn_render_maps_bytes', 'sdn_render_maps_fwd', 'sdn_raster_phase_clocks',
            'sdn_render_maps_bwd', 'sdn_program_create', 'sdn_program_run',
            'sdn_program_destroy']


def check(rc):
    if rc != 0:
        raise SdnHipError('libsdn_hip error %d: %s' % (rc, lib().sdn_last_error().decode()))


def ptr(t):
    # (a plain int: ctypes converts it for the c_void_p parameters declared in _declare(); building a c_void_p object per
    # argument was a measurable share of the ~65 pointer arguments of a frame step)
    return None if t is None else t.data_ptr()


def stream():
    return _vp(torch.cuda.current_stream().cuda_stream)


def want(t, dtype, name):
    """Validate a tensor argument the way chainer's type_check did (rasterize.py:66-90), plus device."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError('%s must be a torch.Tensor, got %r' % (name, type(t)))
    if not t.is_cuda:
        # the reference: Rasterize.forward_cpu raises NotImplementedError (rasterize.py:890-891)
        raise NotImplementedError('%s is on %s; this renderer only runs on the GPU' % (name, t.device))
    if t.dtype != dtype:
        raise TypeError('%s must be %s, got %s' % (name, dtype, t.dtype))
    return t.contiguous()


_const_cache = {}


def const_f32(values, device):
    """Device copy of a small host constant (camera vectors, background colours, per-object tan(angle)), cached by value:
    the reference re-uploads these on every call (derender3d/models/renderer.py:243-248); a cached copy also keeps the
    render path free of host-to-device copies, which a HIP-graph capture of the step requires."""
    import numpy as np
    a = np.ascontiguousarray(np.asarray(values, dtype=np.float32))
    key = (a.tobytes(), a.shape, str(device))
    t = _const_cache.get(key)
    if t is None:
        if len(_const_cache) > 4096:
            _const_cache.clear()
        t = _const_cache[key] = torch.tensor(a, device=device)
    return t


def raster_workspace(bs, nf, S, device):
    n = _sz(0)
    check(lib().sdn_raster_workspace_bytes(bs, nf, S, ctypes.byref(n)))
    return torch.empty(n.value, dtype=torch.uint8, device=device)


def perspective_transform_scratch(n, V):
    """(bytes of `key`, bytes of `acc`) of sdn_perspective_transform / _bwd, asked of the library."""
    kb, ab = _sz(0), _sz(0)
    check(lib().sdn_perspective_transform_scratch(n, V, ctypes.byref(kb), ctypes.byref(ab)))
    return kb.value, ab.value


def raster_bwd_workspace(bs, nf, S, device):
    n = _sz(0)
    check(lib().sdn_raster_bwd_workspace_bytes(bs, nf, S, ctypes.byref(n)))
    return torch.empty(n.value, dtype=torch.uint8, device=device)


def timing_enable(on=True):
    check(lib().sdn_timing_enable(int(bool(on))))


def timing_read():
    """(total k_raster_tiles milliseconds, launches) since the previous read."""
    ms, n = _cd(0), _cl(0)
    check(lib().sdn_timing_read(ctypes.byref(ms), ctypes.byref(n)))
    return ms.value, n.value


SLOT_RASTER_TILES, SLOT_EDGE_SCAN, SLOT_CONV_GEMM, SLOT_CONV_WGRAD, SLOT_CONV_NARROW, SLOT_RASTER_TILES_K1 = 0, 1, 2, 3, 4, 5


def timing_read_slot(slot):
    """(total milliseconds, launches, declared algorithmic work) of one timed kernel family since the previous read."""
    ms, n, w = _cd(0), _cl(0), _cd(0)
    check(lib().sdn_timing_read_slot(slot, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(w)))
    return ms.value, n.value, w.value
