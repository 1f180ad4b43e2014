"""The C-ABI library loads without a GPU and exports every symbol include/sdn_hip.h declares."""
import ctypes
import os
import re

import sdn_hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, 'include', 'sdn_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(sdn_[a-z0-9_]+)\s*\(', src)))


def test_library_exists_and_loads():
    assert os.path.exists(sdn_hip.LIB_PATH), 'run `python __graft_entry__.py build` first'
    L = sdn_hip.lib()
    assert L.sdn_version() == sdn_hip.ABI_VERSION == header_abi_version()
    assert isinstance(L.sdn_last_error(), bytes)


def header_abi_version():
    src = open(os.path.join(ROOT, 'include', 'sdn_hip.h')).read()
    return int(re.search(r'#define\s+SDN_ABI_VERSION\s+(\d+)', src).group(1))


def test_a_library_of_another_abi_revision_is_refused(monkeypatch):
    """buffer sizes changed behind unchanged signatures (sdn_perspective_transform's key / acc); a stale
    lib/libsdn_hip.so must raise at load time instead of writing out of bounds."""
    import pytest
    monkeypatch.setattr(sdn_hip, '_lib', None)
    monkeypatch.setattr(sdn_hip, 'ABI_VERSION', sdn_hip.ABI_VERSION + 1)
    with pytest.raises(sdn_hip.SdnHipError, match='ABI version'):
        sdn_hip.lib()


def test_perspective_transform_scratch_sizes_come_from_the_library():
    kb, ab = sdn_hip.perspective_transform_scratch(16, 30000)
    assert kb == 16 * (1 + (30000 + 255) // 256) * 8 and ab == 16 * 36 * 4
    n = ctypes.c_size_t(0)
    assert sdn_hip.lib().sdn_perspective_transform_scratch(0, 5, ctypes.byref(n), None) == -1


def test_every_header_symbol_is_exported():
    L = ctypes.CDLL(sdn_hip.LIB_PATH)
    names = header_functions()
    assert len(names) >= 11
    for n in names:
        assert hasattr(L, n), 'libsdn_hip.so does not export %s' % n


def test_binding_covers_the_header():
    assert set(sdn_hip.exported_symbols()) == set(header_functions())


def test_argument_validation_without_gpu():
    # error paths return codes and set the thread-local message; no kernel is launched
    L = sdn_hip.lib()
    n = ctypes.c_size_t(0)
    assert L.sdn_raster_workspace_bytes(0, 10, 64, ctypes.byref(n)) == -1
    assert b'bad sizes' in L.sdn_last_error()
    assert L.sdn_raster_workspace_bytes(2, 1000, 768, ctypes.byref(n)) == 0
    assert n.value >= 2 * 1000 * 12
    rc = L.sdn_rasterize_fwd(None, None, 0, 1, 10, 64, 0.1, 100.0, 1e-4, None, 0, 0, None, None, None, None, None,
                             None, None, None, None, 0, None)
    assert rc == -1 and b'nothing to draw' in L.sdn_last_error()
    # the vectorised weight pack / gradient unpack take 8 / 4 columns of one (row, tap) per thread -- a padded channel
    # count that is not a multiple of 8 / 4 (or a misaligned dw) is refused before anything is launched (fake non-null pointers)
    fake = ctypes.c_void_p(4096)
    assert L.sdn_conv_pack_weights(fake, 32, 6, 54, 9, fake, 9, 6, 64, 32, fake, None) == -1
    assert b'multiple of 8' in L.sdn_last_error()
    assert L.sdn_conv_unpack_grad(fake, 32, 6, 54, 9, fake, 9, 6, fake, 0, None) == -1
    assert b'multiple of 4' in L.sdn_last_error()
    assert L.sdn_conv_unpack_grad(ctypes.c_void_p(4100), 32, 8, 72, 9, fake, 9, 8, fake, 0, None) == -1
    assert b'16-byte aligned' in L.sdn_last_error()


def test_cpu_tensors_raise_like_the_reference():
    import pytest
    import torch
    import neural_renderer as nr
    faces = torch.zeros(1, 4, 3, 3)
    with pytest.raises(NotImplementedError):  # rasterize.py:890-894
        nr.rasterize_silhouettes(faces, 16)
    with pytest.raises(Exception):  # rasterize.py:25-27: nothing to draw
        nr.Rasterize(16, 0.1, 100, 1e-4, (0, 0, 0))
