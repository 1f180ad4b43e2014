"""Test infrastructure: run the HOST side of the HIP path on CPU tensors and record what it would launch.

libsdn_hip.so is replaced by callbacks with the library's own ctypes signatures that return 0 and append
(entry point, arguments) to a list; torch.Tensor.is_cuda answers True and the current-stream queries return a dummy.  The
tensors hold garbage -- only the launch sequence is meaningful: which entry points, in which order, with which scalar
arguments and which buffers (data pointers) wired to which.  Used by tests/test_launch_trace.py to pin host logic that
otherwise only a GPU run exercises (stage wiring of a conv chain, the dual discriminator pass, cache invalidation per
optimizer), and to check host-side refactors for identical launch sequences before spending GPU time on them."""
import ctypes

import torch


class Trace:
    def __init__(self):
        self.calls = []

    def names(self):
        return [c[0] for c in self.calls]

    def count(self, name):
        return sum(1 for c in self.calls if c[0] == name)

    def of(self, name):
        return [c[1] for c in self.calls if c[0] == name]

    def clear(self):
        del self.calls[:]


class _Stream:
    cuda_stream = 0

    def wait_stream(self, other):
        pass


def install(monkeypatch):
    """-> Trace.  Everything is undone by the monkeypatch fixture."""
    import sdn_hip
    from sdn_hip import bnnet, conv, ops
    real = sdn_hip.lib()
    trace = Trace()

    class Stub:
        pass
    stub = Stub()
    keep = []
    host_only = ('sdn_raster_workspace_bytes', 'sdn_raster_bwd_workspace_bytes', 'sdn_last_error', 'sdn_version',
                 'sdn_conv_gemm_workspace_bytes', 'sdn_nms_workspace_bytes')
    for name in sdn_hip.exported_symbols():
        f = getattr(real, name)
        if name in host_only:
            setattr(stub, name, f)
            continue

        def make(n):
            def cb(*a):
                trace.calls.append((n, a))
                return 0
            return cb
        proto = ctypes.CFUNCTYPE(f.restype, *(f.argtypes or []))
        c = proto(make(name))
        keep.append(c)
        setattr(stub, name, c)
    stub._keep = keep
    for mod in (sdn_hip, conv, ops, bnnet):
        if hasattr(mod, 'lib'):
            monkeypatch.setattr(mod, 'lib', lambda: stub)
        if hasattr(mod, 'stream'):
            monkeypatch.setattr(mod, 'stream', lambda: None)
    monkeypatch.setattr(torch.Tensor, 'is_cuda', property(lambda self: True))
    monkeypatch.setattr(torch.cuda, 'current_stream', lambda *a, **k: _Stream())
    monkeypatch.setenv('SDN_WGRAD_STREAM', '0')
    monkeypatch.setenv('SDN_D_STREAMS', '0')
    return trace
