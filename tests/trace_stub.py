"""Test infrastructure: run the HOST side of the HIP path on CPU tensors and record what it would launch.

libsdn_hip.so is replaced by callbacks with the library's own ctypes signatures that return 0 and append
(entry point, arguments) to a list; torch.Tensor.is_cuda answers True and the current-stream queries return a dummy.  The
tensors hold garbage -- only the launch sequence is meaningful: which entry points, in which order, with which scalar
arguments and which buffers (data pointers) wired to which.  Used by tests/test_launch_trace.py to pin host logic that
otherwise only a GPU run exercises (stage wiring of a conv chain, the dual discriminator pass, cache invalidation per
optimizer), and to check host-side refactors for identical launch sequences before spending GPU time on them."""
import ctypes

import torch


def _decode(o, P, blob):
    """one sdn_op record -> (entry point name, argument tuple in the entry point's declaration order): the mapping of
    csrc/fast_program.hip's switch, restated"""
    from sdn_hip import program as pg
    i, f, l, b = list(o.i), list(o.f), list(o.l), [P(s) for s in o.buf]
    taps = None
    if o.taps >= 0 and o.code != pg.OP_CONV_GEMM_PHASES:
        n = {pg.OP_CONV_GEMM: i[13], pg.OP_CONV_TILE: i[14], pg.OP_CONV_HALO: i[7]}.get(o.code, i[8])
        taps = (tuple(blob[o.taps:o.taps + n]), tuple(blob[o.taps + n:o.taps + 2 * n]))
    c = o.code
    if c == pg.OP_CONV_GEMM_PHASES:
        # one record = the phase launches of a transposed conv / strided data gradient: restated as the sdn_conv_gemm calls the
        # phases would be one by one (what the trace tests count and wire), in record order
        calls, off = [], o.taps
        for q in range(i[9]):
            qh, qw, py, px, nt, kp = i[16 + 6 * q:22 + 6 * q]
            dy, dx = tuple(blob[off:off + nt]), tuple(blob[off + nt:off + 2 * nt])
            off += 2 * nt
            calls.append(('sdn_conv_gemm', (b[0], *i[0:4], b[1], *i[4:7], qh, qw, i[7], i[8], py, px, nt, dy, dx, i[10], i[11],
                                            b[2 + q], kp, i[12], b[6], i[13], b[7], i[14], i[15], None, 0, o.stream)))
        return calls
    if c == pg.OP_CONV_GEMM:
        return 'sdn_conv_gemm', (b[0], *i[0:4], b[1], *i[4:14], taps[0], taps[1], i[14], i[15], b[2], i[16], i[17], b[3],
                                 i[18], b[4], i[19], i[20], b[5], l[0], o.stream)
    if c == pg.OP_CONV_HEAD_MFMA:
        return 'sdn_conv_head_mfma', (b[0], *i[0:4], b[1], *i[4:8], b[2], *i[8:14], b[3], i[14], b[4], o.stream)
    if c == pg.OP_CONV_NARROW_FWD:
        return 'sdn_conv_narrow_fwd', (b[0], *i[0:4], b[1], *i[4:8], b[2], *i[8:14], b[3], i[14], o.stream)
    if c == pg.OP_IN_APPLY:
        return 'sdn_in_apply', (b[0], b[1], b[2], b[3], b[4], *i[0:4], f[0], i[4], i[5], f[1], b[5], b[6], b[7], l[0], i[6],
                                o.stream)
    if c == pg.OP_IN_BWD:
        return 'sdn_in_bwd', (b[0], b[1], b[2], b[3], *i[0:4], b[4], l[0], o.stream)
    if c == pg.OP_ACT_BWD:
        return 'sdn_act_bwd', (b[0], b[1], b[2], l[0], i[0], i[1], b[3], l[1], o.stream)
    if c == pg.OP_REFLECT_FOLD:
        return 'sdn_reflect_fold', (b[0], b[1], *i[0:6], o.stream)
    if c == pg.OP_CONV_WGRAD:
        return 'sdn_conv_wgrad', (b[0], b[1], b[2], *i[0:9], taps[0], taps[1], *i[9:14], b[3], l[0], o.stream)
    if c == pg.OP_CONV_WGRAD_NARROW:
        return 'sdn_conv_wgrad_narrow', (b[0], b[1], b[2], *i[0:9], taps[0], taps[1], *i[9:12], o.stream)
    if c == pg.OP_CONV_WGRAD_HEAD:
        return 'sdn_conv_wgrad_head_mfma', (b[0], b[1], b[2], *i[0:9], taps[0], taps[1], *i[9:12], o.stream)
    if c == pg.OP_PACK_WEIGHTS:
        return 'sdn_conv_pack_weights', (b[0], i[0], i[1], l[0], l[1], b[1], *i[2:6], b[2], o.stream)
    if c == pg.OP_UNPACK_GRAD:
        return 'sdn_conv_unpack_grad', (b[0], i[0], i[1], l[0], l[1], b[1], i[2], i[3], b[2], i[4], o.stream)
    if c == pg.OP_SPLIT_PLANES:
        return 'sdn_split_planes', (b[0], l[0], i[0], b[1], l[1], o.stream)
    if c == pg.OP_PACK_WEIGHTS_KMAJOR:
        return 'sdn_conv_pack_weights_kmajor', (b[0], i[0], i[1], l[0], l[1], b[1], *i[2:5], b[2], o.stream)
    if c == pg.OP_CONV_TILE:
        return 'sdn_conv_tile', (b[0], l[0], *i[0:4], b[1], b[2], l[1], *i[4:15], taps[0], taps[1], i[15], b[3], i[16], b[4],
                                 i[17], b[5], i[18], i[19], o.stream)
    if c == pg.OP_CONV_HALO:
        return 'sdn_conv_halo', (b[0], l[0], *i[0:4], b[1], *i[4:8], taps[0], taps[1], i[8], b[2], i[9], b[3], i[10], b[4], i[11],
                                 o.stream)
    if c == pg.OP_CONV_WGRAD_TILE:
        return 'sdn_conv_wgrad_tile', (b[0], l[0], b[1], l[1], b[2], *i[0:9], taps[0], taps[1], i[9], o.stream)
    return pg.OP_NAMES[c], (b[0], b[1], b[2], l[0], o.stream)


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
                 'sdn_conv_gemm_workspace_bytes', 'sdn_nms_workspace_bytes', 'sdn_conv_halo_blocks',
                 'sdn_perspective_transform_scratch', 'sdn_timing_declare_work', 'sdn_conv_head_steps')
    from sdn_hip import program as pg
    programs = {}

    def program_create(ops, n_ops, taps, tap_bytes, n_slots, out):
        recs = (pg.SdnOp * n_ops).from_address(ops)
        copy = (pg.SdnOp * n_ops)()
        ctypes.memmove(copy, recs, ctypes.sizeof(copy))
        blob = list((ctypes.c_int8 * tap_bytes).from_address(taps)) if tap_bytes else []
        h = len(programs) + 1
        programs[h] = (copy, blob, n_slots)
        out[0] = h
        return 0

    def program_run(h, slots, n_slots, main, side, op_ms, failed):
        recs, blob, n = programs[h]
        assert n == n_slots
        table = [slots[k] for k in range(n_slots)]
        for o in recs:
            got = _decode(o, lambda s: None if s < 0 else table[s], blob)
            trace.calls.extend(got if isinstance(got, list) else [got])
        return 0
    special = {'sdn_program_create': program_create, 'sdn_program_run': program_run, 'sdn_program_destroy': lambda h: 0}
    for name in sdn_hip.exported_symbols():
        f = getattr(real, name)
        if name in host_only:
            setattr(stub, name, f)
            continue
        if name in special:
            proto = ctypes.CFUNCTYPE(f.restype, *(f.argtypes or []))
            c = proto(special[name])
            keep.append(c)
            setattr(stub, name, c)
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
    for mod in (sdn_hip, conv, ops, bnnet, pg):
        if hasattr(mod, 'lib'):
            monkeypatch.setattr(mod, 'lib', lambda: stub)
        if hasattr(mod, 'stream'):
            monkeypatch.setattr(mod, 'stream', lambda: None)
    monkeypatch.setattr(torch.Tensor, 'is_cuda', property(lambda self: True))
    monkeypatch.setattr(torch.cuda, 'current_stream', lambda *a, **k: _Stream())
    monkeypatch.setenv('SDN_WGRAD_STREAM', '0')
    monkeypatch.setenv('SDN_D_STREAMS', '0')
    monkeypatch.setenv('SDN_UPDATE_STREAM', '0')
    return trace
