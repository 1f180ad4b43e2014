"""Launch lists: host side of `sdn_program_*` (include/sdn_hip.h, csrc/fast_program.hip).

A pass of a conv chain is planned ONCE per (chain, input shape) into an array of `sdn_op` records -- which launcher, its
scalar arguments, and for every pointer argument a slot index -- and replayed with ONE ctypes call per pass.  At replay
time the host only fills the slot table with device pointers:

    static slots   persistent tensors (packed weights, biases, running statistics, tap-index arrays): data_ptr() per run
    arena slots    base pointer of a per-run allocation + an offset fixed at planning time (activations, statistics,
                   gradients, scratch); zero-initialised pieces are laid out first so that one memset covers them
    external slots pointers the caller names at run time (the chain input, incoming gradients)

The records name the library's own launchers one to one (SDN_OP_CONV_GEMM -> sdn_conv_gemm, ...): a program is the launch
sequence the per-launch entry points would be called with, stored.  There is no CPU path: running a program needs the GPU.
"""
import ctypes
import os

import numpy as np
import torch

from . import check, lib

(OP_CONV_GEMM, OP_CONV_NARROW_FWD, OP_IN_APPLY, OP_IN_BWD, OP_ACT_BWD, OP_REFLECT_FOLD, OP_CONV_WGRAD, OP_CONV_WGRAD_NARROW,
 OP_PACK_WEIGHTS, OP_UNPACK_GRAD, OP_MEMSET, OP_COPY, OP_ADD, OP_COLSUM, OP_FORK, OP_JOIN, OP_SPLIT_PLANES,
 OP_PACK_WEIGHTS_KMAJOR, OP_CONV_TILE, OP_CONV_HALO, OP_CONV_WGRAD_TILE, OP_CONV_GEMM_PHASES, OP_CONV_HEAD_MFMA,
 OP_CONV_WGRAD_HEAD) = range(1, 25)

OP_NAMES = {OP_CONV_GEMM: 'sdn_conv_gemm', OP_CONV_NARROW_FWD: 'sdn_conv_narrow_fwd', OP_IN_APPLY: 'sdn_in_apply',
            OP_IN_BWD: 'sdn_in_bwd', OP_ACT_BWD: 'sdn_act_bwd', OP_REFLECT_FOLD: 'sdn_reflect_fold',
            OP_CONV_WGRAD: 'sdn_conv_wgrad', OP_CONV_WGRAD_NARROW: 'sdn_conv_wgrad_narrow',
            OP_PACK_WEIGHTS: 'sdn_conv_pack_weights', OP_UNPACK_GRAD: 'sdn_conv_unpack_grad', OP_MEMSET: 'memset',
            OP_COPY: 'copy', OP_ADD: 'add', OP_COLSUM: 'colsum', OP_FORK: 'fork', OP_JOIN: 'join',
            OP_SPLIT_PLANES: 'sdn_split_planes', OP_PACK_WEIGHTS_KMAJOR: 'sdn_conv_pack_weights_kmajor',
            OP_CONV_TILE: 'sdn_conv_tile', OP_CONV_HALO: 'sdn_conv_halo', OP_CONV_WGRAD_TILE: 'sdn_conv_wgrad_tile',
            OP_CONV_GEMM_PHASES: 'sdn_conv_gemm_phases', OP_CONV_HEAD_MFMA: 'sdn_conv_head_mfma',
            OP_CONV_WGRAD_HEAD: 'sdn_conv_wgrad_head_mfma'}


_TIMED_CODES = (OP_CONV_GEMM, OP_CONV_NARROW_FWD, OP_CONV_WGRAD, OP_CONV_WGRAD_NARROW, OP_CONV_TILE, OP_CONV_HALO,
                OP_CONV_WGRAD_TILE, OP_CONV_GEMM_PHASES, OP_CONV_HEAD_MFMA, OP_CONV_WGRAD_HEAD)
N_INTS = 40   # sdn_op.i[]


class SdnOp(ctypes.Structure):
    """struct sdn_op of include/sdn_hip.h"""
    _fields_ = [('code', ctypes.c_int32), ('stream', ctypes.c_int32), ('buf', ctypes.c_int32 * 8),
                ('i', ctypes.c_int32 * N_INTS), ('f', ctypes.c_float * 4), ('l', ctypes.c_int64 * 2),
                ('taps', ctypes.c_int32), ('reserved', ctypes.c_int32)]


ALIGN = 256


def _round(n):
    return (int(n) + ALIGN - 1) // ALIGN * ALIGN


class Builder:
    """Collects slots and records; finish() lays the arenas out and creates the C-side program."""

    def __init__(self):
        self.ops = []            # (code, stream, bufs, ints, floats, longs, tap offset, description, flops)
        self.taps = bytearray()
        self._tap_cache = {}
        self.kinds = []          # per slot: ('static', tensor or callable) | ('arena', name, index) | ('ext', name)
        self._static_ids = {}
        self._ext = {}
        self.arenas = {}         # name -> list of [nbytes, zero, offset]

    # ---- slots
    def static(self, t):
        """a persistent tensor, or a zero-argument callable returning the tensor (parameters are looked up per run)"""
        key = id(t)
        s = self._static_ids.get(key)
        if s is None:
            s = self._static_ids[key] = len(self.kinds)
            self.kinds.append(('static', t))
        return s

    def ext(self, name):
        s = self._ext.get(name)
        if s is None:
            s = self._ext[name] = len(self.kinds)
            self.kinds.append(('ext', name))
        return s

    def alloc(self, arena, nbytes, zero=False):
        pieces = self.arenas.setdefault(arena, [])
        pieces.append([_round(max(int(nbytes), 1)), bool(zero), None])
        self.kinds.append(('arena', arena, len(pieces) - 1))
        return len(self.kinds) - 1

    # ---- records
    def tap_pair(self, taps):
        key = tuple(taps)
        off = self._tap_cache.get(key)
        if off is None:
            off = self._tap_cache[key] = len(self.taps)
            self.taps += np.asarray([t[0] for t in taps] + [t[1] for t in taps], dtype=np.int8).tobytes()
        return off

    def tap_lists(self, phases):
        """SDN_OP_CONV_GEMM_PHASES: per phase dy[ntaps] then dx[ntaps], the phases concatenated"""
        key = ('phases',) + tuple(tuple(p) for p in phases)
        off = self._tap_cache.get(key)
        if off is None:
            off = self._tap_cache[key] = len(self.taps)
            for p in phases:
                self.taps += np.asarray([t[0] for t in p] + [t[1] for t in p], dtype=np.int8).tobytes()
        return off

    def op(self, code, buf=(), i=(), f=(), l=(), taps=None, stream=0, desc=None, flops=0.0):
        assert len(buf) <= 8 and len(i) <= N_INTS and len(f) <= 4 and len(l) <= 2, (code, len(buf), len(i))
        self.ops.append((code, stream, tuple(-1 if b is None else int(b) for b in buf), tuple(int(v) for v in i),
                         tuple(float(v) for v in f), tuple(int(v) for v in l),
                         -1 if taps is None else (self.tap_lists(taps) if code == OP_CONV_GEMM_PHASES else self.tap_pair(taps)),
                         desc, flops))

    def base(self, arena):
        """slot of an arena's first byte"""
        self.arenas.setdefault(arena, [])
        self.kinds.append(('base', arena))
        return len(self.kinds) - 1

    def finish(self):
        """Zero-initialised pieces are laid out first in their arena: one memset record per arena, ahead of everything."""
        zero = [(n, sum(p[0] for p in pieces if p[1])) for n, pieces in sorted(self.arenas.items())]
        for n, nbytes in reversed(zero):
            if nbytes:
                self.ops.insert(0, (OP_MEMSET, 0, (self.base(n),), (), (), (nbytes,), -1, None, 0.0))
        return Program(self)


class Program:
    def __init__(self, b):
        if not b.ops:
            raise ValueError('empty program')
        self.n_ops, self.n_slots = len(b.ops), len(b.kinds)
        self.kinds = list(b.kinds)
        self.desc = [(OP_NAMES[o[0]], o[7], o[8]) for o in b.ops]
        # ---- arena layout: the zero-initialised pieces first (one memset per arena and run)
        self.arena_bytes, self.arena_zero = {}, {}
        for name, pieces in b.arenas.items():
            off = 0
            for want_zero in (True, False):
                for p in pieces:
                    if p[1] == want_zero:
                        p[2] = off
                        off += p[0]
                if want_zero:
                    self.arena_zero[name] = off
            self.arena_bytes[name] = off
        self.arena_names = sorted(b.arenas)
        self._static = [(s, k[1]) for s, k in enumerate(b.kinds) if k[0] == 'static']
        self._static_idx = np.asarray([s for s, _ in self._static], dtype=np.int64)
        self._ext = {k[1]: s for s, k in enumerate(b.kinds) if k[0] == 'ext'}
        a_slots = [(s, k[1], b.arenas[k[1]][k[2]][2] if k[0] == 'arena' else 0) for s, k in enumerate(b.kinds)
                   if k[0] in ('arena', 'base')]
        self._arena_idx = np.asarray([s for s, _, _ in a_slots], dtype=np.int64)
        self._arena_of = np.asarray([self.arena_names.index(n) for _, n, _ in a_slots], dtype=np.int64)
        self._arena_off = np.asarray([o for _, _, o in a_slots], dtype=np.uint64)
        self._offset = {s: o for s, _, o in a_slots}
        self._arena_name = {s: n for s, n, _ in a_slots}
        # ---- the C-side copy
        arr = (SdnOp * self.n_ops)()
        for r, (code, stream, buf, iv, fv, lv, taps, _d, _fl) in zip(arr, b.ops):
            r.code, r.stream, r.taps = code, stream, taps
            for k in range(8):
                r.buf[k] = buf[k] if k < len(buf) else -1
            for k, v in enumerate(iv):
                r.i[k] = v
            for k, v in enumerate(fv):
                r.f[k] = v
            if _fl and code in _TIMED_CODES:
                r.f[3] = _fl / 1e9     # the record's algorithmic GFLOP from the layer's TRUE channel counts (timing slots)
            for k, v in enumerate(lv):
                r.l[k] = v
        self._records = arr            # kept: tests decode programs through it (tests/trace_stub.py)
        blob = bytes(b.taps)
        self._blob = (ctypes.c_int8 * max(len(blob), 1)).from_buffer_copy(blob or b'\0')
        self._handle = ctypes.c_void_p()
        check(lib().sdn_program_create(arr, self.n_ops, self._blob, len(blob), self.n_slots, ctypes.byref(self._handle)))

    def __del__(self):
        h = getattr(self, '_handle', None)
        if h:
            try:
                lib().sdn_program_destroy(h)
            except Exception:
                pass

    # ---- run-time helpers
    def offset(self, slot):
        """byte offset of an arena slot inside its arena"""
        return self._offset[slot]

    def new_arenas(self, device):
        """{name: uint8 tensor}: one allocation per arena of the program (torch's caching allocator, current stream)"""
        out = {n: torch.empty(max(self.arena_bytes[n], ALIGN), dtype=torch.uint8, device=device)
               for n in self.arena_names}
        if os.environ.get('SDN_DEBUG_POISON') == '1':
            # every byte 0xFF = NaN in fp32 / fp64: a record that reads an arena piece nobody wrote (and the program's own
            # memset did not clear) then shows up as NaN instead of depending on what the allocator handed out
            for t in out.values():
                t.fill_(255)
        return out

    def view(self, arenas, slot, shape, dtype=torch.float32):
        """a tensor over an arena slot"""
        a = arenas[self._arena_name[slot]]
        n = 1
        for d in shape:
            n *= int(d)
        off = self._offset[slot]
        return a[off:off + n * dtype.itemsize].view(dtype).view(shape)

    def run(self, arenas, ext, main_stream, side_stream=None, timed=False):
        """arenas {name: tensor}; ext {name: tensor or int pointer}; streams as raw handles (ints / c_void_p / None)."""
        table = np.zeros(self.n_slots, dtype=np.uint64)
        if len(self._static):
            table[self._static_idx] = [(t() if callable(t) else t).data_ptr() for _, t in self._static]
        if len(self._arena_idx):
            bases = np.asarray([arenas[n].data_ptr() for n in self.arena_names], dtype=np.uint64)
            table[self._arena_idx] = bases[self._arena_of] + self._arena_off
        for name, s in self._ext.items():
            v = ext[name]
            table[s] = v if isinstance(v, int) else v.data_ptr()
        ms = (ctypes.c_float * self.n_ops)() if timed else None
        failed = ctypes.c_int(-1)
        main = main_stream if isinstance(main_stream, ctypes.c_void_p) else ctypes.c_void_p(main_stream)
        side = main if side_stream is None else (side_stream if isinstance(side_stream, ctypes.c_void_p)
                                                 else ctypes.c_void_p(side_stream))
        rc = lib().sdn_program_run(self._handle, table.ctypes.data_as(ctypes.POINTER(ctypes.c_void_p)), self.n_slots,
                                   main, side, ms, ctypes.byref(failed))
        if rc != 0:
            what = self.desc[failed.value] if 0 <= failed.value < self.n_ops else None
            from . import SdnHipError
            raise SdnHipError('libsdn_hip error %d in record %d %r: %s' % (rc, failed.value, what,
                                                                           lib().sdn_last_error().decode()))
        return list(ms) if timed else None
