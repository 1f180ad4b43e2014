"""Executor of conv + InstanceNorm + activation chains on the HIP kernels (sdn_conv_* / sdn_in_* in include/sdn_hip.h).

The textural networks of the reference (textural/models/networks.py) are nn.Sequential chains of
[ReflectionPad2d] Conv2d|ConvTranspose2d [InstanceNorm2d] [ReLU|LeakyReLU|Tanh] groups plus ResnetBlocks.  The product
modules in textural/models/networks.py keep those torch modules as PARAMETER CONTAINERS (identical state_dict keys and
initialisation) and hand the chain to this executor, which runs the whole forward and backward on channels-last fp32
buffers through the C ABI: one autograd.Function per chain, no torch convolution anywhere.  There is no CPU path: a
tensor that is not on the GPU raises.

Launch lists (r03).  A pass of a chain is 50-400 kernel launches whose arguments depend only on the chain and the input
shape, so each pass is PLANNED once per (chain, shape, mode) into a `sdn_hip.program.Program` -- an array of records naming
the library's launchers -- and replayed with ONE C call (`sdn_program_run`); a pass costs the host two allocations (one
arena for everything the pass produces), a pointer table and that call instead of a Python / ctypes round trip per launch
(the GAN step had become host-bound: 77 of 78 ms were the host issuing ~1800 launches).  The planning code below is the
former per-launch executor with "launch" replaced by "append a record".

Stored tensors and the deferred ReLU: a stage with a norm stores xhat = (z - mean) * rstd; if its activation is ReLU the
tensor is flagged `relu` and every consumer (next conv's loader, residual add, weight-gradient loader) applies
max(., 0) on the fly, so the backward pass still has xhat.  LeakyReLU (discriminator features, which are returned to
the caller) is materialised and inverted in the backward kernels.
"""
import collections
import ctypes
import os
import weakref

import torch

from . import check, lib, stream
from . import convplan as cp
from . import program as pg

ACT = {'none': 0, 'lrelu': 1, 'tanh': 2, 'relu': 3}
STAT_SLOTS = 8  # SDN_STAT_SLOTS in include/sdn_hip.h


def default_precision():
    """3 = bf16x3 split products (fp32-class, the parity-gated default); 1 = plain bf16 (SDN_CONV_PRECISION=1)."""
    return int(os.environ.get('SDN_CONV_PRECISION', '3'))


# ---- validity of the packed-weight caches.  A parameter's `_version` counter catches in-place updates made through autograd
# aware ops (load_state_dict, plain Adam) -- but NOT torch's fused / foreach optimizer kernels (torch.optim.Adam(fused=True)
# leaves `_version` untouched), so every optimizer step also stamps the parameters THAT optimizer owns with a fresh epoch,
# which is part of the cache tag (per parameter: the generator's step must not throw away the discriminator's packed
# weights, which the discriminator's own backward pass reuses right after it).  Writes through `param.data` bump neither:
# call sdn_hip.conv.invalidate_weight_caches() after them (weights_init does).
_WEIGHT_EPOCH = [0]   # advanced by invalidate_weight_caches(): invalidates everything
_STEP_COUNT = [0]
_PARAM_EPOCH = {}     # id(parameter) -> number of the last optimizer step that updated it


def invalidate_weight_caches(*_args, **_kwargs):
    _WEIGHT_EPOCH[0] += 1


def _on_optimizer_step(optimizer, *_args, **_kwargs):
    _STEP_COUNT[0] += 1
    e = _STEP_COUNT[0]
    for group in optimizer.param_groups:
        for prm in group['params']:
            _PARAM_EPOCH[id(prm)] = e


from torch.optim.optimizer import register_optimizer_step_post_hook as _register_step_hook  # noqa: E402

_register_step_hook(_on_optimizer_step)


def _tag(t):
    return (t._version, t.data_ptr(), _WEIGHT_EPOCH[0], _PARAM_EPOCH.get(id(t), 0))


def deterministic():
    """True: split-K partial sums are combined in a fixed order (bit-reproducible gradients) instead of with float
    atomics.  Follows torch's own switch -- torch.use_deterministic_algorithms(True) -- or SDN_DETERMINISTIC=1."""
    return torch.are_deterministic_algorithms_enabled() or os.environ.get('SDN_DETERMINISTIC') == '1'


_side_streams = {}


def _wgrad_stream(dev):
    """The HIP stream the weight-gradient launches of a backward pass go to (SDN_WGRAD_STREAM=0: the caller's).  The data
    gradient chain is the critical path of a backward pass and its launches leave CUs idle (544 tiles on 768 workgroup
    slots for the residual blocks); the weight gradients only feed the optimizer, so they run beside it.  One side stream
    per calling stream (the discriminator columns already run on streams of their own)."""
    if os.environ.get('SDN_WGRAD_STREAM', '1') == '0':
        return None
    cur = torch.cuda.current_stream(dev)
    key = (dev.index, cur.cuda_stream)
    st = _side_streams.get(key)
    if st is None:
        st = _side_streams[key] = torch.cuda.Stream(device=dev)
    return st


PROFILE = None  # development aid: set to a list to collect (what, stage description, ms, flops) per timed record


def _run(program, arenas, ext, side=None):
    """Replay a program on the current stream (+ the side stream)."""
    side_h = None if side is None else ctypes.c_void_p(side.cuda_stream)
    if PROFILE is None:
        program.run(arenas, ext, stream(), side_h)
        return
    ms = program.run(arenas, ext, stream(), side_h, timed=True)
    for (name, desc, flops), t in zip(program.desc, ms):
        if desc is not None:
            PROFILE.append((desc[0], desc[1], t, flops))


class _Packed:
    """A persistent device buffer derived from a stage's parameters (packed MFMA weights, a dense fp32 tap window, a padded
    bias) + the parameter tag it was last refreshed for.  Programs hold the buffer as a static slot, so it is refreshed IN
    PLACE: by records of a plan's pack program (`emit`) or by torch ops (`refresh`)."""
    __slots__ = ('buf', 'tag', 'emit', 'refresh', 'params', 'meta')

    def __init__(self, buf, params, emit=None, refresh=None, meta=None):
        self.buf, self.tag, self.emit, self.refresh, self.params, self.meta = buf, None, emit, refresh, params, meta

    def current(self):
        return tuple(_tag(p()) for p in self.params)


class Stage:
    """One conv group of a chain.  kind 'conv' | 'convT'; conv/norm are the torch modules holding the parameters."""

    def __init__(self, kind, conv, src, reflect=0):
        self.kind = kind
        self.conv = conv
        self.src = src          # index of the input tensor in the chain's tensor list
        self.reflect = reflect  # ReflectionPad2d amount folded into the gather (0: the conv's own zero padding)
        self.norm = None
        self.act = 'none'
        self.res = None         # tensor index added after the norm (ResnetBlock)
        self.k = conv.kernel_size[0]
        self.s = conv.stride[0]
        self.p = reflect if reflect else conv.padding[0]
        self.op = conv.output_padding[0] if kind == 'convT' else 0
        if kind == 'conv':
            self.cout, self.cin = conv.weight.shape[0], conv.weight.shape[1]
        else:
            self.cin, self.cout = conv.weight.shape[0], conv.weight.shape[1]
        kk = self.k * self.k
        # strides (row, col) of the parameter tensor for the orientations we pack
        if kind == 'conv':      # [O, I, kh, kw]
            self.str_fwd = (self.cin * kk, kk)      # rows = cout, cols = cin
            self.str_dgrad = (kk, self.cin * kk)    # rows = cin,  cols = cout
        else:                   # [I, O, kh, kw]
            self.str_fwd = (kk, self.cout * kk)     # rows = cout, cols = cin
            self.str_dgrad = (self.cout * kk, kk)   # rows = cin,  cols = cout
        self._packed = {}
        self._tix = {}

    def _w(self):
        return self.conv.weight

    def _b(self):
        return self.conv.bias

    def tix(self, tapidx, dev):
        """device int32 copy of a tap-index list (cached: a fresh H2D copy per launch costs more than the kernel)"""
        key = (tuple(tapidx), str(dev))
        t = self._tix.get(key)
        if t is None:
            t = self._tix[key] = torch.tensor(list(tapidx), dtype=torch.int32, device=dev)
        return t

    def padded_bias(self, cop):
        """_Packed of the bias zero-padded to cop channels, or None (no bias) / the parameter itself (no padding)."""
        b = self.conv.bias
        if b is None:
            return None
        if cop == self.cout:
            return self._b
        key = ('bias', cop)
        e = self._packed.get(key)
        if e is None:
            buf = torch.zeros(cop, dtype=torch.float32, device=b.device)
            cout = self.cout

            def emit(bld, e=None):
                bld.op(pg.OP_COPY, buf=[bld.static(buf), bld.static(lambda: self.conv.bias.detach())], l=[4 * cout])
            e = self._packed[key] = _Packed(buf, (self._b,), emit=emit)
        return e

    # ---- packed weights, refreshed when the parameter changes (see _tag above)
    def packed(self, which, tapidx, precision, ccp, out_cp, rows_range=None):
        """which 'fwd': rows = cout, cols = cin;  'dgrad': rows = cin, cols = cout.  ccp: padded channel count of the
        tensor the gemm reads, out_cp: of the tensor it writes (selects the N tile, hence the row padding).
        rows_range (lo, hi): only these rows of the logical matrix (a data gradient for some input channels).
        -> _Packed with meta (Kp, rows)."""
        key = (which, tuple(tapidx), precision, ccp, out_cp, rows_range)
        e = self._packed.get(key)
        if e is not None:
            return e
        w = self.conv.weight
        if which == 'fwd':
            R, C, (sr, sc) = self.cout, self.cin, self.str_fwd
        else:
            R, C, (sr, sc) = self.cin, self.cout, self.str_dgrad
        row0 = 0
        if rows_range is not None:
            row0, R = rows_range[0], rows_range[1] - rows_range[0]
        assert ccp >= C and out_cp >= R
        rows = cp.weight_rows(out_cp)
        Kp = cp.kpad(len(tapidx), ccp)
        tix = self.tix(tapidx, w.device)
        buf = torch.empty(2 * rows * Kp, dtype=torch.bfloat16, device=w.device)  # fragment-major hi / lo blocks
        ntaps = len(tapidx)
        first = row0 * sr

        def source():
            wd = self.conv.weight.detach()
            return wd if first == 0 else wd.reshape(-1)[first:]

        def emit(bld):
            bld.op(pg.OP_PACK_WEIGHTS, buf=[bld.static(source), bld.static(tix), bld.static(buf)],
                   i=[R, C, ntaps, ccp, Kp, rows], l=[sr, sc])
        e = self._packed[key] = _Packed(buf, (self._w,), emit=emit, meta=(Kp, rows))
        return e

    def packed_kmajor(self, which, tapidx, ccp, out_cp):
        """K-major (hi, lo) weights for sdn_conv_tile (csrc/conv_tile.hip): [rows][step][2][32] bf16, step = channel block *
        ntaps + tap.  -> _Packed with meta (rows,)."""
        key = ('kmajor', which, tuple(tapidx), ccp, out_cp)
        e = self._packed.get(key)
        if e is not None:
            return e
        w = self.conv.weight
        if which == 'fwd':
            R, C, (sr, sc) = self.cout, self.cin, self.str_fwd
        else:
            R, C, (sr, sc) = self.cin, self.cout, self.str_dgrad
        assert ccp >= C and out_cp >= R and ccp % 32 == 0
        rows = cp.tile_weight_rows(out_cp)
        ntaps = len(tapidx)
        tix = self.tix(tapidx, w.device)
        buf = torch.empty(2 * rows * ntaps * ccp, dtype=torch.bfloat16, device=w.device)

        def emit(bld):
            bld.op(pg.OP_PACK_WEIGHTS_KMAJOR, buf=[bld.static(lambda: self.conv.weight.detach()), bld.static(tix), bld.static(buf)],
                   i=[R, C, ntaps, ccp, rows], l=[sr, sc])
        e = self._packed[key] = _Packed(buf, (self._w,), emit=emit, meta=(rows,))
        return e

    def narrow(self, which, taps, tapidx, ccp, rows_range=None):
        """Dense fp32 tap window [KH, KW, ccp, RP] for sdn_conv_narrow_fwd (layers with <= 8 rows), cached like packed().
        which 'fwd': rows = cout, cols = cin;  'dgrad': rows = cin[rows_range], cols = cout.
        -> _Packed with meta (KH, KW, dy_min, dx_min, R)."""
        key = ('narrow', which, tuple(taps), ccp, rows_range)
        e = self._packed.get(key)
        if e is not None:
            return e
        w = self.conv.weight
        kk = self.k * self.k
        R = (w.shape[0] if which == 'fwd' else w.shape[1])
        if rows_range is not None:
            R = rows_range[1] - rows_range[0]
        # (the fp32 vector kernels take 1 / 4 / 8 rows; wider windows -- 16 rows, or four groups of 16 -- only feed head_mfma())
        RP = 1 if R == 1 else (4 if R <= 4 else (8 if R <= 8 else (16 if R <= 16 else 64)))
        assert R <= RP
        dys, dxs = [t[0] for t in taps], [t[1] for t in taps]
        dy_min, dx_min = min(dys), min(dxs)
        KH, KW = max(dys) - dy_min + 1, max(dxs) - dx_min + 1
        dense = torch.zeros(KH, KW, ccp, RP, dtype=torch.float32, device=w.device)
        iy = torch.tensor([d - dy_min for d in dys], device=w.device)
        ix = torch.tensor([d - dx_min for d in dxs], device=w.device)
        it = torch.tensor(list(tapidx), device=w.device)

        def refresh():
            wd = self.conv.weight.detach()
            m = wd.reshape(wd.shape[0], wd.shape[1], kk)           # Conv2d: [cout, cin, taps]
            a = m if which == 'fwd' else m.permute(1, 0, 2)        # [rows, cols, taps]
            if rows_range is not None:
                a = a[rows_range[0]:rows_range[1]]
            dense[iy, ix, :a.shape[1], :a.shape[0]] = a[:, :, it].permute(2, 1, 0)
        e = self._packed[key] = _Packed(dense, (self._w,), refresh=refresh, meta=(KH, KW, dy_min, dx_min, R))
        return e

    def head_mfma(self, which, taps, tapidx, ccp, rows_range=None):
        """The same dense tap window as narrow(), pre-split to bf16 (hi, lo) in the fragment order of sdn_conv_head_mfma
        (csrc/conv_head.hip): [row groups][steps][2][64 lanes][8], element (rg, s, part, lane, j) = weight of output row
        16 rg + lane % 16 at the 8-channel slot u = 4 s + lane // 16, u = tap * (ccp // 8) + channel group, tap = ky * KW + kx
        (one row group for up to 16 rows: the buffer is then [steps][2][64][8]).  Built by torch ops from narrow()'s buffer (a few
        thousand values per head).  -> _Packed with narrow()'s meta."""
        key = ('head_mfma', which, tuple(tapidx), ccp, rows_range)
        e = self._packed.get(key)
        if e is not None:
            return e
        nar = self.narrow(which, taps, tapidx, ccp, rows_range)
        KH, KW, dy_min, dx_min, R = nar.meta
        dense = nar.buf                                            # [KH, KW, ccp, RP]
        RP, CG, ntaps = dense.shape[3], ccp // 8, KH * KW
        U = ntaps * CG
        S = (U + 3) // 4
        RG = (RP + 15) // 16                                       # row groups of 16 (r06: 4 for a 64-row data gradient)
        buf = torch.zeros((S, 2, 64, 8) if RG == 1 else (RG, S, 2, 64, 8), dtype=torch.bfloat16, device=dense.device)
        bufv = buf.view(RG, S, 2, 64, 8)
        full = torch.zeros(16 * RG, S * 4, 8, dtype=torch.float32, device=dense.device)

        def refresh():
            nar.refresh()
            full[:RP, :U] = dense.reshape(ntaps, CG, 8, RP).permute(3, 0, 1, 2).reshape(RP, U, 8)
            frag = full.reshape(RG, 16, S, 4, 8).permute(0, 2, 3, 1, 4).reshape(RG, S, 64, 8)   # lane = k-group * 16 + output row
            hi = frag.to(torch.bfloat16)
            bufv[:, :, 0] = hi
            bufv[:, :, 1] = (frag - hi.float()).to(torch.bfloat16)
        e = self._packed[key] = _Packed(buf, (self._w,), refresh=refresh, meta=nar.meta)
        return e


def _head_mfma_ok(KH, KW, Cip, Cop, precision):
    """sdn_conv_head_mfma (r05, 'm' in SDN_TILE_KERNELS): 7 x 7 windows over 16 or 64 (padded) input channels into a 16-channel
    output tensor -- the generator / encoder heads and the stem's data gradient towards the encoder features; bf16 x 3 like
    the other MFMA layers (the deterministic mode keeps it: there are no atomics in it).  r06: also 64 output channels over a
    16-channel input (four row groups: the generator head's data gradient)."""
    return ('m' in tile_kernels() and precision == 3 and KH == 7 and KW == 7 and Cip in (16, 64)
            and (Cop == 16 or (Cop == 64 and Cip == 16)))


def _head16_fwd_ok(st, Cip, Cop, precision, det):
    """r06: a 7 x 7 stride-1 conv into <= 16 channels UNDER InstanceNorm -- the encoder's stem, networks.py:291-293 (3 -> 16) -- on
    sdn_conv_head_mfma with its statistics epilogue (float64 atomics into the slots: as every other conv epilogue, also in
    deterministic mode).  Was sdn_conv_gemm with 13 of 16 K channels zero: 0.58 ms at 384 x 1248, batch 4."""
    return (_head_wide() and st.kind == 'conv' and st.s == 1 and st.k == 7 and st.cout <= 16 and st.res is None
            and _head_mfma_ok(7, 7, Cip, Cop, precision))


def _head_wgrad_ok(st, Cip, Cop, precision, det):
    """r06: weight gradient of a 7 x 7 stride-1 conv whose d(out) fits 16 channels on sdn_conv_wgrad_head_mfma.  Float atomics: not
    in deterministic mode.  SDN_WGRAD_HEAD: 0 = off; 1 = layers with a 16-channel (padded) input -- the encoder stem 3 -> 16 (was the
    64-row tile kernel at 12 TFLOP/s) and the encoder head 16 -> 5 (was exact fp32 on the vector ALUs, sdn_conv_wgrad_narrow); 2 = also
    the generator head 64 -> 3, where the vector-ALU kernel is as fast alone and shares the chip better with the MFMA chain."""
    level = int(os.environ.get('SDN_WGRAD_HEAD', '1'))
    return (level > 0 and not det and precision == 3 and st.kind == 'conv' and st.s == 1 and st.k == 7 and Cop == 16
            and (Cip == 16 or (Cip == 64 and level > 1)))


def _head_wide():
    """SDN_HEAD_WIDE=0: the r05 routing of the 7 x 7 narrow-channel layers (A/B switch of the r06 head-kernel extensions)"""
    return os.environ.get('SDN_HEAD_WIDE', '1') != '0'


class _T:
    """A tensor of the chain: channels-last padded buffer + logical facts."""
    __slots__ = ('data', 'C', 'relu', 'xhat', 'stats', 'mode')

    def __init__(self, data, C, relu=False):
        self.data = data
        self.C = C
        self.relu = relu      # consumers apply ReLU on load
        self.xhat = None      # for residual stages: the normalised conv output (data = res + xhat)
        self.stats = None
        self.mode = 0


class _PT:
    """A planned tensor of the chain: slots instead of buffers."""
    __slots__ = ('slot', 'C', 'Cp', 'H', 'W', 'relu', 'xhat', 'mr', 'mode', 'pl', 'pls')

    def __init__(self, slot, C, Cp, H, W, relu=False):
        self.slot, self.C, self.Cp, self.H, self.W, self.relu = slot, C, Cp, H, W, relu
        self.xhat = None      # slot of the normalised conv output of a residual stage (slot = res + xhat)
        self.mr = None        # slot of (mean, rstd) per (n, c)
        self.mode = 0
        self.pl = None        # slot of the bf16 (hi, lo) operand planes of what consumers multiply (ReLU applied), r04
        self.pls = 0          # elements between the two planes


NARROW_KW = (3, 4, 7)  # window sizes sdn_conv_narrow_fwd is built for
PLAN_CACHE = 8         # compiled forward plans a chain keeps (one per input shape / mode); backward plans: four times as many


def plane_stride(n):
    """elements between the hi and the lo plane of an n-element tensor (a multiple of 8: 16-byte aligned planes)"""
    return (int(n) + 7) // 8 * 8


def tile_kernels():
    """SDN_TILE_KERNELS: which of the r04 tiled MFMA kernels (LDS-DMA on bf16 operand planes) the executor uses -- a subset
    of 'w' (weight gradients: sdn_conv_wgrad_tile), 'f' (forward launches of wide layers: sdn_conv_tile), 'h' (stride-1
    3x3 / 4x4 forward launches with the input patch staged in LDS: sdn_conv_halo) and 'd' (data gradients of wide stride-1
    layers: sdn_conv_tile with a K-split tail), 'p' (r05: the phase launches of transposed convs / strided data gradients on
    sdn_conv_gemm as one sdn_conv_gemm_phases launch), 'm' (r05: the 7 x 7 head layers on sdn_conv_head_mfma instead of the
    fp32 vector kernel); default all, '' = the r03 kernels."""
    return os.environ.get('SDN_TILE_KERNELS', 'wfhdpm')


_CUS = []


def compute_units():
    """CUs of the current device (256 on an MI355X), asked once -- the grid rules below count rounds of one workgroup per
    CU, and csrc/conv_wtile.hip sizes its stream-K grid from the same device property (the rules used a literal
    256).  Without a GPU (planning tests on the CPU stub) the MI355X's figure."""
    if not _CUS:
        try:
            _CUS.append(int(torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count)
                        if torch.cuda.is_available() else 256)
        except Exception:
            _CUS.append(256)
    return _CUS[0]


def _tile_fwd_ok(st, launches, N, Cip, Cop, precision):
    """sdn_conv_tile for a forward stage: 32-channel K steps, a wide N tile, and a launch grid that fills whole rounds of the
    256 CUs (one 256 x 128 tile per CU at a time: 288 tiles would take as long as 512).  Measured against sdn_conv_gemm on
    the layer shapes of the GAN step (profiles/r04*_tile_lab.log): 1.02-1.08 x where this holds, slower where it does not."""
    if 'f' not in tile_kernels() or precision != 3 or Cip % 32 or Cop < 256:
        return False
    for L in launches:
        if not L.taps:
            return False
        cus = compute_units()
        tiles = ((L.QH * L.QW + 255) // 256) * N * ((Cop + 127) // 128)
        rounds = (tiles + cus - 1) // cus
        if tiles < cus // 2 or tiles < 0.8 * rounds * cus:
            return False
    return True


def _tile_dgrad_plan(st, launches, N, GH, GW, Cz, Cg, precision, det, acc):
    """sdn_conv_tile for a data gradient: -> None, or the K split of the tail tiles (0 = none).  One dense launch (stride-1
    layers), dz's channels in 32-deep K steps, >= 256 gradient channels.  The grid rule of _tile_fwd_ok, with one more
    option: when Q = GH * GW leaves a partial last tile per image (26 x 80 = 8 x 256 + 32), that tile is cut into K slices
    (float atomics, so not in deterministic mode, and not onto a gradient that is already there).  Measured on the 1024-channel
    residual layers (profiles/r04*_layer_times*.log): 0.51 ms on sdn_conv_gemm, 2 rounds = 0.74 ms unsplit."""
    if 'd' not in tile_kernels() or precision != 3 or Cz % 32 or Cg < 256 or len(launches) != 1:
        return None
    L = launches[0]
    if not L.taps or L.istride != 1 or L.ostride != 1 or L.py or L.px or (L.QH, L.QW) != (GH, GW):
        return None
    Q, nt = GH * GW, (Cg + 127) // 128
    mt = (Q + 255) // 256
    cus = compute_units()
    useful = N * Q * nt / 256.0                                  # tiles' worth of MFMA rows that are real outputs
    tiles = N * mt * nt
    best, split = useful / (((tiles + cus - 1) // cus) * float(cus)), 0
    nsteps = len(L.taps) * (Cz // 32)
    if not det and not acc and Q % 256:
        full, tail = N * (mt - 1) * nt, N * nt
        for S in (16, 8, 4, 2):
            if tail * S <= cus and nsteps >= 8 * S:
                eff = useful / ((((full + cus - 1) // cus) + 1.0 / S + 0.02) * float(cus))
                if eff > best + 0.05:
                    best, split = eff, S
                break
    if tiles < cus // 2 or best < 0.8:
        return None
    return split


def _halo_fwd_ok(st, launches, N, OH, OW, Cip, Cop, precision):
    """sdn_conv_halo for a forward stage: one stride-1 launch whose taps fill a window of >= 9 taps (3x3, 4x4), 32-channel K
    steps, 128-channel N tiles, and a grid that fills whole rounds of the CUs.  Measured (profiles/r04*_tile_lab.log): the
    1024-channel residual layers 0.358 ms against 0.368 (sdn_conv_tile) and 0.416 (sdn_conv_gemm), at 2.2 x less L2 traffic."""
    if 'h' not in tile_kernels() or precision != 3 or Cip % 32 or Cop < 256 or len(launches) != 1:
        return False
    L = launches[0]
    if L.istride != 1 or L.ostride != 1 or L.py or L.px or len(L.taps) < 9 or (L.QH, L.QW) != (OH, OW):
        return False
    dys, dxs = [t[0] for t in L.taps], [t[1] for t in L.taps]
    kh, kw = max(dys) - min(dys) + 1, max(dxs) - min(dxs) + 1
    if kh * kw != len(L.taps):
        return False
    th, tw, blocks = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_long(0)
    check(lib().sdn_conv_halo_blocks(N, OH, OW, Cop, kh, kw, ctypes.byref(th), ctypes.byref(tw), ctypes.byref(blocks)))
    cus = compute_units()
    if blocks.value < cus // 2:
        return False
    rounds = (blocks.value + cus - 1) // cus
    useful = N * OH * OW * ((Cop + 127) // 128) / 256.0        # blocks' worth of MFMA rows that are real outputs
    return useful >= 0.8 * rounds * cus


def _tile_wgrad_ok(st, N, QH, QW, Cr, GH, GW, Cc, precision, det):
    """sdn_conv_wgrad_tile: stream-K over one workgroup per CU, float atomics (so not in deterministic mode)"""
    if 'w' not in tile_kernels() or precision != 3 or det:
        return False
    if st.kind == 'conv' and st.s == 1 and st.cout <= 8:
        return False                      # head layers: the narrow fp32 kernel
    npos = N * QH * QW
    return npos < (1 << 24) and npos * Cr * 2 < 0xffffff00 and N * GH * GW * Cc * 2 < 0xffffff00


def update_running(running):
    """running <- (1 - momentum) * running + momentum * batch statistic, for the (norm, mean, var) triples a dual pass
    collected: what one training-mode forward does to nn.InstanceNorm2d(track_running_stats=True)."""
    groups = {}   # one multi-tensor launch pair per momentum value instead of four small launches per layer
    for nm, bm, bv in running:
        m = float(nm.momentum if nm.momentum is not None else 0.1)
        dst, src = groups.setdefault(m, ([], []))
        dst += [nm.running_mean, nm.running_var]
        src += [bm, bv]
    with torch.no_grad():
        for m, (dst, src) in groups.items():
            torch._foreach_mul_(dst, 1.0 - m)
            torch._foreach_add_(dst, src, alpha=m)


class _Plan:
    """A compiled pass: the program, the packed buffers it reads (+ the pack program that refreshes them)."""

    def __init__(self, builder, packs):
        self.program = builder.finish()
        self.kinds = self.program.kinds
        seen, self.packs = set(), []
        for e in packs:
            if e is not None and not callable(e) and id(e) not in seen:
                seen.add(id(e))
                self.packs.append(e)
        self._pack_program = None

    def refresh_packs(self):
        """Re-pack what the parameters' tags say is stale -- one launch list for the whole plan."""
        stale = [(e, t) for e, t in ((e, e.current()) for e in self.packs) if e.tag != t]
        stale.sort(key=lambda et: et[0].meta is None)     # weight packs first, then the bias copies: runs of one launch each
        if os.environ.get('SDN_DEBUG_CHECKS') == '1':
            self._check_fresh([e for e in self.packs if not any(e is s_ for s_, _ in stale)])
        if not stale:
            return
        if len(stale) != len(self.packs) or self._pack_program is None:
            b = pg.Builder()
            for e, _ in stale:
                if e.emit is not None:
                    e.emit(b)
            prog = b.finish() if b.ops else None
            if len(stale) == len(self.packs):
                self._pack_program = prog or False
        else:
            prog = self._pack_program
        if prog:
            _run(prog, {}, {})
        for e, t in stale:
            if e.refresh is not None:
                e.refresh()
            e.tag = t
        if os.environ.get('SDN_DEBUG_CHECKS') == '1':
            self._check_fresh([e for e, _ in stale])    # (a cached pack program must do what a fresh one does)

    def _check_fresh(self, packs):
        """SDN_DEBUG_CHECKS=1: every pack the tags call fresh is re-derived from the parameters and compared (a tag that
        misses an update would otherwise show up as a 1 % gradient error one optimizer step later)."""
        for e in packs:
            before = e.buf.clone()
            if e.emit is not None:
                b = pg.Builder()
                e.emit(b)
                _run(b.finish(), {}, {})
            if e.refresh is not None:
                e.refresh()
            torch.cuda.synchronize()
            if not torch.equal(before.view(torch.uint8).reshape(-1), e.buf.view(torch.uint8).reshape(-1)):
                raise RuntimeError('stale packed weights behind a fresh tag: tag %r, current %r, meta %r, steps %d'
                                   % (e.tag, e.current(), e.meta, _STEP_COUNT[0]))


def eager_repack(modules):
    """Refresh, on the CURRENT stream, the packed weights of the plans the chains of `modules` ran last (forward and backward).
    r06: a pass used to re-pack stale weights at its own start -- after an optimizer step that is ~0.3 GB of HBM-bound traffic
    per 100 M parameters in front of the next forward pass, on the critical path.  Pix2PixHDModel.train_step calls this on a side
    stream right behind the generator's optimizer step, beside the discriminator's backward pass (MFMA-bound and independent of
    the generator's weights); the next pass then finds every tag current.  `modules`: nn.Modules whose sub-modules carry chains
    (the `_chains` caches of textural/models/networks.py)."""
    for mod in modules:
        for m in mod.modules():
            cache = m.__dict__.get('_chains')
            if not cache or cache.get('__owner__') != id(m):
                continue
            for key, chain in cache.items():
                if key == '__owner__' or not chain._fwd_plans:
                    continue
                fplan = next(reversed(chain._fwd_plans.values()))        # most recently used
                fplan.refresh_packs()
                for bkey, (bplan, ref) in chain._bwd_plans.items():
                    if bkey[0] == id(fplan) and ref() is fplan and bplan is not None:
                        bplan.refresh_packs()


def _bias_slot(b, packs, bias):
    """bias: None | callable (the parameter itself) | _Packed (zero-padded copy)"""
    if bias is None:
        return None
    if callable(bias):
        return b.static(lambda: bias().detach())
    packs.append(bias)
    return b.static(bias.buf)


def _emit_gemm(b, packs, st, which, x_slot, N, IH, IW, Cip, out_slot, OH, OW, Cop, L, pad_mode, in_relu, precision, bias, act,
               stats, accumulate, ws, rows_range=None, desc=None, flops=0.0):
    e = st.packed(which, L.tapidx, precision, Cip, Cop, rows_range)
    packs.append(e)
    Kp, rows = e.meta
    wsn = 0
    if ws is not None:
        n = ctypes.c_size_t(0)
        check(lib().sdn_conv_gemm_workspace_bytes(N, OH, OW, Cop, ctypes.byref(n)))
        wsn = ws.need(n.value)
    b.op(pg.OP_CONV_GEMM, buf=[x_slot, out_slot, b.static(e.buf), bias, stats, ws.slot if ws is not None else None],
         i=[N, IH, IW, Cip, OH, OW, Cop, L.QH, L.QW, L.istride, L.ostride, L.py, L.px, len(L.taps), pad_mode, int(in_relu),
            Kp, rows, act, int(accumulate), precision], l=[wsn], taps=L.taps, desc=desc, flops=flops)


def _phases_ok(launches, precision, N=None, Cop=None, Cip=None):
    """sdn_conv_gemm_phases (r05): the 2-4 phase launches of a transposed conv / strided data gradient as ONE launch ('p' in
    SDN_TILE_KERNELS; every phase <= 16 taps, i.e. kernels up to 7 x 7 at stride 2).  The merged launch never splits K, the
    per-phase launches can: when all phases together have fewer 128 x 128 tiles than the chip has CUs AND a phase walks a long
    K (>= 2048: the deep layers at small N, where a few dozen workgroups would each run a very long loop) the per-phase
    launches stay (short-K layers such as the encoder's 256 -> 128 at batch 4 measured better merged)."""
    if not ('p' in tile_kernels() and 2 <= len(launches) <= 4 and all(L.taps and len(L.taps) <= 16 for L in launches)
            and len({(L.istride, L.ostride) for L in launches}) == 1):
        return False
    if N is not None and Cop is not None:
        tiles = sum(((L.QH * L.QW + 127) // 128) * N for L in launches) * ((Cop + 127) // 128)
        if tiles < compute_units() and Cip is not None and max(len(L.taps) for L in launches) * Cip >= 2048:
            return False
    return True


def _emit_gemm_phases(b, packs, st, which, x_slot, N, IH, IW, Cip, out_slot, OH, OW, Cop, launches, pad_mode, in_relu, precision,
                      bias, act, stats, accumulate, rows_range=None, desc=None, flops=0.0):
    """one sdn_conv_gemm_phases record for `launches` (what _emit_gemm would emit as one record per phase)"""
    es = [st.packed(which, L.tapidx, precision, Cip, Cop, rows_range) for L in launches]
    packs.extend(es)
    rows = es[0].meta[1]
    assert all(e.meta[1] == rows for e in es)
    ws = [b.static(e.buf) for e in es] + [None] * (4 - len(es))
    L0 = launches[0]
    ints = [N, IH, IW, Cip, OH, OW, Cop, L0.istride, L0.ostride, len(launches), pad_mode, int(in_relu), rows, act,
            int(accumulate), precision]
    for L, e in zip(launches, es):
        ints += [L.QH, L.QW, L.py, L.px, len(L.taps), e.meta[0]]
    b.op(pg.OP_CONV_GEMM_PHASES, buf=[x_slot, out_slot] + ws + [bias, stats], i=ints, taps=[L.taps for L in launches],
         desc=desc, flops=flops)


def _emit_tile(b, packs, st, which, X, N, IH, IW, Cip, out_slot, OH, OW, Cop, L, pad_mode, bias, act, stats, accumulate,
               desc=None, flops=0.0, ksplit=0):
    """one sdn_conv_tile record: the launch `L` of stage `st` reading the operand planes of X (a _PT, or (slot, stride))"""
    e = st.packed_kmajor(which, L.tapidx, Cip, Cop)
    packs.append(e)
    pl, pls = (X.pl, X.pls) if isinstance(X, _PT) else X
    b.op(pg.OP_CONV_TILE, buf=[pl, out_slot, None, b.static(e.buf), bias, stats],
         i=[N, IH, IW, Cip, 0, OH, OW, Cop, L.QH, L.QW, L.istride, L.ostride, L.py, L.px, len(L.taps), pad_mode, e.meta[0], act,
            int(accumulate), int(ksplit)], l=[pls, 0], taps=L.taps, desc=desc, flops=flops)


class _Workspace:
    """One shared scratch region per stream of a plan (ordered split-K sums): records on a stream run in order, so they
    can share it; its size is the largest request."""

    def __init__(self, b, arena):
        self.b, self.arena, self.bytes = b, arena, 0
        self.slot = b.alloc(arena, 1)
        self._piece = b.arenas[arena][-1]

    def need(self, nbytes):
        self.bytes = max(self.bytes, int(nbytes))
        self._piece[0] = pg._round(self.bytes)
        return int(nbytes)


class ConvChain:
    """Runs a list of Stages.  tensors[0] is the chain input; tensors[i + 1] the output of stage i.
    `outputs`: indices (into tensors) returned to the caller, in order."""

    def __init__(self, stages, outputs, in_channels):
        self.stages = stages
        self.outputs = outputs
        self.in_channels = in_channels
        self._fwd_plans = collections.OrderedDict()   # least recently used first; at most PLAN_CACHE entries (ADVICE r03:
        self._bwd_plans = collections.OrderedDict()   # inference on images of many sizes must not pile up C programs)

    def params(self):
        ps = []
        for st in self.stages:
            ps.append(st.conv.weight)
            ps.append(st.conv.bias)
        return ps

    def __call__(self, x_nchw, detach_weights=False, dual=False):
        """x [N, C, H, W] fp32 cuda -> list of [N, C_i, H_i, W_i] tensors (channels-last storage).
        x may also be a list / tuple of tensors that the reference would torch.cat along the channels first: they are
        written side by side into the channels-last input buffer, and the backward pass computes the input gradient
        only for the channel range of the parts that require one (e.g. the 5 encoder features of the generator's 48
        input channels, the 3 image channels of the discriminator's 18).
        detach_weights: the parameters take no part in autograd for this call (no weight-gradient launches).
        dual: ONE forward pass, two autograd views of it -- returns (outs_w, outs_x, running): outs_w is what
        chain(x.detach()) would return (gradients reach the parameters only), outs_x what
        chain(x, detach_weights=True) would (gradients reach x only).  Same values, same gradients as the two separate
        calls, one forward instead of two (the GAN step scores the same fake image once for each loss).  The norm
        layers' running statistics are left alone: `running` lists (norm, batch mean, batch variance) and the caller
        applies update_running(running) where each of the two passes it replaces would have run."""
        parts = list(x_nchw) if isinstance(x_nchw, (list, tuple)) else [x_nchw]
        for t in parts:
            if not t.is_cuda:
                raise NotImplementedError('the textural conv stack only runs on the GPU (got %s); there is no CPU or '
                                          'PyTorch fallback' % t.device)
            if t.dtype != torch.float32:
                raise TypeError('expected float32 input, got %s' % t.dtype)
        lib()
        C = sum(int(t.shape[1]) for t in parts)
        if C != self.in_channels:
            raise ValueError('expected %d input channels, got %d' % (self.in_channels, C))
        params = self.params()
        if dual:
            self._keep_state = True
            try:
                outs_w = _ChainFn.apply(self, len(parts), *[t.detach() for t in parts], *params)
                state = self.__dict__.pop('_state')
            finally:
                self._keep_state = False
            outs_x = _ChainSharedFn.apply(self, state, len(parts), *parts)
            return self._present(outs_w), self._present(outs_x), state.running
        if detach_weights:
            params = [p.detach() if p is not None else None for p in params]
        return self._present(_ChainFn.apply(self, len(parts), *parts, *params))

    def _present(self, outs):
        """channels-last padded chain outputs -> the NCHW views handed to the caller"""
        if not isinstance(outs, tuple):
            outs = (outs,)
        res = []
        for o, ti in zip(outs, self.outputs):
            st = self.stages[ti - 1]
            if st.cout != o.shape[-1]:
                o = o[..., :st.cout]  # (a full-range slice would still cost a zero fill + copy in its backward)
            o = o.permute(0, 3, 1, 2)
            if st.act == 'relu':
                # stored un-activated (consumers inside the chain apply ReLU on load): materialise for the caller.  The
                # chain's backward masks by the stored sign as well, which is idempotent with this op's own mask.
                o = torch.relu(o)
            res.append(o)
        return res

    # ------------------------------------------------------------------ forward
    def _forward_plan(self, N, H, W, Cp_in, precision, training, collect, det, wplanes=False):
        key = (N, H, W, Cp_in, precision, training, collect, det, wplanes, tile_kernels())
        plan = self._fwd_plans.get(key)
        if plan is None:
            plan = self._fwd_plans[key] = self._compile_forward(N, H, W, Cp_in, precision, training, collect, det, wplanes)
            while len(self._fwd_plans) > PLAN_CACHE:
                _, old = self._fwd_plans.popitem(last=False)
                for k in [k for k in self._bwd_plans if k[0] == id(old)]:    # its backward plans go with it
                    del self._bwd_plans[k]
        else:
            self._fwd_plans.move_to_end(key)
        return plan

    def _compile_forward(self, N, H, W, Cp_in, precision, training, collect, det, wplanes=False):
        """collect: the norm layers' running statistics are NOT updated; the batch mean / unbiased variance of every norm
        layer are left in arena buffers instead, for update_running() to apply (dual passes).
        wplanes: a backward pass with weight gradients may follow -- every tensor a tiled weight-gradient launch will read
        is also stored as bf16 operand planes (by the op that produces it)."""
        b = pg.Builder()
        packs = []
        ts = [_PT(b.ext('x'), self.in_channels, Cp_in, H, W)]
        ws = _Workspace(b, 'T') if det else None
        running = []    # (norm module, slot of the batch mean, slot of the batch variance)
        # ---- which tensors are needed as operand planes: shapes first (a dry walk over the stages)
        shapes = [(H, W, Cp_in)]
        ftile, need_pl = [], set()
        for st in self.stages:
            IH, IW, Cip = shapes[st.src]
            Cop = cp.cpad_pow2(st.cout)
            if st.kind == 'conv':
                launches, (OH, OW) = cp.conv_fwd(st.k, st.s, st.p, IH, IW)
                wq = (OH, OW, Cop, IH, IW, Cip)
            else:
                launches, (OH, OW) = cp.convT_fwd(st.k, st.s, st.p, st.op, IH, IW)
                wq = (IH, IW, Cip, OH, OW, Cop)
            shapes.append((OH, OW, Cop))
            narrow = (st.kind == 'conv' and st.s == 1 and st.cout <= 8 and st.norm is None and st.k in NARROW_KW
                      and precision == 3 and (st.cin <= 128 or N * OH * OW >= 16384))
            narrow = narrow or _head16_fwd_ok(st, Cip, Cop, precision, det)
            ft = (not narrow) and _tile_fwd_ok(st, launches, N, Cip, Cop, precision)
            if (not narrow) and st.kind == 'conv' and _halo_fwd_ok(st, launches, N, OH, OW, Cip, Cop, precision):
                ft = 'halo'
            ftile.append(ft)
            if ft:
                need_pl.add(st.src)
            if wplanes and _tile_wgrad_ok(st, N, wq[0], wq[1], wq[2], wq[3], wq[4], wq[5], precision, det):
                need_pl.add(st.src)
        if 0 in need_pl:
            X0 = ts[0]
            n0 = N * H * W * Cp_in
            X0.pls = plane_stride(n0)
            X0.pl = b.alloc('F', 4 * X0.pls)
            b.op(pg.OP_SPLIT_PLANES, buf=[X0.slot, X0.pl], l=[n0, X0.pls], i=[0], desc=('split', 'chain input'))
        for si, st in enumerate(self.stages):
            X = ts[st.src]
            IH, IW, Cip = X.H, X.W, X.Cp
            Cop = cp.cpad_pow2(st.cout)
            pad_mode = 1 if st.reflect else 0
            if st.kind == 'conv':
                launches, (OH, OW) = cp.conv_fwd(st.k, st.s, st.p, IH, IW)
            else:
                launches, (OH, OW) = cp.convT_fwd(st.k, st.s, st.p, st.op, IH, IW)
            z = b.alloc('F', 4 * N * OH * OW * Cop)
            stats = b.alloc('F', 8 * N * STAT_SLOTS * Cop * 2, zero=True) if st.norm is not None else None
            bias = _bias_slot(b, packs, st.padded_bias(Cop))
            if st.norm is not None or st.act == 'relu':
                epi_act = 0
            else:
                epi_act = ACT[st.act]
            desc = '%s k%d s%d %d->%d @%dx%d' % (st.kind, st.k, st.s, st.cin, st.cout, OH, OW)
            flops = 2.0 * N * OH * OW * st.k * st.k * st.cin * st.cout / (st.s * st.s if st.kind == 'convT' else 1)
            narrow = (st.kind == 'conv' and st.s == 1 and st.cout <= 8 and st.norm is None and st.k in NARROW_KW
                      and precision == 3
                      # the discriminator heads (512 -> 1, 4x4) at the coarse scales have too few positions to fill the chip
                      # with the narrow kernel's position tiles (0.15 ms whatever the size; MFMA path 0.04-0.07 ms)
                      and (st.cin <= 128 or N * OH * OW >= 16384))
            head16 = (not narrow) and _head16_fwd_ok(st, Cip, Cop, precision, det)
            if narrow or head16:  # head layers: exact fp32 on the vector ALUs (conv_narrow.hip), or the 16-row MFMA (conv_head.hip)
                L = launches[0]
                e = st.narrow('fwd', L.taps, L.tapidx, Cip)
                KH, KW, dy_min, dx_min, R = e.meta
                if head16 or _head_mfma_ok(KH, KW, Cip, Cop, precision):
                    e = st.head_mfma('fwd', L.taps, L.tapidx, Cip)
                    packs.append(e)
                    b.op(pg.OP_CONV_HEAD_MFMA, buf=[X.slot, z, b.static(e.buf), bias, stats],
                         i=[N, IH, IW, Cip, OH, OW, Cop, R, KH, KW, dy_min, dx_min, pad_mode, int(X.relu), epi_act],
                         desc=('fwd', desc + ' head mfma'), flops=flops)
                else:
                    packs.append(e)
                    b.op(pg.OP_CONV_NARROW_FWD, buf=[X.slot, z, b.static(e.buf), bias],
                         i=[N, IH, IW, Cip, OH, OW, Cop, R, KH, KW, dy_min, dx_min, pad_mode, int(X.relu), epi_act],
                         desc=('fwd', desc + ' narrow'), flops=flops)
            elif ftile[si] == 'halo':
                L = launches[0]
                e = st.packed_kmajor('fwd', L.tapidx, Cip, Cop)
                packs.append(e)
                b.op(pg.OP_CONV_HALO, buf=[X.pl, z, b.static(e.buf), bias, stats],
                     i=[N, IH, IW, Cip, OH, OW, Cop, len(L.taps), pad_mode, e.meta[0], epi_act, 0], l=[X.pls], taps=L.taps,
                     desc=('fwd', desc + ' halo'), flops=flops)
            elif ftile[si]:
                for li, L in enumerate(launches):
                    _emit_tile(b, packs, st, 'fwd', X, N, IH, IW, Cip, z, OH, OW, Cop, L, pad_mode, bias, epi_act, stats, False,
                               desc=('fwd', desc + ' tile'), flops=flops / len(launches))
            elif _phases_ok(launches, precision, N, Cop, Cip):
                _emit_gemm_phases(b, packs, st, 'fwd', X.slot, N, IH, IW, Cip, z, OH, OW, Cop, launches, pad_mode, X.relu,
                                  precision, bias, epi_act, stats, False, desc=('fwd', desc + ' phases'), flops=flops)
            else:
                for li, L in enumerate(launches):
                    _emit_gemm(b, packs, st, 'fwd', X.slot, N, IH, IW, Cip, z, OH, OW, Cop, L, pad_mode, X.relu, precision,
                               bias, epi_act, stats, False, ws, desc=('fwd', desc), flops=flops / len(launches))
            T = _PT(z, st.cout, Cop, OH, OW)
            want_pl = (si + 1) in need_pl
            if want_pl:
                T.pls = plane_stride(N * OH * OW * Cop)
                T.pl = b.alloc('F', 4 * T.pls)
            if st.norm is not None:
                nm = st.norm
                rm = rv = None
                momentum = float(nm.momentum if nm.momentum is not None else 0.1)
                if training and nm.track_running_stats and nm.running_mean is not None:
                    # torch's InstanceNorm updates the running mean / variance but leaves num_batches_tracked at 0
                    if collect:
                        # (1 - 1) * 0 + 1 * b: the kernel's update leaves the batch statistics themselves in the buffers
                        rm, rv = b.alloc('F', 4 * st.cout, zero=True), b.alloc('F', 4 * st.cout, zero=True)
                        running.append((nm, rm, rv, st.cout))
                        momentum = 1.0
                    else:
                        rm, rv = b.static(lambda nm=nm: nm.running_mean), b.static(lambda nm=nm: nm.running_var)
                out2 = res = None
                res_relu = False
                if st.res is not None:
                    R = ts[st.res]
                    res, res_relu = R.slot, R.relu
                    if (R.H, R.W, R.Cp) != (OH, OW, Cop):
                        raise ValueError('residual %s does not match the block output %s (channel padding: the block '
                                         'input must have a power-of-two channel count)' % ((N, R.H, R.W, R.Cp), (N, OH, OW, Cop)))
                    out2 = b.alloc('F', 4 * N * OH * OW * Cop)
                mr = b.alloc('F', 4 * N * Cop * 2)
                b.op(pg.OP_IN_APPLY, buf=[z, stats, mr, res, out2, rm, rv, T.pl],
                     i=[N, OH * OW, st.cout, Cop, 1 if st.act == 'lrelu' else 0, int(res_relu),
                        int(st.act == 'relu' and st.res is None)],
                     f=[float(nm.eps), momentum], l=[T.pls], desc=('in_apply', desc))
                T.mr = mr  # (mean, rstd) per (n, c): what the backward pass needs
                if st.res is not None:
                    if st.act != 'none':
                        raise NotImplementedError('activation after a residual add')
                    T.xhat = z
                    T.slot = out2
                    T.mode = 0
                else:
                    T.relu = st.act == 'relu'
                    T.mode = 1 if st.act == 'relu' else (2 if st.act == 'lrelu' else 0)
                    if st.act == 'tanh':
                        raise NotImplementedError('tanh after a norm')
            else:
                if st.res is not None:
                    raise NotImplementedError('residual without a norm')
                T.relu = st.act == 'relu'
                if want_pl:
                    b.op(pg.OP_SPLIT_PLANES, buf=[z, T.pl], l=[N * OH * OW * Cop, T.pls], i=[int(T.relu)],
                         desc=('split', desc))
            ts.append(T)
        plan = _Plan(b, packs)    # (finish() put one memset of the arena's zero-initialised pieces -- the statistics -- first)
        plan.ts, plan.running, plan.shape = ts, running, (N, H, W, Cp_in)
        return plan

    def _run_forward(self, x, precision, training, collect, wplanes=False):
        """x: channels-last padded input [N, H, W, Cp].  -> _State"""
        N, H, W, Cp_in = x.shape
        plan = self._forward_plan(N, H, W, Cp_in, precision, training, collect, deterministic(), wplanes)
        plan.refresh_packs()
        arenas = plan.program.new_arenas(x.device)
        _run(plan.program, arenas, {'x': x})
        return _State(self, plan, arenas['F'], x, precision)     # ('T', the ordered-sum scratch, is dropped here)

    def forward(self, x, precision, training=True, collect_running=None):
        """Inspection API (tests, diagnostics): runs the pass and returns (ts, geo) -- ts[i] a _T over the arena buffers
        of tensor i, geo[i] its (H, W).  collect_running: a list that receives (norm, batch mean, batch variance)."""
        state = self._run_forward(x, precision, training, collect_running is not None)
        if collect_running is not None:
            collect_running += state.running
        return state.tensors(), [(t.H, t.W) for t in state.plan.ts]

    # ------------------------------------------------------------------ backward
    def _backward_plan(self, fplan, gkeys, precision, need_input_grad, need_weight_grads, in_range, det, side):
        key = (id(fplan), gkeys, precision, need_input_grad, need_weight_grads, in_range, det, side)
        hit = self._bwd_plans.get(key)
        if hit is not None and hit[1]() is not fplan:
            hit = None        # the id belonged to a forward plan that has been evicted and freed since
        if hit is None:
            plan = self._compile_backward(fplan, gkeys, precision, need_input_grad, need_weight_grads, in_range, det, side)
            self._bwd_plans[key] = hit = (plan, weakref.ref(fplan))
            while len(self._bwd_plans) > 4 * PLAN_CACHE:
                self._bwd_plans.popitem(last=False)
        self._bwd_plans.move_to_end(key)
        return hit[0]

    def _compile_backward(self, fplan, gkeys, precision, need_input_grad, need_weight_grads, in_range, det, side):
        """gkeys: tuple of tensor indices that receive a gradient from outside (channels-last, padded, contiguous; external
        slots 'g<index>').  Arena 'F' is the forward pass's (external base pointer 'F'), 'S' scratch, 'P' the parameter
        gradients handed to autograd."""
        b = pg.Builder()
        packs = []
        ts = fplan.ts
        N = fplan.shape[0]
        fprog = fplan.program

        def fslot(slot):
            """a slot of the forward plan, re-expressed in this program"""
            if slot is None:
                return None
            kind = fprog_kinds[slot]
            if kind[0] == 'ext':
                return b.ext(kind[1])
            key = ('F', slot)
            s = fwd_slots.get(key)
            if s is None:
                s = fwd_slots[key] = b.ext('F+%d' % fprog.offset(slot))
            return s
        fprog_kinds = fplan.kinds
        fwd_slots = {}
        G = {}
        gin_cp = ts[0].Cp
        for ti in gkeys:
            T = ts[ti]
            buf = b.alloc('S', 4 * N * T.H * T.W * T.Cp)
            b.op(pg.OP_COPY, buf=[buf, b.ext('g%d' % ti)], l=[4 * N * T.H * T.W * T.Cp])  # the kernels work in place
            G[ti] = buf
        pgrads = [None] * (2 * len(self.stages))      # (slot, shape) per parameter
        ws_main = _Workspace(b, 'S') if det else None
        ws_side = (_Workspace(b, 'S') if side else ws_main) if det else None
        sd = 1 if side else 0
        late_unpacks = []
        for si in range(len(self.stages) - 1, -1, -1):
            st = self.stages[si]
            T = ts[si + 1]
            g = G.pop(si + 1, None)
            if g is None:
                continue
            X = ts[st.src]
            IH, IW, Cip = X.H, X.W, X.Cp
            OH, OW, Cop = T.H, T.W, T.Cp
            nb = 4 * N * OH * OW * Cop
            if st.res is not None:  # d(res) = g, before g is overwritten by dz
                if st.res in G:
                    b.op(pg.OP_ADD, buf=[G[st.res], G[st.res], g], l=[nb // 4])
                else:
                    G[st.res] = b.alloc('S', nb)
                    b.op(pg.OP_COPY, buf=[G[st.res], g], l=[nb])
            bgrad = None
            has_b = st.conv.bias is not None
            # tiled weight gradient: dz is also written as operand planes by the op that produces it
            if st.kind == 'conv':
                wq = (OH, OW, Cop, IH, IW, Cip)
            else:
                wq = (IH, IW, Cip, OH, OW, Cop)
            # 7 x 7 layers with <= 16 channels on the d(out) side: the generator head, the encoder's head and stem (conv_whead.hip)
            whead = (need_weight_grads and _head_wgrad_ok(st, Cip, Cop, precision, det))
            wtile = (need_weight_grads and X.pl is not None and not whead
                     and _tile_wgrad_ok(st, N, wq[0], wq[1], wq[2], wq[3], wq[4], wq[5], precision, det))
            # tiled data gradient (stride-1 layers): reads dz as planes too
            dsplit = None
            if not (st.src == 0 and not need_input_grad) and st.kind == 'conv' and st.s == 1 and not (
                    st.src == 0 and in_range is not None and in_range != (0, st.cin)):
                dl, (dgh, dgw) = cp.conv_dgrad(st.k, st.s, st.p, IH, IW, bool(st.reflect))
                dsplit = _tile_dgrad_plan(st, dl, N, dgh, dgw, Cop, Cip, precision, det,
                                          acc=(not st.reflect and st.src in G))
            dz_pl, dz_pls = None, 0
            if wtile or dsplit is not None:
                dz_pls = plane_stride(N * OH * OW * Cop)
                dz_pl = b.alloc('S', 4 * dz_pls)
            if st.norm is not None:
                stored = T.xhat if T.xhat is not None else T.slot
                sums = b.alloc('S', 8 * N * Cop * 2, zero=True)   # cleared by the arena's one memset (SDN_IN_BWD_SUMS_ZEROED)
                b.op(pg.OP_IN_BWD, buf=[g, fslot(stored), fslot(T.mr), sums, dz_pl], i=[N, OH * OW, Cop, T.mode | 8], l=[dz_pls],
                     desc=('in_bwd', '%d ch @%dx%d' % (st.cout, OH, OW)))
                if has_b and need_weight_grads:
                    bgrad = b.alloc('P', 4 * st.cout, zero=True)  # a bias in front of InstanceNorm has zero gradient
            else:
                ordered = has_b and det   # the kernel's bias sum meets in float atomics
                bg = b.alloc('P', 4 * Cop, zero=True) if has_b and not ordered else None
                b.op(pg.OP_ACT_BWD, buf=[g, fslot(T.slot), bg, dz_pl], l=[N * OH * OW, dz_pls], i=[Cop, ACT[st.act]])
                if ordered:
                    bgrad = b.alloc('P', 4 * st.cout)
                    b.op(pg.OP_COLSUM, buf=[g, bgrad], l=[N * OH * OW], i=[Cop, st.cout])
                elif has_b:
                    bgrad = bg                                    # the first cout entries are the gradient
            dz = g
            # ---- weight gradient
            pad_mode = 1 if st.reflect else 0
            if need_weight_grads:
                if side:
                    b.op(pg.OP_FORK)              # dz is final on the calling stream
                if st.kind == 'conv':
                    WL = cp.conv_wgrad(st.k, st.s, st.p, OH, OW)
                    rows_t, gath_t, Cr, Cc, GH, GW = dz, fslot(X.slot), Cop, Cip, IH, IW
                    relu_rows, relu_gath, wpad = False, X.relu, pad_mode
                    R_, C_, (sr, sc) = st.cout, st.cin, st.str_fwd
                else:
                    WL = cp.convT_wgrad(st.k, st.s, st.p, IH, IW)
                    rows_t, gath_t, Cr, Cc, GH, GW = fslot(X.slot), dz, Cip, Cop, OH, OW
                    relu_rows, relu_gath, wpad = X.relu, False, 0
                    R_, C_, (sr, sc) = st.cin, st.cout, st.str_dgrad
                ntaps = len(WL.taps)
                dwp = b.alloc('S', 4 * Cr * ntaps * Cc, zero=True)   # the K slices of the weight gradient meet in it
                n_tiles = ((Cr + 127) // 128 if Cr > 64 else 1) * ((ntaps * Cc + 127) // 128)
                splits = cp.wgrad_splits(N * WL.QH * WL.QW, n_tiles)
                desc = '%s k%d s%d %d->%d @%dx%d' % (st.kind, st.k, st.s, st.cin, st.cout, OH, OW)
                flops = 2.0 * N * OH * OW * st.k * st.k * st.cin * st.cout / (st.s * st.s if st.kind == 'convT' else 1)
                if wtile:
                    if st.kind == 'conv':
                        rp, rps, gp, gps = dz_pl, dz_pls, fslot(X.pl), X.pls
                    else:
                        rp, rps, gp, gps = fslot(X.pl), X.pls, dz_pl, dz_pls
                    b.op(pg.OP_CONV_WGRAD_TILE, buf=[rp, gp, dwp], i=[N, WL.QH, WL.QW, Cr, GH, GW, Cc, WL.istride, ntaps, wpad],
                         l=[rps, gps], taps=WL.taps, stream=sd, desc=('wgrad', desc + ' tile'), flops=flops)
                elif whead:
                    b.op(pg.OP_CONV_WGRAD_HEAD, buf=[rows_t, gath_t, dwp],
                         i=[N, WL.QH, WL.QW, Cr, st.cout, GH, GW, Cc, ntaps, wpad, int(relu_rows), int(relu_gath)],
                         taps=WL.taps, stream=sd, desc=('wgrad', desc + ' head mfma'), flops=flops)
                elif st.kind == 'conv' and st.s == 1 and st.cout <= 8 and not det:
                    # head layers (1-5 output channels): exact fp32 on the vector ALUs, input tile + halo kept in LDS
                    # (its blocks meet in dw through float atomics: the deterministic mode takes the MFMA kernel below)
                    b.op(pg.OP_CONV_WGRAD_NARROW, buf=[rows_t, gath_t, dwp],
                         i=[N, WL.QH, WL.QW, Cr, st.cout, GH, GW, Cc, ntaps, wpad, int(relu_rows), int(relu_gath)],
                         taps=WL.taps, stream=sd, desc=('wgrad', desc + ' narrow'), flops=flops)
                else:
                    wsn = 0
                    if det and splits > 1:
                        wsn = ws_side.need(splits * Cr * ntaps * Cc * 4)
                    b.op(pg.OP_CONV_WGRAD, buf=[rows_t, gath_t, dwp, ws_side.slot if wsn else None],
                         i=[N, WL.QH, WL.QW, Cr, GH, GW, Cc, WL.istride, ntaps, wpad, int(relu_rows), int(relu_gath), splits,
                            precision], l=[wsn], taps=WL.taps, stream=sd, desc=('wgrad', desc + ' splits %d' % splits),
                         flops=flops)
                # the gradient in the parameter's layout: the plan's taps cover the whole window, so every element is
                # written exactly once (plain stores, no zero fill)
                assert ntaps == st.k * st.k
                wshape = tuple(st.conv.weight.shape)
                wgrad = b.alloc('P', 4 * st.conv.weight.numel())
                tix = st.tix(WL.tapidx, st.conv.weight.device)
                # (emitted behind the last stage: a run of unpack records is ONE launch, see k_weights_multi)
                late_unpacks.append(dict(buf=[dwp, b.static(tix), wgrad], i=[R_, C_, ntaps, Cc, 0], l=[sr, sc], stream=sd))
                pgrads[2 * si] = (wgrad, wshape)
                if bgrad is not None:
                    pgrads[2 * si + 1] = (bgrad, (st.cout,))
            # ---- data gradient
            if st.src == 0 and not need_input_grad:
                continue
            have = st.src in G
            if st.kind == 'conv':
                launches, (GHt, GWt) = cp.conv_dgrad(st.k, st.s, st.p, IH, IW, bool(st.reflect))
            else:
                launches, (GHt, GWt) = cp.convT_dgrad(st.k, st.s, st.p, IH, IW)
            rr, Cg = None, Cip
            if st.src == 0 and in_range is not None and in_range != (0, st.cin):
                rr, Cg = in_range, cp.cpad(in_range[1] - in_range[0])  # gradient for these input channels only
            if st.src == 0:
                gin_cp = Cg
            if st.reflect:
                target = b.alloc('S', 4 * N * GHt * GWt * Cg)
                acc = False
            elif have:
                target, acc = G[st.src], True
            else:
                target = b.alloc('S', 4 * N * IH * IW * Cg)
                acc = False
            desc = '%s k%d s%d %d->%d @%dx%d' % (st.kind, st.k, st.s, st.cin, st.cout, OH, OW)
            flops = 2.0 * N * OH * OW * st.k * st.k * st.cin * st.cout / (st.s * st.s if st.kind == 'convT' else 1)
            if rr is not None:
                desc += ' ch %d:%d' % rr
                flops *= (rr[1] - rr[0]) / float(st.cin)
            narrow = (rr is not None and rr[1] - rr[0] <= 8 and st.kind == 'conv' and st.s == 1 and not acc
                      and st.k in NARROW_KW and precision == 3)
            # r06: the data gradient of a 7 x 7 head towards ALL its input channels, when dz has 16 (padded) channels and the
            # gradient 16 or 64 -- the encoder head (16 -> 5) and the generator head (64 -> 3): sdn_conv_head_mfma with dz as its
            # input; was sdn_conv_gemm over K = 49 x 16 with 3-5 real channels per tap (0.58 / 0.69 ms at 384 x 1248, batch 4)
            head_d = (_head_wide() and (not narrow) and st.kind == 'conv' and st.s == 1 and st.k == 7 and not acc and len(launches) == 1
                      and (rr is None or rr[1] - rr[0] > 8) and _head_mfma_ok(7, 7, Cop, Cg, precision))
            if narrow or head_d:
                L = launches[0]
                e = st.narrow('dgrad', L.taps, L.tapidx, Cop, rr)
                KH, KW, dy_min, dx_min, R = e.meta
                if head_d or _head_mfma_ok(KH, KW, Cop, Cg, precision):
                    e = st.head_mfma('dgrad', L.taps, L.tapidx, Cop, rr)
                    packs.append(e)
                    b.op(pg.OP_CONV_HEAD_MFMA, buf=[dz, target, b.static(e.buf), None, None],
                         i=[N, OH, OW, Cop, GHt, GWt, Cg, R, KH, KW, dy_min, dx_min, 0, 0, 0],
                         desc=('dgrad', desc + ' head mfma'), flops=flops)
                else:
                    packs.append(e)
                    b.op(pg.OP_CONV_NARROW_FWD, buf=[dz, target, b.static(e.buf), None],
                         i=[N, OH, OW, Cop, GHt, GWt, Cg, R, KH, KW, dy_min, dx_min, 0, 0, 0],
                         desc=('dgrad', desc + ' narrow'), flops=flops)
            elif dsplit is not None:
                _emit_tile(b, packs, st, 'dgrad', (dz_pl, dz_pls), N, OH, OW, Cop, target, GHt, GWt, Cg, launches[0], 0, None, 0,
                           None, acc, desc=('dgrad', desc + (' tile ksplit %d' % dsplit if dsplit else ' tile')), flops=flops,
                           ksplit=dsplit)
            else:
                if not acc and any(not L.taps for L in launches):
                    # phases no kernel tap reaches (a 1x1 stride-2 conv reads every other pixel only)
                    b.op(pg.OP_MEMSET, buf=[target], l=[4 * N * GHt * GWt * Cg])
                live = [L for L in launches if L.taps]
                if _phases_ok(live, precision, N, Cg, Cop):
                    _emit_gemm_phases(b, packs, st, 'dgrad', dz, N, OH, OW, Cop, target, GHt, GWt, Cg, live, 0, False, precision,
                                      None, 0, None, acc, rows_range=rr, desc=('dgrad', desc + ' phases'), flops=flops)
                else:
                    for L in live:
                        _emit_gemm(b, packs, st, 'dgrad', dz, N, OH, OW, Cop, target, GHt, GWt, Cg, L, 0, False, precision, None,
                                   0, None, acc, ws_main, rows_range=rr, desc=('dgrad', desc), flops=flops / len(live))
            if st.reflect:
                if have:
                    out = G[st.src]
                else:
                    out = b.alloc('S', 4 * N * IH * IW * Cg)
                b.op(pg.OP_REFLECT_FOLD, buf=[target, out], i=[N, IH, IW, Cg, st.reflect, int(have)])
                G[st.src] = out
            else:
                G[st.src] = target
        for rec in late_unpacks:
            b.op(pg.OP_UNPACK_GRAD, **rec)
        if side and need_weight_grads:
            b.op(pg.OP_JOIN)
        gin = G.get(0)
        if not b.ops:
            return None
        plan = _Plan(b, packs)
        plan.pgrads, plan.gin = pgrads, gin
        plan.gin_shape = (N, ts[0].H, ts[0].W, gin_cp) if gin is not None else None
        plan.fwd_ext = sorted(set(k for k in plan.program._ext if k.startswith('F+')))
        plan.fwd_offsets = [int(k[2:]) for k in plan.fwd_ext]
        return plan

    def backward(self, state, gouts, need_input_grad, need_weight_grads=True, in_range=None):
        """state: the forward pass's _State; gouts: {tensor index: grad buffer (channels-last, padded)}.
        Returns (grad_input or None, [grad per param])."""
        dev = state.x.device
        gkeys = tuple(sorted(gouts))
        side = _wgrad_stream(dev) if need_weight_grads and len(self.stages) > 1 else None
        # (one-conv chains -- the ResNet-18 encoder's layers -- are issue-bound on the host: the extra stream bookkeeping
        # cost their train step 9.0 -> 10.8 ms, tools/encoder_streams_lab.py)
        plan = self._backward_plan(state.plan, gkeys, state.precision, bool(need_input_grad), bool(need_weight_grads),
                                   in_range, deterministic(), side is not None)
        pgrads = [None] * (2 * len(self.stages))
        if plan is None:
            return None, pgrads
        plan.refresh_packs()
        arenas = plan.program.new_arenas(dev)
        ext = {'x': state.x}
        base = state.arena.data_ptr()
        for k, off in zip(plan.fwd_ext, plan.fwd_offsets):
            ext[k] = base + off
        for ti in gkeys:
            g = gouts[ti]
            ext['g%d' % ti] = g if g.is_contiguous() else g.contiguous()
        # (the side stream works on memory allocated on the calling stream; the program joins the streams at its end, so
        # whatever the caching allocator hands out again afterwards is ordered behind the side stream's work)
        _run(plan.program, arenas, ext, side)
        prog = plan.program
        for k, e in enumerate(plan.pgrads):
            if e is not None:
                pgrads[k] = prog.view(arenas, e[0], e[1])
        gin = prog.view(arenas, plan.gin, plan.gin_shape) if plan.gin is not None else None
        return gin, pgrads


class _State:
    """What a forward pass leaves behind: the plan, the activation arena, the input buffer."""
    __slots__ = ('chain', 'plan', 'arena', 'x', 'precision', '_running')

    def __init__(self, chain, plan, arena, x, precision):
        self.chain, self.plan, self.arena, self.x, self.precision = chain, plan, arena, x, precision
        self._running = None

    def view(self, slot, shape, dtype=torch.float32):
        kind = self.plan.kinds[slot]
        if kind[0] == 'ext':
            return self.x
        return self.plan.program.view({'F': self.arena}, slot, shape, dtype)

    def output(self, ti):
        T = self.plan.ts[ti]
        return self.view(T.slot, (self.plan.shape[0], T.H, T.W, T.Cp))

    @property
    def running(self):
        """(norm, batch mean, batch unbiased variance) per norm layer of a collecting pass"""
        if self._running is None:
            self._running = [(nm, self.view(rm, (c,)), self.view(rv, (c,))) for nm, rm, rv, c in self.plan.running]
        return self._running

    def tensors(self):
        N = self.plan.shape[0]
        out = []
        for T in self.plan.ts:
            t = _T(self.view(T.slot, (N, T.H, T.W, T.Cp)), T.C, T.relu)
            if T.xhat is not None:
                t.xhat = self.view(T.xhat, (N, T.H, T.W, T.Cp))
            if T.mr is not None:
                t.stats = self.view(T.mr, (N, T.Cp, 2))
            t.mode = T.mode
            out.append(t)
        return out


def _input_buffer(parts):
    """channels-last input buffer, the parts side by side (what torch.cat + permute + pad would build)"""
    N, _, H, W = parts[0].shape
    C = sum(int(t.shape[1]) for t in parts)
    Cp = cp.cpad(C)
    if len(parts) == 1 and Cp == C:
        return parts[0].permute(0, 2, 3, 1).contiguous()
    if (parts[0].is_cuda and len(parts) <= 8 and Cp <= 128
            and all(t.dtype == torch.float32 and t.is_contiguous() and t.shape[0] == N and t.shape[2:] == (H, W) for t in parts)):
        # one launch: every part read once along W, the buffer written once along C, pad channels zero (sdn_assemble_nhwc)
        x = torch.empty(N, H, W, Cp, dtype=torch.float32, device=parts[0].device)
        ptrs = (ctypes.c_void_p * len(parts))(*[t.data_ptr() for t in parts])
        chans = (ctypes.c_int32 * len(parts))(*[int(t.shape[1]) for t in parts])
        check(lib().sdn_assemble_nhwc(ptrs, chans, len(parts), N, H, W, Cp, x.data_ptr(), stream()))
        return x
    x = (torch.zeros if Cp != C else torch.empty)(N, H, W, Cp, dtype=torch.float32, device=parts[0].device)
    c0 = 0
    for t in parts:
        x[..., c0:c0 + t.shape[1]] = t.permute(0, 2, 3, 1)
        c0 += int(t.shape[1])
    return x


class _ChainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, chain, nparts, *args):
        parts = args[:nparts]
        precision = default_precision()
        norms = [st.norm for st in chain.stages if st.norm is not None]
        training = any(nm.training for nm in norms)
        if norms and not training and any(nm.track_running_stats for nm in norms):
            # torch would normalise with the running statistics here; the reference never calls .eval() on the textural
            # networks (no such call under textural/), so that mode is deliberately not implemented
            raise NotImplementedError('InstanceNorm2d(track_running_stats=True) in eval mode')
        with torch.no_grad():
            x = _input_buffer(parts)
            keep = chain.__dict__.get('_keep_state')   # ConvChain.__call__(dual=True): a second autograd view follows
            wplanes = any(ctx.needs_input_grad[2 + nparts:])   # a weight-gradient pass may follow
            state = chain._run_forward(x, precision, training, bool(keep), wplanes)
        ctx.chain, ctx.state = chain, state
        ctx.nparts, ctx.part_channels = nparts, [int(t.shape[1]) for t in parts]
        if keep:
            chain.__dict__['_state'] = state
        # outputs that take no gradient arrive as None in backward, not as zero tensors (the discriminator's own loss reads
        # only the last of a column's five feature maps: autograd used to fill the other four with zeros, and the backward
        # pass copied and added them -- ~100 MB each at the finest scale)
        ctx.set_materialize_grads(False)
        outs = tuple(state.output(i) for i in chain.outputs)
        return outs if len(outs) > 1 else outs[0]

    @staticmethod
    def backward(ctx, *gouts):
        nparts = ctx.nparts
        gparts, pgr = _chain_backward(ctx, gouts, ctx.needs_input_grad[2:2 + nparts], any(ctx.needs_input_grad[2 + nparts:]))
        return (None, None) + tuple(gparts) + tuple(pgr)


def _chain_backward(ctx, gouts, need_parts, need_w):
    chain = ctx.chain
    g = {}
    for ti, go in zip(chain.outputs, gouts):
        if go is not None:
            g[ti] = go
    nparts = ctx.nparts
    # channel range covering every part that wants a gradient
    starts = [sum(ctx.part_channels[:i]) for i in range(nparts)]
    lo = min([starts[i] for i in range(nparts) if need_parts[i]], default=0)
    hi = max([starts[i] + ctx.part_channels[i] for i in range(nparts) if need_parts[i]], default=0)
    with torch.no_grad():
        gin, pgr = chain.backward(ctx.state, g, any(need_parts), need_w, in_range=(lo, hi) if any(need_parts) else None)
    ctx.state = None
    gparts = []
    for i in range(nparts):
        if need_parts[i] and gin is not None:
            a = starts[i] - lo
            gparts.append(gin[..., a:a + ctx.part_channels[i]].permute(0, 3, 1, 2))
        else:
            gparts.append(None)
    return gparts, pgr


class _ChainSharedFn(torch.autograd.Function):
    """A second autograd view of a forward pass _ChainFn has already run (ConvChain.__call__(dual=True)): the outputs
    alias the stored activations, the backward pass runs the data-gradient kernels only (the stored activations are
    read-only in every backward kernel, so the two views can be back-propagated in either order)."""

    @staticmethod
    def forward(ctx, chain, state, nparts, *parts):
        ctx.chain, ctx.state = chain, state
        ctx.nparts, ctx.part_channels = nparts, [int(t.shape[1]) for t in parts]
        ctx.set_materialize_grads(False)
        outs = tuple(state.output(i).detach() for i in chain.outputs)   # new tensor objects on the same storage
        return outs if len(outs) > 1 else outs[0]

    @staticmethod
    def backward(ctx, *gouts):
        gparts, _ = _chain_backward(ctx, gouts, ctx.needs_input_grad[3:3 + ctx.nparts], False)
        return (None, None, None) + tuple(gparts)


# ---------------------------------------------------------------------------------------------------------------------
def compile_sequential(modules, stages=None, src=0, base=0):
    """Turn a list of torch modules (the reference's nn.Sequential contents) into Stages.  Returns (stages, index of the
    last tensor).  Tensor indices are offset by `base` stages already in the list."""
    import torch.nn as nn
    if stages is None:
        stages = []
    cur = src
    pending_reflect = 0
    for m in modules:
        name = m.__class__.__name__
        if isinstance(m, nn.ReflectionPad2d):
            pending_reflect = int(m.padding[0])
        elif isinstance(m, nn.ConvTranspose2d):
            stages.append(Stage('convT', m, cur))
            cur = len(stages)
        elif isinstance(m, nn.Conv2d):
            stages.append(Stage('conv', m, cur, reflect=pending_reflect))
            pending_reflect = 0
            cur = len(stages)
        elif isinstance(m, nn.InstanceNorm2d):
            if m.affine:
                raise NotImplementedError('affine InstanceNorm2d')
            stages[-1].norm = m
        elif isinstance(m, nn.ReLU):
            stages[-1].act = 'relu'
        elif isinstance(m, nn.LeakyReLU):
            if abs(m.negative_slope - 0.2) > 1e-12:
                raise NotImplementedError('LeakyReLU slope %g' % m.negative_slope)
            stages[-1].act = 'lrelu'
        elif isinstance(m, nn.Tanh):
            stages[-1].act = 'tanh'
        elif name == 'ResnetBlock':
            block_in = cur
            _, cur = compile_sequential(list(m.conv_block), stages, src=cur)
            stages[-1].res = block_in
        elif isinstance(m, nn.Sequential):
            _, cur = compile_sequential(list(m), stages, src=cur)
        elif isinstance(m, nn.Dropout):
            raise NotImplementedError('dropout inside the fused conv chain')
        else:
            raise NotImplementedError('module %s in a fused conv chain' % name)
    return stages, cur
