"""Executor of conv + InstanceNorm + activation chains on the HIP kernels (sdn_conv_* / sdn_in_* in include/sdn_hip.h).

The textural networks of the reference (textural/models/networks.py) are nn.Sequential chains of
[ReflectionPad2d] Conv2d|ConvTranspose2d [InstanceNorm2d] [ReLU|LeakyReLU|Tanh] groups plus ResnetBlocks.  The product
modules in textural/models/networks.py keep those torch modules as PARAMETER CONTAINERS (identical state_dict keys and
initialisation) and hand the chain to this executor, which runs the whole forward and backward on channels-last fp32
buffers through the C ABI: one autograd.Function per chain, no torch convolution anywhere.  There is no CPU path: a
tensor that is not on the GPU raises.

Stored tensors and the deferred ReLU: a stage with a norm stores xhat = (z - mean) * rstd; if its activation is ReLU the
tensor is flagged `relu` and every consumer (next conv's loader, residual add, weight-gradient loader) applies
max(., 0) on the fly, so the backward pass still has xhat.  LeakyReLU (discriminator features, which are returned to
the caller) is materialised and inverted in the backward kernels.
"""
import contextlib
import ctypes
import os

import torch

from . import check, lib, ptr, stream
from . import convplan as cp

_i8 = ctypes.c_int8
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


_workspaces = {}


def _workspace(dev, nbytes):
    """One growing scratch buffer per device for the ordered split-K reduction (stream-ordered reuse: every launch that
    writes it is followed by its own reduce on the same stream)."""
    key = (dev.type, dev.index, torch.cuda.current_stream(dev).cuda_stream)
    t = _workspaces.get(key)
    if t is None or t.numel() < nbytes:
        t = _workspaces[key] = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
    return t


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


class _ZeroArena:
    """One zero-filled buffer per chain pass, carved into the many small zero-initialised tensors a pass needs
    (InstanceNorm statistics, weight-gradient accumulators, the gradients handed to autograd): one memset instead of
    several hundred fill launches per train step."""

    def __init__(self, dev, dtype):
        self.dev, self.dtype, self.want, self.buf, self.off = dev, dtype, 0, None, 0

    def reserve(self, numel):
        self.want += (int(numel) + 63) // 64 * 64     # keep every piece 256-byte aligned

    def materialize(self):
        """allocate + zero now (on the current stream) instead of at the first take"""
        if self.buf is None:
            self.buf = torch.zeros(max(self.want, 64), dtype=self.dtype, device=self.dev)
        return self.buf

    def take(self, shape):
        n = 1
        for d in shape:
            n *= int(d)
        if self.buf is None:
            self.buf = torch.zeros(max(self.want, 64), dtype=self.dtype, device=self.dev)
        if self.off + n > self.buf.numel():            # not reserved (shape-dependent path): fall back to its own fill
            return torch.zeros(shape, dtype=self.dtype, device=self.dev)
        t = self.buf[self.off:self.off + n].view(shape)
        self.off += (n + 63) // 64 * 64
        return t


PROFILE = None  # development aid: set to a list to collect (what, stage description, ms, flops) per kernel group


class _timed:
    """with _timed('fwd', stage, flops): ... -- records wall GPU time of the enclosed launches when PROFILE is a list."""

    def __init__(self, what, desc, flops=0.0):
        self.what, self.desc, self.flops = what, desc, flops

    def __enter__(self):
        if PROFILE is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *a):
        if PROFILE is not None:
            self.e1.record()
            self.e1.synchronize()
            PROFILE.append((self.what, self.desc, self.e0.elapsed_time(self.e1), self.flops))
        return False


def _taps_c(taps):
    n = len(taps)
    return (_i8 * n)(*[t[0] for t in taps]), (_i8 * n)(*[t[1] for t in taps])


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
        self._bias = None

    def tix(self, tapidx, dev):
        """device int32 copy of a tap-index list (cached: a fresh H2D copy per launch costs more than the kernel)"""
        key = (tuple(tapidx), str(dev))
        t = self._tix.get(key)
        if t is None:
            t = self._tix[key] = torch.tensor(list(tapidx), dtype=torch.int32, device=dev)
        return t

    def padded_bias(self, cop):
        b = self.conv.bias
        if b is None:
            return None
        if cop == self.cout:
            return b.detach()
        hit = self._bias
        if hit is None or hit[0] != _tag(b) or hit[1].shape[0] != cop:
            self._bias = hit = (_tag(b), torch.nn.functional.pad(b.detach(), (0, cop - self.cout)))
        return hit[1]

    # ---- packed weights, refreshed when the parameter changes (see _tag above)
    def packed(self, which, tapidx, precision, ccp, out_cp, rows_range=None):
        """which 'fwd': rows = cout, cols = cin;  'dgrad': rows = cin, cols = cout.  ccp: padded channel count of the
        tensor the gemm reads, out_cp: of the tensor it writes (selects the N tile, hence the row padding).
        rows_range (lo, hi): only these rows of the logical matrix (a data gradient for some input channels)."""
        w = self.conv.weight
        key = (which, tuple(tapidx), precision, ccp, out_cp, rows_range)
        hit = self._packed.get(key)
        if hit is not None and hit[0] == _tag(w):
            return hit[1]
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
        dev = w.device
        tix = self.tix(tapidx, dev)
        packed_w = torch.empty(2 * rows * Kp, dtype=torch.bfloat16, device=dev)  # fragment-major hi / lo blocks
        check(lib().sdn_conv_pack_weights(ctypes.c_void_p(w.data_ptr() + 4 * row0 * sr), R, C, sr, sc, ptr(tix), len(tapidx), ccp, Kp, rows,
                                          ptr(packed_w), stream()))
        val = (packed_w, Kp, rows)
        self._packed[key] = (_tag(w), val)
        return val

    def narrow(self, which, taps, tapidx, ccp, rows_range=None):
        """Dense fp32 tap window [KH, KW, ccp, RP] for sdn_conv_narrow_fwd (layers with <= 8 rows), cached like packed().
        which 'fwd': rows = cout, cols = cin;  'dgrad': rows = cin[rows_range], cols = cout."""
        w = self.conv.weight
        key = ('narrow', which, tuple(taps), ccp, rows_range)
        hit = self._packed.get(key)
        if hit is not None and hit[0] == _tag(w):
            return hit[1]
        kk = self.k * self.k
        m = w.detach().reshape(w.shape[0], w.shape[1], kk)        # Conv2d: [cout, cin, taps]
        a = m if which == 'fwd' else m.permute(1, 0, 2)            # [rows, cols, taps]
        if rows_range is not None:
            a = a[rows_range[0]:rows_range[1]]
        R, C = a.shape[0], a.shape[1]
        RP = 1 if R == 1 else (4 if R <= 4 else 8)
        dys, dxs = [t[0] for t in taps], [t[1] for t in taps]
        dy_min, dx_min = min(dys), min(dxs)
        KH, KW = max(dys) - dy_min + 1, max(dxs) - dx_min + 1
        dense = torch.zeros(KH, KW, ccp, RP, dtype=torch.float32, device=w.device)
        iy = torch.tensor([d - dy_min for d in dys], device=w.device)
        ix = torch.tensor([d - dx_min for d in dxs], device=w.device)
        it = torch.tensor(list(tapidx), device=w.device)
        dense[iy, ix, :C, :R] = a[:, :, it].permute(2, 1, 0)
        val = (dense, KH, KW, dy_min, dx_min, R)
        self._packed[key] = (_tag(w), val)
        return val


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


def _gemm(x, N, IH, IW, Cip, out, OH, OW, Cop, L, pad_mode, in_relu, packed, bias, act, stats, accumulate, precision):
    pw, Kp, rows = packed
    dy, dx = _taps_c(L.taps)
    ws, wsn = None, 0
    if deterministic():
        n = ctypes.c_size_t(0)
        check(lib().sdn_conv_gemm_workspace_bytes(N, OH, OW, Cop, ctypes.byref(n)))
        ws = _workspace(x.device, n.value)
        wsn = ws.numel()
    check(lib().sdn_conv_gemm(ptr(x), N, IH, IW, Cip, ptr(out), OH, OW, Cop, L.QH, L.QW, L.istride, L.ostride, L.py,
                              L.px, len(L.taps), dy, dx, pad_mode, int(in_relu), ptr(pw), Kp, rows, ptr(bias),
                              act, ptr(stats), int(accumulate), precision, ptr(ws), wsn, stream()))


NARROW_KW = (3, 4, 7)  # window sizes sdn_conv_narrow_fwd is built for


def _narrow_fwd(x, N, IH, IW, Cip, out, QH, QW, Cop, nw, pad_mode, in_relu, bias, act):
    dense, KH, KW, dy_min, dx_min, R = nw
    check(lib().sdn_conv_narrow_fwd(ptr(x), N, IH, IW, Cip, ptr(out), QH, QW, Cop, R, ptr(dense), KH, KW, dy_min, dx_min,
                                    pad_mode, int(in_relu), ptr(bias), act, stream()))


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


class ConvChain:
    """Runs a list of Stages.  tensors[0] is the chain input; tensors[i + 1] the output of stage i.
    `outputs`: indices (into tensors) returned to the caller, in order."""

    def __init__(self, stages, outputs, in_channels):
        self.stages = stages
        self.outputs = outputs
        self.in_channels = in_channels

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
            return self._present(outs_w), self._present(outs_x), state[3]
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
    def forward(self, x, precision, training=True, collect_running=None):
        """collect_running: a list -> the norm layers' running statistics are NOT updated; (norm, batch mean, batch
        unbiased variance) is appended per norm layer instead, for update_running() to apply (dual passes)."""
        N, H, W, _ = x.shape
        ts = [_T(x, self.in_channels)]
        geo = [(H, W)]
        stat_arena = _ZeroArena(x.device, torch.float64)
        for st in self.stages:
            if st.norm is not None:
                stat_arena.reserve(N * STAT_SLOTS * cp.cpad_pow2(st.cout) * 2)
        for st in self.stages:
            X = ts[st.src]
            IH, IW = geo[st.src]
            Cip = X.data.shape[3]
            Cop = cp.cpad_pow2(st.cout)
            pad_mode = 1 if st.reflect else 0
            if st.kind == 'conv':
                launches, (OH, OW) = cp.conv_fwd(st.k, st.s, st.p, IH, IW)
            else:
                launches, (OH, OW) = cp.convT_fwd(st.k, st.s, st.p, st.op, IH, IW)
            z = torch.empty(N, OH, OW, Cop, dtype=torch.float32, device=x.device)
            stats = stat_arena.take((N, STAT_SLOTS, Cop, 2)) if st.norm is not None else None
            bias = st.padded_bias(Cop)
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
            with _timed('fwd', desc + (' narrow' if narrow else ''), flops):
                if narrow:  # head layers: exact fp32 on the vector ALUs (conv_narrow.hip)
                    L = launches[0]
                    _narrow_fwd(X.data, N, IH, IW, Cip, z, OH, OW, Cop, st.narrow('fwd', L.taps, L.tapidx, Cip), pad_mode,
                                X.relu, bias, epi_act)
                else:
                    for L in launches:
                        _gemm(X.data, N, IH, IW, Cip, z, OH, OW, Cop, L, pad_mode, X.relu,
                              st.packed('fwd', L.tapidx, precision, Cip, Cop), bias, epi_act, stats, False, precision)
            T = _T(z, st.cout)
            if st.norm is not None:
                nm = st.norm
                rm = rv = None
                momentum = float(nm.momentum if nm.momentum is not None else 0.1)
                if training and nm.track_running_stats and nm.running_mean is not None:
                    # torch's InstanceNorm updates the running mean / variance but leaves num_batches_tracked at 0
                    rm, rv = nm.running_mean, nm.running_var
                    if collect_running is not None:
                        # (1 - 1) * 0 + 1 * b: the kernel's update leaves the batch statistics themselves in the buffers
                        rm, rv = torch.zeros(2, st.cout, dtype=torch.float32, device=x.device).unbind(0)
                        collect_running.append((nm, rm, rv))
                        momentum = 1.0
                out2 = res = None
                res_relu = False
                if st.res is not None:
                    R = ts[st.res]
                    res, res_relu = R.data, R.relu
                    if tuple(res.shape) != tuple(z.shape):
                        raise ValueError('residual %s does not match the block output %s (channel padding: the block '
                                         'input must have a power-of-two channel count)' % (tuple(res.shape), tuple(z.shape)))
                    out2 = torch.empty_like(z)
                mr = torch.empty(N, Cop, 2, dtype=torch.float32, device=x.device)
                with _timed('in_apply', desc):
                    check(lib().sdn_in_apply(ptr(z), ptr(stats), ptr(mr), ptr(res), ptr(out2), N, OH * OW, st.cout, Cop,
                                             float(nm.eps), 1 if st.act == 'lrelu' else 0, int(res_relu),
                                             momentum, ptr(rm), ptr(rv), stream()))
                T.stats = mr  # (mean, rstd) per (n, c): what the backward pass needs
                if st.res is not None:
                    if st.act != 'none':
                        raise NotImplementedError('activation after a residual add')
                    T.xhat = z
                    T.data = out2
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
            ts.append(T)
            geo.append((OH, OW))
        return ts, geo

    # ------------------------------------------------------------------ backward
    def backward(self, ts, geo, gouts, precision, need_input_grad, need_weight_grads=True, in_range=None):
        """gouts: {tensor index: grad buffer (channels-last, padded)}.  Returns (grad_input or None, [grad per param])."""
        dev = ts[0].data.device
        N = ts[0].data.shape[0]
        G = dict(gouts)
        pgrads = [None] * (2 * len(self.stages))
        arena = _ZeroArena(dev, torch.float32)
        if need_weight_grads:
            for si, st in enumerate(self.stages):
                Cip_, Cop_ = ts[st.src].data.shape[3], ts[si + 1].data.shape[3]
                arena.reserve(st.k * st.k * Cip_ * Cop_)          # dwp (the K slices of the weight gradient meet in it)
                if st.conv.bias is not None:
                    arena.reserve(max(st.conv.bias.numel(), Cop_))
        main = torch.cuda.current_stream(dev)
        # (one-conv chains -- the ResNet-18 encoder's layers -- are issue-bound on the host: the extra stream bookkeeping
        # cost their train step 9.0 -> 10.8 ms, tools/encoder_streams_lab.py)
        side = _wgrad_stream(dev) if need_weight_grads and len(self.stages) > 1 else None
        if side is not None:
            arena.materialize().record_stream(side)   # zeroed on this stream, carved up on both
        for si in range(len(self.stages) - 1, -1, -1):
            st = self.stages[si]
            T = ts[si + 1]
            g = G.pop(si + 1, None)
            if g is None:
                continue
            X = ts[st.src]
            IH, IW = geo[st.src]
            OH, OW = geo[si + 1]
            Cip = X.data.shape[3]
            Cop = T.data.shape[3]
            if not g.is_contiguous():
                g = g.contiguous()
            if st.res is not None:  # d(res) = g, before g is overwritten by dz
                if st.res in G:
                    G[st.res] = G[st.res] + g
                else:
                    G[st.res] = g.clone()
            bgrad = None
            if st.norm is not None:
                stored = T.xhat if T.xhat is not None else T.data
                sums = torch.empty(N, Cop, 2, dtype=torch.float64, device=dev)
                with _timed('in_bwd', '%d ch @%dx%d' % (st.cout, OH, OW)):
                    check(lib().sdn_in_bwd(ptr(g), ptr(stored), ptr(T.stats), ptr(sums), N, OH * OW, Cop, T.mode,
                                           stream()))
                if st.conv.bias is not None:
                    bgrad = arena.take(st.conv.bias.shape)  # a bias in front of InstanceNorm has zero gradient
            else:
                has_b = st.conv.bias is not None
                ordered = has_b and deterministic()   # the kernel's bias sum meets in float atomics
                bg = arena.take((Cop,)) if has_b and not ordered else None
                check(lib().sdn_act_bwd(ptr(g), ptr(T.data), ptr(bg), N * OH * OW, Cop, ACT[st.act], stream()))
                if ordered:
                    bgrad = g.reshape(-1, Cop)[:, :st.cout].sum(dim=0)
                elif has_b:
                    bgrad = bg[:st.cout].clone()
            dz = g
            # ---- weight gradient
            pad_mode = 1 if st.reflect else 0
            if need_weight_grads:
                if side is not None:
                    side.wait_stream(main)        # dz is final on the calling stream
                    dz.record_stream(side)        # and must outlive the side stream's reads of it
                with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
                    if st.kind == 'conv':
                        WL = cp.conv_wgrad(st.k, st.s, st.p, OH, OW)
                        rows_t, gath_t, Cr, Cc, GH, GW = dz, X.data, Cop, Cip, IH, IW
                        relu_rows, relu_gath, wpad = False, X.relu, pad_mode
                        R_, C_, (sr, sc) = st.cout, st.cin, st.str_fwd
                    else:
                        WL = cp.convT_wgrad(st.k, st.s, st.p, IH, IW)
                        rows_t, gath_t, Cr, Cc, GH, GW = X.data, dz, Cip, Cop, OH, OW
                        relu_rows, relu_gath, wpad = X.relu, False, 0
                        R_, C_, (sr, sc) = st.cin, st.cout, st.str_dgrad
                    ntaps = len(WL.taps)
                    dwp = arena.take((Cr, ntaps * Cc))
                    n_tiles = ((Cr + 127) // 128 if Cr > 64 else 1) * ((ntaps * Cc + 127) // 128)
                    splits = cp.wgrad_splits(N * WL.QH * WL.QW, n_tiles)
                    dy, dx = _taps_c(WL.taps)
                    desc = '%s k%d s%d %d->%d @%dx%d' % (st.kind, st.k, st.s, st.cin, st.cout, OH, OW)
                    flops = 2.0 * N * OH * OW * st.k * st.k * st.cin * st.cout / (st.s * st.s if st.kind == 'convT' else 1)
                    if st.kind == 'conv' and st.s == 1 and st.cout <= 8 and not deterministic():
                        # head layers (1-5 output channels): exact fp32 on the vector ALUs, input tile + halo kept in LDS
                        # (its blocks meet in dw through float atomics: the deterministic mode takes the MFMA kernel below)
                        with _timed('wgrad', desc + ' narrow', flops):
                            check(lib().sdn_conv_wgrad_narrow(ptr(rows_t), ptr(gath_t), ptr(dwp), N, WL.QH, WL.QW, Cr,
                                                              st.cout, GH, GW, Cc, ntaps, dy, dx, wpad, int(relu_rows),
                                                              int(relu_gath), stream()))
                    else:
                        ws, wsn = None, 0
                        if deterministic() and splits > 1:
                            ws = _workspace(dev, splits * Cr * ntaps * Cc * 4)
                            wsn = ws.numel()
                        with _timed('wgrad', desc + ' splits %d' % splits, flops):
                            check(lib().sdn_conv_wgrad(ptr(rows_t), ptr(gath_t), ptr(dwp), N, WL.QH, WL.QW, Cr, GH, GW, Cc,
                                                       WL.istride, ntaps, dy, dx, wpad, int(relu_rows), int(relu_gath),
                                                       splits, precision, ptr(ws), wsn, stream()))
                    # the gradient in the parameter's layout: the plan's taps cover the whole window, so every element is
                    # written exactly once (plain stores, no zero fill)
                    assert ntaps == st.k * st.k
                    wgrad = torch.empty(st.conv.weight.shape, dtype=torch.float32, device=dev)
                    tix = st.tix(WL.tapidx, dev)
                    check(lib().sdn_conv_unpack_grad(ptr(dwp), R_, C_, sr, sc, ptr(tix), ntaps, Cc, ptr(wgrad), 0, stream()))
                    pgrads[2 * si] = wgrad
                    pgrads[2 * si + 1] = bgrad
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
            if st.reflect:
                target = torch.empty(N, GHt, GWt, Cg, dtype=torch.float32, device=dev)
                acc = False
            elif have:
                target, acc = G[st.src], True
            else:
                target = torch.empty(N, IH, IW, Cg, dtype=torch.float32, device=dev)
                acc = False
            desc = '%s k%d s%d %d->%d @%dx%d' % (st.kind, st.k, st.s, st.cin, st.cout, OH, OW)
            flops = 2.0 * N * OH * OW * st.k * st.k * st.cin * st.cout / (st.s * st.s if st.kind == 'convT' else 1)
            if rr is not None:
                desc += ' ch %d:%d' % rr
                flops *= (rr[1] - rr[0]) / float(st.cin)
            narrow = (rr is not None and rr[1] - rr[0] <= 8 and st.kind == 'conv' and st.s == 1 and not acc
                      and st.k in NARROW_KW and precision == 3)
            with _timed('dgrad', desc + (' narrow' if narrow else ''), flops):
                if narrow:
                    L = launches[0]
                    _narrow_fwd(dz, N, OH, OW, Cop, target, GHt, GWt, Cg, st.narrow('dgrad', L.taps, L.tapidx, Cop, rr),
                                0, False, None, 0)
                else:
                    if not acc and any(not L.taps for L in launches):
                        target.zero_()   # phases no kernel tap reaches (a 1x1 stride-2 conv reads every other pixel only)
                    for L in launches:
                        if not L.taps:
                            continue
                        _gemm(dz, N, OH, OW, Cop, target, GHt, GWt, Cg, L, 0, False,
                              st.packed('dgrad', L.tapidx, precision, Cop, Cg, rr), None, 0, None, acc, precision)
            if st.reflect:
                if have:
                    out = G[st.src]
                else:
                    out = torch.empty(N, IH, IW, Cg, dtype=torch.float32, device=dev)
                check(lib().sdn_reflect_fold(ptr(target), ptr(out), N, IH, IW, Cg, st.reflect, int(have), stream()))
                G[st.src] = out
            else:
                G[st.src] = target
        if side is not None:
            main.wait_stream(side)
            for t in pgrads:
                if t is not None:
                    t.record_stream(main)
        return G.get(0), pgrads


class _ChainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, chain, nparts, *args):
        parts, params = args[:nparts], args[nparts:]
        precision = default_precision()
        norms = [st.norm for st in chain.stages if st.norm is not None]
        training = any(nm.training for nm in norms)
        if norms and not training and any(nm.track_running_stats for nm in norms):
            # torch would normalise with the running statistics here; the reference never calls .eval() on the textural
            # networks (no such call under textural/), so that mode is deliberately not implemented
            raise NotImplementedError('InstanceNorm2d(track_running_stats=True) in eval mode')
        with torch.no_grad():
            # channels-last input buffer, the parts side by side (what torch.cat + permute + pad would build)
            N, _, H, W = parts[0].shape
            C = sum(int(t.shape[1]) for t in parts)
            Cp = cp.cpad(C)
            if nparts == 1 and Cp == C:
                x = parts[0].permute(0, 2, 3, 1).contiguous()
            else:
                x = (torch.zeros if Cp != C else torch.empty)(N, H, W, Cp, dtype=torch.float32, device=parts[0].device)
                c0 = 0
                for t in parts:
                    x[..., c0:c0 + t.shape[1]] = t.permute(0, 2, 3, 1)
                    c0 += int(t.shape[1])
            keep = chain.__dict__.get('_keep_state')   # ConvChain.__call__(dual=True): a second autograd view follows
            running = [] if keep else None
            ts, geo = chain.forward(x, precision, training=training, collect_running=running)
        ctx.chain, ctx.ts, ctx.geo, ctx.precision = chain, ts, geo, precision
        ctx.nparts, ctx.part_channels = nparts, [int(t.shape[1]) for t in parts]
        if keep:
            chain.__dict__['_state'] = (ts, geo, precision, running)
        outs = tuple(ts[i].data for i in chain.outputs)
        return outs if len(outs) > 1 else outs[0]

    @staticmethod
    def backward(ctx, *gouts):
        nparts = ctx.nparts
        gparts, pg = _chain_backward(ctx, gouts, ctx.needs_input_grad[2:2 + nparts], any(ctx.needs_input_grad[2 + nparts:]))
        return (None, None) + tuple(gparts) + tuple(pg)


def _chain_backward(ctx, gouts, need_parts, need_w):
    chain = ctx.chain
    g = {}
    for ti, go in zip(chain.outputs, gouts):
        if go is not None:
            g[ti] = go.clone() if ti in g else go.contiguous().clone()
    nparts = ctx.nparts
    # channel range covering every part that wants a gradient
    starts = [sum(ctx.part_channels[:i]) for i in range(nparts)]
    lo = min([starts[i] for i in range(nparts) if need_parts[i]], default=0)
    hi = max([starts[i] + ctx.part_channels[i] for i in range(nparts) if need_parts[i]], default=0)
    with torch.no_grad():
        gin, pg = chain.backward(ctx.ts, ctx.geo, g, ctx.precision, any(need_parts), need_w,
                                 in_range=(lo, hi) if any(need_parts) else None)
    ctx.ts = None
    gparts = []
    for i in range(nparts):
        if need_parts[i] and gin is not None:
            a = starts[i] - lo
            gparts.append(gin[..., a:a + ctx.part_channels[i]].permute(0, 3, 1, 2))
        else:
            gparts.append(None)
    return gparts, pg


class _ChainSharedFn(torch.autograd.Function):
    """A second autograd view of a forward pass _ChainFn has already run (ConvChain.__call__(dual=True)): the outputs
    alias the stored activations, the backward pass runs the data-gradient kernels only (the stored activations are
    read-only in every backward kernel, so the two views can be back-propagated in either order)."""

    @staticmethod
    def forward(ctx, chain, state, nparts, *parts):
        ts, geo, precision = state[:3]
        ctx.chain, ctx.ts, ctx.geo, ctx.precision = chain, ts, geo, precision
        ctx.nparts, ctx.part_channels = nparts, [int(t.shape[1]) for t in parts]
        outs = tuple(ts[i].data.detach() for i in chain.outputs)   # new tensor objects on the same storage
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
