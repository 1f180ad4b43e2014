"""Executor pieces for BatchNorm networks on the HIP kernels: what the derender3d encoder (a torchvision ResNet-18,
geometric/derender3d/models/derenderer.py:25-27,48) needs besides convolutions.

    conv2d(module, x)                 nn.Conv2d through the MFMA implicit-GEMM kernels (a one-stage sdn_hip.conv.ConvChain:
                                      forward, data gradient, weight gradient)
    batch_norm(module, x, res, relu)  nn.BatchNorm2d, train (batch statistics, running statistics updated) or eval, fused
                                      with the BasicBlock tail `relu(bn(x) + identity)` -- csrc/conv_bn.hip
    max_pool_3x3_s2(x)                nn.MaxPool2d(3, 2, 1) of the ResNet stem
    global_avg_pool(x)                nn.AdaptiveAvgPool2d(1) + flatten

Tensors travel between these ops as NCHW *views* of channels-last storage (what ConvChain returns), so no layout
conversion happens inside the network; each op is a torch.autograd.Function whose backward is again HIP kernels.  The
modules stay ordinary torch modules (parameter containers: same state_dict keys, same initialisation).  There is no CPU
path: CPU tensors raise NotImplementedError.
"""
import torch

from . import check, lib, ptr, stream
from . import conv as _hc


def _need_gpu(t, what):
    if not t.is_cuda:
        raise NotImplementedError('%s only runs on the GPU (got %s); there is no CPU or PyTorch fallback' % (what, t.device))
    if t.dtype != torch.float32:
        raise TypeError('%s expects float32, got %s' % (what, t.dtype))


def _cl(x):
    """[N, C, H, W] (any strides) -> contiguous channels-last [N, H, W, C]; free when x already is such a view."""
    return x.permute(0, 2, 3, 1).contiguous()


def conv2d(m, x):
    """m: nn.Conv2d (square kernel, any stride / zero padding, groups 1).  x [N, Cin, H, W] -> [N, Cout, OH, OW]."""
    _need_gpu(x, 'conv2d')
    chain = m.__dict__.get('_sdn_chain')
    if chain is None or chain[0] != id(m):   # a DataParallel replica is a shallow copy: it must not reuse the original's chain
        if m.groups != 1 or m.dilation[0] != 1 or m.kernel_size[0] != m.kernel_size[1] or m.padding_mode != 'zeros':
            raise NotImplementedError('conv2d: %r' % (m,))
        chain = (id(m), _hc.ConvChain([_hc.Stage('conv', m, 0)], [1], m.in_channels))
        m.__dict__['_sdn_chain'] = chain
    return chain[1](x)[0]


class _BatchNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, res, bn, relu):
        xl = _cl(x)
        rl = _cl(res) if res is not None else None
        N, H, W, C = xl.shape
        rows = N * H * W
        dev = xl.device
        training = bool(bn.training or bn.running_mean is None)
        out = torch.empty_like(xl)
        mr = torch.empty(C, 2, dtype=torch.float32, device=dev)
        ss = torch.empty(C, 2, dtype=torch.float32, device=dev)
        sums = torch.empty(C, 2, dtype=torch.float64, device=dev)
        rm = rv = None
        momentum = 0.0
        if bn.track_running_stats and bn.running_mean is not None:
            rm, rv = bn.running_mean, bn.running_var
            if training:
                if bn.num_batches_tracked is not None:
                    bn.num_batches_tracked += 1
                momentum = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
        if training and rows == 1:
            raise ValueError('Expected more than 1 value per channel when training')
        check(lib().sdn_bn_forward(ptr(xl), rows, C, ptr(gamma), ptr(beta), ptr(rm),
                                   ptr(rv), float(momentum), float(bn.eps), int(training), ptr(rl), int(relu), ptr(out),
                                   ptr(mr), ptr(ss), ptr(sums), stream()))
        ctx.save_for_backward(xl, out if relu else None, mr, gamma)
        ctx.training, ctx.relu, ctx.has_res = training, bool(relu), res is not None
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        xl, out, mr, gamma = ctx.saved_tensors
        gl = _cl(g)
        rows, C = xl.shape[0] * xl.shape[1] * xl.shape[2], xl.shape[3]
        gm = torch.empty_like(xl)
        dx = torch.empty_like(xl)
        sums = torch.empty(C, 2, dtype=torch.float64, device=xl.device)
        check(lib().sdn_bn_backward(ptr(gl), ptr(out), ptr(xl), ptr(mr), ptr(gamma), rows, C, int(ctx.training),
                                    int(ctx.relu), ptr(gm), ptr(dx), ptr(sums), stream()))
        dgamma = sums[:, 1].float() if gamma is not None and ctx.needs_input_grad[1] else None
        dbeta = sums[:, 0].float() if ctx.needs_input_grad[2] else None
        dres = gm.permute(0, 3, 1, 2) if ctx.has_res and ctx.needs_input_grad[3] else None
        return dx.permute(0, 3, 1, 2), dgamma, dbeta, dres, None, None


def batch_norm(bn, x, res=None, relu=False):
    """relu?(bn(x) + res?).  bn: nn.BatchNorm2d (its mode, parameters and running statistics are honoured and updated)."""
    _need_gpu(x, 'batch_norm')
    if res is not None and res.shape != x.shape:
        raise ValueError('residual %s does not match %s' % (tuple(res.shape), tuple(x.shape)))
    return _BatchNormFn.apply(x, bn.weight, bn.bias, res, bn, relu)


class _MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        xl = _cl(x)
        N, H, W, C = xl.shape
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        out = torch.empty(N, OH, OW, C, dtype=torch.float32, device=xl.device)
        idx = torch.empty(N, OH, OW, C, dtype=torch.int8, device=xl.device)
        check(lib().sdn_maxpool3x3s2_fwd(ptr(xl), N, H, W, C, ptr(out), ptr(idx), stream()))
        ctx.save_for_backward(idx)
        ctx.shape = (N, H, W, C)
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        idx, = ctx.saved_tensors
        N, H, W, C = ctx.shape
        gin = torch.empty(N, H, W, C, dtype=torch.float32, device=idx.device)
        check(lib().sdn_maxpool3x3s2_bwd(ptr(_cl(g)), ptr(idx), N, H, W, C, ptr(gin), stream()))
        return gin.permute(0, 3, 1, 2)


def max_pool_3x3_s2(x):
    _need_gpu(x, 'max_pool_3x3_s2')
    return _MaxPoolFn.apply(x)


class _AvgPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        xl = _cl(x)
        N, H, W, C = xl.shape
        out = torch.empty(N, C, dtype=torch.float32, device=xl.device)
        check(lib().sdn_avgpool_global(ptr(xl), N, H * W, C, ptr(out), 0, stream()))
        ctx.shape = (N, H, W, C)
        return out

    @staticmethod
    def backward(ctx, g):
        N, H, W, C = ctx.shape
        gin = torch.empty(N, H, W, C, dtype=torch.float32, device=g.device)
        check(lib().sdn_avgpool_global(ptr(g.contiguous()), N, H * W, C, ptr(gin), 1, stream()))
        return gin.permute(0, 3, 1, 2)


def global_avg_pool(x):
    """AdaptiveAvgPool2d(1) + flatten: [N, C, H, W] -> [N, C]."""
    _need_gpu(x, 'global_avg_pool')
    return _AvgPoolFn.apply(x)

