"""textural.models.networks on MI355X: the operator surface of the reference's pix2pixHD-style networks
(/root/reference/textural/models/networks.py) with every convolution, InstanceNorm and activation executed by the HIP
kernels of libsdn_hip.so (MFMA implicit GEMM; sdn_hip/conv.py), forward and backward.

Drop-in contract: same public names and signatures -- weights_init, get_norm_layer, define_G, define_D,
print_network, GANLoss, VGGLoss, LocalEnhancer, GlobalGenerator, ResnetBlock, Encoder(.forward(input, inst),
.generate_feat_dict), MultiscaleDiscriminator, NLayerDiscriminator, Vgg19 -- and the same state_dict keys / shapes
(`model.N.weight`, `model.N.conv_block.M.weight`, `scaleS_layerJ.0.weight`, InstanceNorm `running_mean` / `running_var` /
`num_batches_tracked`), so the reference's checkpoints load and textural/train.py / edit_*.py run unmodified.

How: each network still OWNS the torch modules the reference builds (nn.Conv2d, nn.InstanceNorm2d, ... in the same
nn.Sequential positions) but only as parameter containers; forward() hands the sequence to sdn_hip.conv.ConvChain.
CPU tensors raise NotImplementedError: there is no fallback path.
"""
import functools

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from sdn_hip import conv as _hc
from sdn_hip import ops as _ops


# ---------------------------------------------------------------------------------------------------------------------
# helpers with the reference's names (networks.py:14-85)

def weights_init(m):
    """N(0, 0.02) on every Conv* weight, N(1, 0.02) / 0 on BatchNorm2d (networks.py:14-21)."""
    cls = type(m).__name__
    _hc.invalidate_weight_caches()   # writes through `.data` do not advance the parameters' version counters
    if 'Conv' in cls:
        m.weight.data.normal_(0.0, 0.02)
    elif 'BatchNorm2d' in cls:
        m.weight.data.normal_(1.0, 0.02)
        m.bias.data.fill_(0)


def get_norm_layer(norm_type='instance'):
    """networks.py:24-31.  Only 'instance' runs on the fused kernels; 'batch' constructs but cannot execute."""
    if norm_type == 'instance':
        return functools.partial(nn.InstanceNorm2d, affine=False, track_running_stats=True)
    if norm_type == 'batch':
        return functools.partial(nn.BatchNorm2d, affine=True, track_running_stats=True)
    raise NotImplementedError('normalization layer [%s] is not found' % norm_type)


def get_non_linearity(layer_type='relu'):
    table = {'relu': functools.partial(nn.ReLU, inplace=True),
             'lrelu': functools.partial(nn.LeakyReLU, negative_slope=0.2, inplace=True),
             'elu': functools.partial(nn.ELU, inplace=True)}
    if layer_type not in table:
        raise NotImplementedError('nonlinearity activitation [%s] is not found' % layer_type)
    return table[layer_type]


def define_G(input_nc, output_nc, ngf, netG, n_downsample_global=3, n_blocks_global=9, n_local_enhancers=1,
             n_blocks_local=3, norm='instance', gpu_ids=[], isTrain=False):
    norm_layer = get_norm_layer(norm_type=norm)
    if netG == 'global':
        net = GlobalGenerator(input_nc, output_nc, ngf, n_downsample_global, n_blocks_global, norm_layer)
    elif netG == 'local':
        net = LocalEnhancer(input_nc, output_nc, ngf, n_downsample_global, n_blocks_global, n_local_enhancers,
                            n_blocks_local, norm_layer)
    elif netG == 'encoder':
        net = Encoder(input_nc, output_nc, ngf, n_downsample_global, norm_layer, isTrain=isTrain)
    else:
        raise NotImplementedError('generator [%s] not implemented' % netG)
    if len(gpu_ids) > 0:
        assert torch.cuda.is_available()
        net.cuda(gpu_ids[0])
    net.apply(weights_init)
    return net


def define_D(input_nc, ndf, n_layers_D, norm='instance', use_sigmoid=False, num_D=1, getIntermFeat=False, gpu_ids=[]):
    net = MultiscaleDiscriminator(input_nc, ndf, n_layers_D, get_norm_layer(norm_type=norm), use_sigmoid, num_D,
                                  getIntermFeat)
    if len(gpu_ids) > 0:
        assert torch.cuda.is_available()
        net.cuda(gpu_ids[0])
    net.apply(weights_init)
    return net


def print_network(net):
    if isinstance(net, list):
        net = net[0]
    print(net)
    print('Total number of parameters: %d' % sum(p.numel() for p in net.parameters()))


# ---------------------------------------------------------------------------------------------------------------------
# losses (networks.py:92-149): plain tensor arithmetic on the outputs of the fused networks

class GANLoss(nn.Module):
    """LSGAN (MSE) or BCE against a cached constant target; accepts the multiscale list-of-lists (networks.py:92-134)."""

    def __init__(self, use_lsgan=True, target_real_label=1.0, target_fake_label=0.0, tensor=torch.FloatTensor):
        super().__init__()
        self.real_label = target_real_label
        self.fake_label = target_fake_label
        self.real_label_var = None
        self.fake_label_var = None
        self.Tensor = tensor
        self.loss = nn.MSELoss() if use_lsgan else nn.BCELoss()

    def get_target_tensor(self, input, target_is_real):
        attr, value = ('real_label_var', self.real_label) if target_is_real else ('fake_label_var', self.fake_label)
        cached = getattr(self, attr)
        if cached is None or cached.numel() != input.numel() or cached.device != input.device:
            cached = torch.full(input.shape, float(value), dtype=input.dtype, device=input.device)
            setattr(self, attr, cached)
        return cached

    def __call__(self, input, target_is_real):
        if isinstance(input[0], list):
            loss = 0
            for scale in input:
                pred = scale[-1]
                loss = loss + self.loss(pred, self.get_target_tensor(pred, target_is_real))
            return loss
        pred = input[-1]
        return self.loss(pred, self.get_target_tensor(pred, target_is_real))


class L1Loss(nn.Module):
    """torch.nn.L1Loss() (`criterionFeat`, pix2pixHD_model.py:86): mean |input - target|.  Operand pairs that share one
    dense fp32 GPU layout -- the discriminator feature maps of the feature-matching loss -- take the fused HIP kernels
    (sdn_hip.ops.L1LossFn: 2 + 3 passes over the maps instead of torch's 6 + 5); anything else torch's own op."""

    def forward(self, input, target):
        if _ops.l1_loss_supported(input, target):
            return _ops.L1LossFn.apply(input, target)
        return F.l1_loss(input, target)


def load_pretrained(module, env_var, what, strict=True):
    """The reference downloads ImageNet weights through torchvision (`pretrained=True`: networks.py:470 VGG-19,
    derenderer.py:25 ResNet-18).  There is no network here, so the file must be named: `env_var` holds the path of a
    torchvision-format state_dict (torch.save).  Without it the module keeps its random initialisation ONLY when
    SDN_ALLOW_RANDOM_INIT=1 says so (benchmarks / tests); otherwise this raises -- a perceptual loss on random features
    or a randomly initialised encoder must never be a silent default."""
    import os
    path = os.environ.get(env_var)
    if path:
        state = torch.load(path, map_location='cpu')
        state = state.get('state_dict', state) if isinstance(state, dict) else state
        return module.load_state_dict(state, strict=strict)
    if os.environ.get('SDN_ALLOW_RANDOM_INIT') == '1':
        import warnings
        warnings.warn('%s: pretrained weights requested but %s is not set; RANDOM initialisation '
                      '(SDN_ALLOW_RANDOM_INIT=1)' % (what, env_var), RuntimeWarning)
        return None
    raise RuntimeError('%s needs pretrained weights: set %s to a torchvision-format state_dict file, or '
                       'SDN_ALLOW_RANDOM_INIT=1 to run with random weights (benchmarks only)' % (what, env_var))


class VGGLoss(nn.Module):
    """sum_i w_i * L1(VGG_i(x), VGG_i(y).detach())  (networks.py:137-149)."""

    def __init__(self, gpu_ids):
        super().__init__()
        vgg = Vgg19()
        # torchvision's vgg19().features keys are `features.N.*`; Vgg19's are `sliceK.N.*`
        path_keys = {('features.%d.' % i): ('slice%d.%d.' % (k, i)) for k, (a, b) in enumerate(_VGG19_SLICES, 1)
                     for i in range(a, b)}
        _load_vgg(vgg, path_keys)
        self.vgg = vgg.cuda()
        self.criterion = L1Loss()
        self.weights = [1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0]

    def forward(self, x, y):
        fx, fy = self.vgg(x), self.vgg(y)
        loss = 0
        for w, a, b in zip(self.weights, fx, fy):
            loss = loss + w * self.criterion(a, b.detach())
        return loss


# ---------------------------------------------------------------------------------------------------------------------
# fused execution of an nn.Sequential

class _Fused:
    """Compiles module lists into ConvChains once and caches them on the owning module (not in state_dict)."""

    def _chain(self, key, modules, in_channels, outputs=None):
        # the cache belongs to THIS module object: nn.DataParallel's replicas are shallow copies (`__dict__.copy()`), so a
        # replica would otherwise inherit chains whose stages point at the device-0 parameters
        cache = self.__dict__.get('_chains')
        if cache is None or cache.get('__owner__') != id(self):
            cache = {'__owner__': id(self)}
            self.__dict__['_chains'] = cache
        if key not in cache:
            stages, last = _hc.compile_sequential(list(modules))
            cache[key] = _hc.ConvChain(stages, outputs(stages) if outputs else [last], in_channels)
        return cache[key]


class _Pyramid(nn.AvgPool2d):
    """nn.AvgPool2d(3, stride=2, padding=[1, 1], count_include_pad=False) (networks.py:190, 392) -- the same module, so
    `print(net)` and attribute access look like the reference's -- whose forward runs the HIP kernels of csrc/fast_pool.hip:
    torch's own backward of this op returns wrong gradients on ROCm for channels-last-strided inputs (which is what the
    generator hands the discriminator)."""

    def __init__(self):
        super().__init__(3, stride=2, padding=[1, 1], count_include_pad=False)

    def forward(self, x):
        return _ops.avg_pool_3x3_s2(x)


def _c7s1(cin, cout, norm_layer, act):
    """ReflectionPad2d(3) + 7x7 conv (+ norm + activation): the stem / head group of networks.py:218,236,291,306."""
    mods = [nn.ReflectionPad2d(3), nn.Conv2d(cin, cout, kernel_size=7, padding=0)]
    if norm_layer is not None:
        mods.append(norm_layer(cout))
    mods.append(act)
    return mods


class ResnetBlock(nn.Module):
    """x + conv_block(x), conv_block = [pad, conv3, norm, act, (dropout), pad, conv3, norm]  (networks.py:244-283)."""

    def __init__(self, dim, padding_type, norm_layer, activation=nn.ReLU(True), use_dropout=False):
        super().__init__()
        self.conv_block = self.build_conv_block(dim, padding_type, norm_layer, activation, use_dropout)

    def build_conv_block(self, dim, padding_type, norm_layer, activation, use_dropout):
        def padded_conv():
            if padding_type == 'reflect':
                return [nn.ReflectionPad2d(1), nn.Conv2d(dim, dim, kernel_size=3, padding=0)]
            if padding_type == 'replicate':
                return [nn.ReplicationPad2d(1), nn.Conv2d(dim, dim, kernel_size=3, padding=0)]
            if padding_type == 'zero':
                return [nn.Conv2d(dim, dim, kernel_size=3, padding=1)]
            raise NotImplementedError('padding [%s] is not implemented' % padding_type)
        block = padded_conv() + [norm_layer(dim), activation]
        if use_dropout:
            block.append(nn.Dropout(0.5))
        block += padded_conv() + [norm_layer(dim)]
        return nn.Sequential(*block)

    def forward(self, x):
        raise RuntimeError('ResnetBlock runs as part of its generator\'s fused chain')


class GlobalGenerator(nn.Module, _Fused):
    """c7s1-ngf, n_downsampling x (3x3 stride-2 conv), n_blocks ResnetBlocks, n_downsampling x ConvTranspose2d,
    c7s1-output_nc + tanh (networks.py:211-239)."""

    def __init__(self, input_nc, output_nc, ngf=64, n_downsampling=3, n_blocks=9, norm_layer=nn.BatchNorm2d,
                 padding_type='reflect'):
        assert n_blocks >= 0
        super().__init__()
        self.input_nc = input_nc
        act = nn.ReLU(True)
        layers = _c7s1(input_nc, ngf, norm_layer, act)
        ch = ngf
        for _ in range(n_downsampling):
            layers += [nn.Conv2d(ch, ch * 2, kernel_size=3, stride=2, padding=1), norm_layer(ch * 2), act]
            ch *= 2
        for _ in range(n_blocks):
            layers.append(ResnetBlock(ch, padding_type=padding_type, activation=act, norm_layer=norm_layer))
        for _ in range(n_downsampling):
            layers += [nn.ConvTranspose2d(ch, ch // 2, kernel_size=3, stride=2, padding=1, output_padding=1),
                       norm_layer(ch // 2), act]
            ch //= 2
        layers += _c7s1(ngf, output_nc, None, nn.Tanh())
        self.model = nn.Sequential(*layers)

    accepts_parts = True  # forward() also takes the list of tensors the caller would torch.cat (ConvChain.__call__)

    def forward(self, input):
        return self._chain('model', self.model, self.input_nc)(input)[0]


class LocalEnhancer(nn.Module, _Fused):
    """networks.py:156-206: a GlobalGenerator trunk at half resolution plus per-level enhancer branches."""

    def __init__(self, input_nc, output_nc, ngf=32, n_downsample_global=3, n_blocks_global=9, n_local_enhancers=1,
                 n_blocks_local=3, norm_layer=nn.BatchNorm2d, padding_type='reflect'):
        super().__init__()
        self.n_local_enhancers = n_local_enhancers
        self.input_nc = input_nc
        trunk = GlobalGenerator(input_nc, output_nc, ngf * (2 ** n_local_enhancers), n_downsample_global,
                                n_blocks_global, norm_layer).model
        self.model = nn.Sequential(*list(trunk)[:-3])  # without the final pad / conv / tanh
        for n in range(1, n_local_enhancers + 1):
            ch = ngf * (2 ** (n_local_enhancers - n))
            down = _c7s1(input_nc, ch, norm_layer, nn.ReLU(True)) + [
                nn.Conv2d(ch, ch * 2, kernel_size=3, stride=2, padding=1), norm_layer(ch * 2), nn.ReLU(True)]
            up = [ResnetBlock(ch * 2, padding_type=padding_type, norm_layer=norm_layer) for _ in range(n_blocks_local)]
            up += [nn.ConvTranspose2d(ch * 2, ch, kernel_size=3, stride=2, padding=1, output_padding=1),
                   norm_layer(ch), nn.ReLU(True)]
            if n == n_local_enhancers:
                up += _c7s1(ngf, output_nc, None, nn.Tanh())
            setattr(self, 'model%d_1' % n, nn.Sequential(*down))
            setattr(self, 'model%d_2' % n, nn.Sequential(*up))
        self.downsample = _Pyramid()

    def forward(self, input):
        pyramid = [input]
        for _ in range(self.n_local_enhancers):
            pyramid.append(self.downsample(pyramid[-1]))
        out = self._chain('model', self.model, self.input_nc)(pyramid[-1])[0]
        for n in range(1, self.n_local_enhancers + 1):
            down = getattr(self, 'model%d_1' % n)
            up = getattr(self, 'model%d_2' % n)
            x = pyramid[self.n_local_enhancers - n]
            mid = self._chain('d%d' % n, down, self.input_nc)(x)[0] + out
            out = self._chain('u%d' % n, up, mid.shape[1])(mid)[0]
        return out


class Encoder(nn.Module, _Fused):
    """Feature encoder + instance-wise average pooling (networks.py:286-346)."""

    def __init__(self, input_nc, output_nc, ngf=32, n_downsampling=4, norm_layer=nn.BatchNorm2d, isTrain=True):
        super().__init__()
        self.isTrain = isTrain
        self.output_nc = output_nc
        self.input_nc = input_nc
        layers = _c7s1(input_nc, ngf, norm_layer, nn.ReLU(True))
        ch = ngf
        for _ in range(n_downsampling):
            layers += [nn.Conv2d(ch, ch * 2, kernel_size=3, stride=2, padding=1), norm_layer(ch * 2), nn.ReLU(True)]
            ch *= 2
        for _ in range(n_downsampling):
            layers += [nn.ConvTranspose2d(ch, ch // 2, kernel_size=3, stride=2, padding=1, output_padding=1),
                       norm_layer(ch // 2), nn.ReLU(True)]
            ch //= 2
        layers += _c7s1(ngf, output_nc, None, nn.Tanh())
        self.model = nn.Sequential(*layers)

    @staticmethod
    def _disambiguate(inst):
        """`inst[i] = inst[i] * batch + i` in place, as the reference does (networks.py:313-316): the same id in two
        images of the batch must not be pooled together."""
        bs = inst.size(0)
        for i in range(bs):
            inst[i] = inst[i] * bs + i
        return inst

    def _pooled(self, input, inst):
        feats = self._chain('model', self.model, self.input_nc)(input)[0]          # [N, C, H, W]
        inst = self._disambiguate(inst)
        ids, inverse = torch.unique(inst.reshape(-1).long(), return_inverse=True)  # one sync for the id count
        N, C, H, W = feats.shape
        seg = inverse.to(torch.int32).reshape(N, H, W)
        out, means = _ops.SegmentMeanFn.apply(feats.contiguous(), seg, int(ids.numel()))
        return out, ids, means

    def forward(self, input, inst):
        out, _, _ = self._pooled(input, inst)
        return (out, 0) if self.isTrain else out

    def generate_feat_dict(self, input, inst):
        _, ids, means = self._pooled(input, inst)
        table = means.t().detach().cpu().tolist()
        return {int(i): [float(v) for v in row] for i, row in zip(ids.cpu().tolist(), table)}


class NLayerDiscriminator(nn.Module, _Fused):
    """PatchGAN: 4x4 convs, stride 2 x n_layers then stride 1 x 2, LeakyReLU(0.2), norm on the middle layers
    (networks.py:412-461)."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=nn.BatchNorm2d, use_sigmoid=False, getIntermFeat=False):
        super().__init__()
        self.getIntermFeat = getIntermFeat
        self.n_layers = n_layers
        self.input_nc = input_nc
        kw, padw = 4, int(np.ceil((4 - 1.0) / 2))
        groups = [[nn.Conv2d(input_nc, ndf, kernel_size=kw, stride=2, padding=padw), nn.LeakyReLU(0.2, True)]]
        nf = ndf
        for n in range(1, n_layers + 1):
            nf_prev, nf = nf, min(nf * 2, 512)
            groups.append([nn.Conv2d(nf_prev, nf, kernel_size=kw, stride=2 if n < n_layers else 1, padding=padw),
                           norm_layer(nf), nn.LeakyReLU(0.2, True)])
        groups.append([nn.Conv2d(nf, 1, kernel_size=kw, stride=1, padding=padw)])
        if use_sigmoid:
            groups.append([nn.Sigmoid()])
        self.use_sigmoid = use_sigmoid
        if getIntermFeat:
            for n, g in enumerate(groups):
                setattr(self, 'model' + str(n), nn.Sequential(*g))
        else:
            self.model = nn.Sequential(*[m for g in groups for m in g])

    def _groups(self):
        if self.getIntermFeat:
            return [getattr(self, 'model' + str(n)) for n in range(self.n_layers + 2)]
        return [self.model]

    def forward(self, input, detach_weights=False):
        return _run_discriminator(self, 'model', self._groups(), self.input_nc, self.getIntermFeat, input,
                                  self.use_sigmoid, detach_weights)


def _run_discriminator(owner, key, groups, input_nc, interm, input, use_sigmoid=False, detach_weights=False, dual=False):
    """One PatchGAN column as a single fused chain; with getIntermFeat every group's output is returned.
    detach_weights: gradients flow to `input` only (no weight-gradient kernels are launched).
    dual: one forward pass, returned twice -- (as for input.detach(), as with detach_weights): ConvChain.__call__."""
    mods = [m for g in groups for m in g if not isinstance(m, nn.Sigmoid)]
    n_groups = len([g for g in groups if not (len(g) == 1 and isinstance(g[0], nn.Sigmoid))])

    def outs(stages):
        return list(range(1, len(stages) + 1)) if interm else [len(stages)]

    def finish(res):
        assert not interm or len(res) == n_groups
        if use_sigmoid:
            res = res + [torch.sigmoid(res[-1])] if interm else [torch.sigmoid(res[-1])]
        return res if interm else res[0]
    chain = owner._chain(key, mods, input_nc, outs)
    if dual:
        res_w, res_x, running = chain(input, dual=True)
        return finish(res_w), finish(res_x), running
    return finish(chain(input, detach_weights=detach_weights))


import threading as _threading  # noqa: E402

_STREAMS_LOCK = _threading.Lock()


def _tensors(obj):
    """the tensors of a (nested) list / tuple"""
    if isinstance(obj, torch.Tensor):
        return [obj]
    return [t for o in obj for t in _tensors(o)]


class MultiscaleDiscriminator(nn.Module, _Fused):
    """num_D PatchGANs on an average-pooled pyramid; attribute / key names as the reference (networks.py:368-407)."""
    accepts_parts = True

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=nn.BatchNorm2d, use_sigmoid=False, num_D=3,
                 getIntermFeat=False):
        super().__init__()
        self.num_D = num_D
        self.n_layers = n_layers
        self.getIntermFeat = getIntermFeat
        self.input_nc = input_nc
        self.use_sigmoid = use_sigmoid
        for i in range(num_D):
            netD = NLayerDiscriminator(input_nc, ndf, n_layers, norm_layer, use_sigmoid, getIntermFeat)
            if getIntermFeat:
                for j in range(n_layers + 2):
                    setattr(self, 'scale%d_layer%d' % (i, j), getattr(netD, 'model' + str(j)))
            else:
                setattr(self, 'layer' + str(i), netD.model)
        self.downsample = _Pyramid()
        # side streams per device, created here so that nn.DataParallel's replicas (shallow copies of __dict__, new objects
        # every forward) SHARE this dict with the module they were copied from instead of building streams of their own
        self._streams = {}

    def forward(self, input, detach_weights=False):
        """detach_weights (extension): score `input` without accumulating gradients into this discriminator's own
        parameters -- what the generator loss needs (pix2pixHD_model.py:210).  `input` may be the list of tensors the
        caller would otherwise torch.cat (extension): the input gradient is then only computed for the parts that
        require one."""
        return self._run(input, detach_weights, False)

    def forward_dual(self, input):
        """(forward(input.detach()), forward(input, detach_weights=True), second_pass) from ONE pass over the pyramid
        (extension).  pix2pixHD_model.py:191-210 scores the same fake image twice with the same discriminator weights --
        once detached for the discriminator's loss, once attached for the generator's; the activations are identical,
        so they are computed once and back-propagated twice (weight gradients through the first view, the input
        gradient through the second).  The InstanceNorm running statistics see what the two passes would have done
        to them: the first update happens here, and the caller invokes `second_pass()` at the point where the second
        forward would have run (the reference scores the real image in between, :194)."""
        res_w, res_x, running = self._run(input, False, True)
        _hc.update_running(running)
        return res_w, res_x, (lambda: _hc.update_running(running))

    def _run(self, input, detach_weights, dual):
        # the pyramid first: a list of tensors (the un-concatenated parts, see ConvChain.__call__) is pooled part by part
        pyramid = [input]
        for i in range(1, self.num_D):
            x = pyramid[-1]
            pyramid.append([self.downsample(t) for t in x] if isinstance(x, (list, tuple)) else self.downsample(x))
        outs = [None] * self.num_D
        side = self._side_streams(pyramid[0]) if self.num_D > 1 else None
        if side:
            # The coarse columns have too few output tiles to fill 256 CUs (13 x 40 positions at the third scale):
            # each runs on its own HIP stream, concurrently with the full-resolution column on the caller's stream.
            # A column always uses the same stream, so its packed weights and scratch are stream-ordered; autograd runs
            # every backward node on its forward's stream and synchronises gradients that cross streams.
            cur = torch.cuda.current_stream()
            for i in range(self.num_D - 1, 0, -1):
                st = side[i - 1]
                st.wait_stream(cur)
                for t in _tensors(pyramid[i]):
                    t.record_stream(st)
                with torch.cuda.stream(st):
                    outs[i] = self._column(i, pyramid[i], detach_weights, dual)
            outs[0] = self._column(0, pyramid[0], detach_weights, dual)
            for i in range(1, self.num_D):
                cur.wait_stream(side[i - 1])
                for t in _tensors(outs[i][:2] if dual else outs[i]):
                    t.record_stream(cur)   # allocated on the side stream, consumed by the caller's
        else:
            for i in range(self.num_D):
                outs[i] = self._column(i, pyramid[i], detach_weights, dual)
        result, result_x, running = [], [], []
        for r in outs:
            if dual:
                r, rx, run = r
                result_x.append(rx if self.getIntermFeat else [rx])
                running += run
            result.append(r if self.getIntermFeat else [r])
        return (result, result_x, running) if dual else result

    def _column(self, i, x, detach_weights, dual):
        s = self.num_D - 1 - i
        if self.getIntermFeat:
            groups = [getattr(self, 'scale%d_layer%d' % (s, j)) for j in range(self.n_layers + 2)]
        else:
            groups = [getattr(self, 'layer' + str(s))]
        return _run_discriminator(self, 'scale%d' % s, groups, self.input_nc, self.getIntermFeat, x, self.use_sigmoid,
                                  detach_weights, dual)

    def _side_streams(self, x):
        """One extra HIP stream per coarse column (SDN_D_STREAMS=0 keeps every column on the caller's stream)."""
        import os
        t = x[0] if isinstance(x, (list, tuple)) else x
        if not t.is_cuda or os.environ.get('SDN_D_STREAMS', '1') == '0':
            return None
        # per device, not per module object: nn.DataParallel's replicas (new objects every forward, one per device, each
        # in its own thread) share the dict created in __init__ with the module they were copied from
        cache = self.__dict__.get('_streams')
        if cache is None:                       # (a module unpickled from an older build)
            cache = self.__dict__.setdefault('_streams', {})
        key = t.device
        streams = cache.get(key)
        if streams is None:
            with _STREAMS_LOCK:                 # replicas run in threads; create a device's streams once
                streams = cache.get(key)
                if streams is None:
                    streams = cache[key] = [torch.cuda.Stream(device=t.device) for _ in range(self.num_D - 1)]
        return streams


# ---------------------------------------------------------------------------------------------------------------------
# VGG-19 feature slices for the perceptual loss (networks.py:467-497).  torchvision is not a dependency here: the layer
# table below reproduces torchvision's `vgg19().features[:30]` indices, so `slice{1..5}.{idx}.weight` keys match and a
# torchvision checkpoint loads with load_state_dict.  Pretrained weights must be supplied by the caller (no network).
_VGG19_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512]
_VGG19_SLICES = [(0, 2), (2, 7), (7, 12), (12, 21), (21, 30)]


def _load_vgg(vgg, key_map):
    import os
    path = os.environ.get('SDN_VGG19_WEIGHTS')
    if not path:
        return load_pretrained(vgg, 'SDN_VGG19_WEIGHTS', 'VGGLoss (networks.py:470 vgg19(pretrained=True))')
    state = torch.load(path, map_location='cpu')
    out = {}
    for k, v in state.items():
        for a, b in key_map.items():
            if k.startswith(a):
                out[b + k[len(a):]] = v
                break
        else:
            if k.startswith('slice'):
                out[k] = v
    return vgg.load_state_dict(out, strict=True)


def _vgg19_features():
    feats, cin = [], 3
    for v in _VGG19_CFG:
        if v == 'M':
            feats.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            feats += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    return feats[:30]


class Vgg19(nn.Module, _Fused):
    def __init__(self, requires_grad=False):
        super().__init__()
        feats = _vgg19_features()
        for n, (a, b) in enumerate(_VGG19_SLICES, 1):
            seq = nn.Sequential()
            for i in range(a, b):
                seq.add_module(str(i), feats[i])
            setattr(self, 'slice%d' % n, seq)
        if not requires_grad:
            for p in self.parameters():
                p.requires_grad = False

    def forward(self, X):
        outs, h = [], X
        for n in range(1, 6):
            seq = getattr(self, 'slice%d' % n)
            run, cin = [], h.shape[1]
            for m in seq:
                if isinstance(m, nn.MaxPool2d):
                    if run:
                        h = self._chain('s%d_%d' % (n, id(run[0])), run, cin)(h)[0]
                        run = []
                    h = torch.nn.functional.max_pool2d(h, 2, 2)
                    cin = h.shape[1]
                else:
                    run.append(m)
            if run:
                h = self._chain('s%d_%d' % (n, id(run[0])), run, cin)(h)[0]
            outs.append(h)
        return outs
