"""CPU oracle of the textural networks -- TEST INFRASTRUCTURE ONLY (checker, never product).

A functional restatement, in plain torch CPU ops, of what the reference's modules compute
(/root/reference/textural/models/networks.py); every function takes the network's `state_dict` (reference key names)
plus the input and evaluates the layers in order.  Pinned against the REAL reference modules: tests/golden/
make_textural_golden.py imports the reference's networks.py in the build container (with a stub `torchvision`, which only
its Vgg19 touches) and stores inputs, weights, outputs and gradients in tests/golden/textural_golden.npz;
tests/test_textural_oracle.py demands this file reproduce them.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline may import it.

All functions work in the dtype of `x` (float32 to mirror the reference, float64 for a tighter yardstick) and are
differentiable by torch autograd, so the same code checks backward passes.
"""
import torch
import torch.nn.functional as F

EPS = 1e-5  # nn.InstanceNorm2d default, networks.py:27


def _w(sd, key, x):
    return sd[key].to(dtype=x.dtype)


def _inorm(x):
    """nn.InstanceNorm2d(affine=False, track_running_stats=True) in TRAINING mode (the textural scripts never call
    .eval()): per-(n, c) biased variance over H x W, eps inside the square root (networks.py:24-30)."""
    return F.instance_norm(x, None, None, None, None, True, 0.1, EPS)


def _c7(sd, key, x):
    """ReflectionPad2d(3) + Conv2d(k=7, padding=0)  (networks.py:218, 236, 291, 306)"""
    return F.conv2d(F.pad(x, (3, 3, 3, 3), mode='reflect'), _w(sd, key + '.weight', x), _w(sd, key + '.bias', x))


def _down(sd, key, x):
    """Conv2d(k=3, stride=2, padding=1)  (networks.py:224, 297)"""
    return F.conv2d(x, _w(sd, key + '.weight', x), _w(sd, key + '.bias', x), stride=2, padding=1)


def _up(sd, key, x):
    """ConvTranspose2d(k=3, stride=2, padding=1, output_padding=1)  (networks.py:233, 303)"""
    return F.conv_transpose2d(x, _w(sd, key + '.weight', x), _w(sd, key + '.bias', x), stride=2, padding=1,
                              output_padding=1)


def _resblock(sd, prefix, x, relu=F.relu):
    """x + [pad1, conv3, IN, ReLU, pad1, conv3, IN](x)  (networks.py:244-283, padding_type 'reflect', no dropout)"""
    h = F.conv2d(F.pad(x, (1, 1, 1, 1), mode='reflect'), _w(sd, prefix + '.conv_block.1.weight', x),
                 _w(sd, prefix + '.conv_block.1.bias', x))
    h = relu(_inorm(h))
    h = F.conv2d(F.pad(h, (1, 1, 1, 1), mode='reflect'), _w(sd, prefix + '.conv_block.5.weight', x),
                 _w(sd, prefix + '.conv_block.5.bias', x))
    return x + _inorm(h)


def global_generator(sd, x, n_downsampling, n_blocks, collect=None, relu_masks=None, preacts=None):
    """GlobalGenerator.forward (networks.py:211-239).  `collect`: list that receives every stage's activation.
    relu_masks: optional list of 0/1 tensors, one per ReLU in execution order (stem, downsampling stages, one inside each
    block, upsampling stages); ReLU(t) is then evaluated as t * mask, so that a BACKWARD pass can be compared under the
    activation pattern another implementation's forward produced (a 1e-5 forward difference otherwise flips a few units,
    each flip a finite change of the gradient)."""
    def keep(t):
        if collect is not None:
            collect.append(t)
        return t
    masks = iter(relu_masks) if relu_masks is not None else None

    def relu(t):
        if preacts is not None:
            preacts.append(t)      # every ReLU's input, in execution order (how close a unit is to its kink)
        return F.relu(t) if masks is None else t * next(masks).to(t.dtype)
    i = 1
    h = keep(relu(_inorm(_c7(sd, 'model.%d' % i, x))))
    i += 3
    for _ in range(n_downsampling):
        h = keep(relu(_inorm(_down(sd, 'model.%d' % i, h))))
        i += 3
    for _ in range(n_blocks):
        h = keep(_resblock(sd, 'model.%d' % i, h, relu))
        i += 1
    for _ in range(n_downsampling):
        h = keep(relu(_inorm(_up(sd, 'model.%d' % i, h))))
        i += 3
    return keep(torch.tanh(_c7(sd, 'model.%d' % (i + 1), h)))


def encoder_features(sd, x, n_downsampling):
    """Encoder.model (networks.py:286-308): the generator skeleton without residual blocks."""
    return global_generator(sd, x, n_downsampling, 0)


def encoder(sd, x, inst, n_downsampling):
    """Encoder.forward (networks.py:310-326): instance-wise average pooling of the features.  `inst` [N,1,H,W] integer
    ids; ids are made unique per batch element (id * N + n) exactly as the reference does, without mutating the input."""
    out = encoder_features(sd, x, n_downsampling)
    N = x.shape[0]
    ids = inst.clone().long()
    for n in range(N):
        ids[n] = ids[n] * N + n
    res = out.clone()
    for i in torch.unique(ids).tolist():
        mask = (ids == i).expand_as(out)                     # same pixels in every channel
        for c in range(out.shape[1]):
            m = mask[:, c]
            res[:, c][m] = out[:, c][m].mean()
    return res


def nlayer_discriminator(sd, x, prefix, n_layers=3, lrelu_masks=None):
    """NLayerDiscriminator with getIntermFeat (networks.py:412-461): returns the n_layers + 2 group outputs.
    prefix: e.g. 'scale0_layer' -> keys 'scale0_layer{j}.0.weight'.
    lrelu_masks: optional list of n_layers + 1 boolean tensors; LeakyReLU(t) is then evaluated as t * (mask ? 1 : 0.2), i.e.
    under the slope pattern another implementation's forward produced (as `relu_masks` of global_generator)."""
    feats = []
    h = x
    for j in range(n_layers + 2):
        key = '%s%d.0' % (prefix, j)
        stride = 2 if j < n_layers else 1
        h = F.conv2d(h, _w(sd, key + '.weight', x), _w(sd, key + '.bias', x), stride=stride, padding=2)
        if 0 < j < n_layers + 1:
            h = _inorm(h)
        if j < n_layers + 1:
            if lrelu_masks is None:
                h = F.leaky_relu(h, 0.2)
            else:
                m = lrelu_masks[j]
                h = h * torch.where(m, torch.ones((), dtype=h.dtype), torch.full((), 0.2, dtype=h.dtype))
        feats.append(h)
    return feats


def multiscale_discriminator(sd, x, num_D, n_layers=3, lrelu_masks=None):
    """MultiscaleDiscriminator.forward with getIntermFeat (networks.py:395-407): scale num_D-1 sees the full image, each
    following one AvgPool2d(3, stride 2, padding 1, count_include_pad=False) of the previous input."""
    result = []
    h = x
    for i in range(num_D):
        result.append(nlayer_discriminator(sd, h, 'scale%d_layer' % (num_D - 1 - i), n_layers,
                                           None if lrelu_masks is None else lrelu_masks[i]))
        if i != num_D - 1:
            h = F.avg_pool2d(h, 3, stride=2, padding=1, count_include_pad=False)
    return result


def pix2pixhd_step_losses(sdG, sdD, sdE, batch, opt, masks=None):
    """The loss terms of one training iteration as the reference computes them (Pix2PixHDModel.forward,
    textural/models/pix2pixHD_model.py:176-246, with --no_vgg_loss and the encoder features): returns
    {'G_GAN', 'G_GAN_Feat', 'G_L1', 'D_real', 'D_fake', 'fake'}; `loss_G = G_GAN + G_GAN_Feat + G_L1`,
    `loss_D = (D_fake + D_real) / 2` (train.py:79-80).
    sd*: state_dicts (parameters may be autograd leaves).  batch: dict of label / inst / image / pose / normal tensors in the
    computing dtype.  opt: dict with label_nc, feat_pose_num_bins, n_downsample_global, n_blocks_global, n_downsample_E,
    num_D, n_layers_D, lambda_feat, lambda_L1.
    masks: None, or {'G': [...], 'E': [...], 'D_fake': [[...] per scale], 'D_real': [[...]]} -- ReLU / LeakyReLU patterns of
    another implementation's forward, under which the activations are then evaluated (see global_generator) -- and, optional,
    'L1_sign' (sign of fake - image) / 'Feat_sign' ([[sign of fake feature - real feature per layer] per scale]): |x| is then
    evaluated as sign * x, i.e. the L1 terms take the OTHER implementation's side of their kink as well (one element of
    fake - image within rounding of zero moves d loss / d fake by 2 * lambda_L1 / numel at that pixel, and every generator
    and encoder gradient by ~1 % with it)."""
    masks = masks or {}

    def l1(x, sign):
        return x.abs().mean() if sign is None else (x * sign.to(x.dtype)).mean()
    fs = masks.get('Feat_sign')
    label, ins, image = batch['label'], batch['inst'], batch['image']
    N, _, H, W = label.shape
    dt = image.dtype
    one_hot = torch.zeros(N, opt['label_nc'], H, W, dtype=dt).scatter_(1, label.long(), 1.0)
    edge = torch.zeros(N, 1, H, W, dtype=torch.bool)          # pix2pixHD_model.py:343-349
    edge[:, :, :, 1:] |= ins[:, :, :, 1:] != ins[:, :, :, :-1]
    edge[:, :, :, :-1] |= ins[:, :, :, 1:] != ins[:, :, :, :-1]
    edge[:, :, 1:, :] |= ins[:, :, 1:, :] != ins[:, :, :-1, :]
    edge[:, :, :-1, :] |= ins[:, :, 1:, :] != ins[:, :, :-1, :]
    input_label = torch.cat([one_hot, edge.to(dt)], 1)
    nE = opt['n_downsample_E']
    feats = global_generator(sdE, image, nE, 0, relu_masks=masks.get('E'))
    feat = _instance_mean(feats, ins)
    pose_oh = torch.zeros(N, opt['feat_pose_num_bins'] + 1, H, W, dtype=dt).scatter_(1, batch['pose'].long(), 1.0)
    fake = global_generator(sdG, torch.cat([input_label, feat, pose_oh, batch['normal']], 1), opt['n_downsample_global'],
                            opt['n_blocks_global'], relu_masks=masks.get('G'))
    nD, nl = opt['num_D'], opt['n_layers_D']
    pf = multiscale_discriminator(sdD, torch.cat([input_label, fake], 1), nD, nl, lrelu_masks=masks.get('D_fake'))
    pf_det = multiscale_discriminator(sdD, torch.cat([input_label, fake.detach()], 1), nD, nl, lrelu_masks=masks.get('D_fake'))
    pr = multiscale_discriminator(sdD, torch.cat([input_label, image], 1), nD, nl, lrelu_masks=masks.get('D_real'))

    def mse(t, v):
        return ((t - v) ** 2).mean()
    fw = (4.0 / (nl + 1)) * (1.0 / nD) * opt['lambda_feat']
    return {
        'G_GAN': sum(mse(s[-1], 1.0) for s in pf),
        'G_GAN_Feat': sum(fw * l1(a - b.detach(), None if fs is None else fs[i][j])
                          for i, (sf, sr) in enumerate(zip(pf, pr)) for j, (a, b) in enumerate(zip(sf[:-1], sr[:-1]))),
        'G_L1': l1(fake - image, masks.get('L1_sign')) * opt['lambda_L1'],
        'D_fake': sum(mse(s[-1], 0.0) for s in pf_det),
        'D_real': sum(mse(s[-1], 1.0) for s in pr),
        'fake': fake,
    }


def _instance_mean(out, inst):
    """the pooling half of encoder() on already computed features (differentiable)"""
    N = out.shape[0]
    ids = inst.clone().long()
    for n in range(N):
        ids[n] = ids[n] * N + n
    res = out.clone()
    for i in torch.unique(ids).tolist():
        mask = (ids == i).expand_as(out)
        for c in range(out.shape[1]):
            m = mask[:, c]
            res[:, c][m] = out[:, c][m].mean()
    return res


# ---- VGG19 perceptual loss (networks.py:137-149, 467-497; torchvision 0.2.1 models/vgg.py cfg 'E').  torchvision is not in
# this image: the backbone's LAYOUT is restated here (un-vendored dependency, pinned version from the reference's
# environment.yml), the slicing and the loss are pinned to the reference's own VGGLoss / Vgg19 classes by
# tests/golden/make_vgg_golden.py -> tests/golden/vgg_golden.npz (seeded weights; no pretrained file without a network).
VGG19_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']
VGG19_SLICES = [(0, 2), (2, 7), (7, 12), (12, 21), (21, 30)]   # Vgg19.__init__: relu1_1, relu2_1, relu3_1, relu4_1, relu5_1


def vgg19_layout():
    """index in torchvision's `features` Sequential -> ('conv', cin, cout) | ('relu',) | ('pool',)"""
    out, cin = [], 3
    for v in VGG19_CFG:
        if v == 'M':
            out.append(('pool',))
        else:
            out += [('conv', cin, v), ('relu',)]
            cin = v
    return out


def vgg19_seeded_state(seed, dtype=torch.float32):
    """{'features.<i>.weight' / '.bias'}: He-normal weights, N(0, 0.05) biases drawn from `seed` in layer order (what the
    golden script hands to the reference's Vgg19 through a stub `models.vgg19` and the tests hand to the product)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for i, lay in enumerate(vgg19_layout()):
        if lay[0] == 'conv':
            _, cin, cout = lay
            std = (2.0 / (cin * 9)) ** 0.5
            sd['features.%d.weight' % i] = (torch.randn(cout, cin, 3, 3, generator=g) * std).to(dtype)
            sd['features.%d.bias' % i] = (torch.randn(cout, generator=g) * 0.05).to(dtype)
    return sd


def vgg19_slices(sd, x):
    """the five slice outputs of Vgg19.forward (networks.py:489-497) from a torchvision-keyed state dict"""
    outs, h = [], x
    lay = vgg19_layout()
    for a, b in VGG19_SLICES:
        for i in range(a, b):
            if lay[i][0] == 'conv':
                h = F.conv2d(h, sd['features.%d.weight' % i].to(h.dtype), sd['features.%d.bias' % i].to(h.dtype), padding=1)
            elif lay[i][0] == 'relu':
                h = F.relu(h)
            else:
                h = F.max_pool2d(h, 2, 2)
        outs.append(h)
    return outs


def vgg_loss(sd, x, y):
    """VGGLoss.forward (networks.py:144-149): sum_i w_i * L1(vgg_i(x), vgg_i(y).detach())"""
    w = [1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0]
    fx, fy = vgg19_slices(sd, x), vgg19_slices(sd, y)
    return sum(wi * (a - b.detach()).abs().mean() for wi, a, b in zip(w, fx, fy)), fx
