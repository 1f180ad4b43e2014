"""Pix2PixHDModel on the fused HIP networks: the model wrapper the reference's textural scripts drive
(/root/reference/textural/models/pix2pixHD_model.py), same method names and loss list.

What is kept: `initialize(opt)`, `encode_input`, `discriminate`, `forward` -> [[G_GAN, G_GAN_Feat, G_VGG, D_real, D_fake,
G_L1, E_VAE, E_regress], fake_image|None], `fake_inference`, `inference`, `get_edges`, `save`,
`update_learning_rate`, `update_fixed_params`; `optimizer_G` / `optimizer_D` (Adam, lr 2e-4, betas (beta1, 0.999));
the input assembly (one-hot labels + instance edges + encoded features + one-hot pose + normals, :124-166).
What changes: every network is a textural.models.networks module running on libsdn_hip.so; tensors are created on the
input's device instead of through torch.cuda.FloatTensor; `train_step` packages textural/train.py:69-95.

One deliberate saving (results identical): the discriminator pass that scores the fake image for the GENERATOR loss
(:210) runs with the discriminator's parameters detached.  In the reference `loss_G.backward()` also fills the
discriminator's weight gradients, which `optimizer_D.zero_grad()` throws away before `loss_D.backward()`
(train.py:88-95); not computing them changes no parameter update.

A second one (results identical): with the default `pool_size` 0 the image pool returns its input, so the reference's
`discriminate(input_label, fake_image, use_pool=True)` (:192, on fake_image.detach()) and
`netD.forward(cat(input_label, fake_image))` (:210) evaluate the same discriminator on the same values.  They run as one
forward pass with two autograd views (`MultiscaleDiscriminator.forward_dual`): the discriminator's loss back-propagates
into the weights through the first, the generator's loss into fake_image through the second.  With a non-empty pool the
two passes stay separate, as in the reference.
"""
import os
from types import SimpleNamespace

import numpy as np
import torch

from . import networks
from .base_model import BaseModel


def default_options(**overrides):
    """The option values Pix2PixHDModel reads, with the reference's defaults (textural/options/base_options.py:28-88,
    train_options.py:10-42)."""
    opt = SimpleNamespace(
        name='baseline', gpu_ids=[0], checkpoints_dir='./checkpoints', model='pix2pixHD', norm='instance',
        batchSize=1, label_nc=14, output_nc=3, resize_or_crop='scale_width_and_crop',
        netG='global', ngf=64, n_downsample_global=4, n_blocks_global=9, n_blocks_local=3, n_local_enhancers=0,
        niter_fix_global=0, no_global_encoder=1, global_encoder_nz=3,
        no_instance=False, instance_feat=True, label_feat=False, feat_num=5, load_features=False, n_downsample_E=4,
        nef=16, feat_pose='', feat_pose_num_bins=24, feat_normal='', feat_depth='',
        isTrain=True, continue_train=False, load_pretrain='', which_epoch='latest', niter=100, niter_decay=100,
        beta1=0.5, lr=0.0002, num_D=2, n_layers_D=3, ndf=64, lambda_feat=5.0, no_ganFeat_loss=False,
        no_vgg_loss=False, no_lsgan=False, pool_size=0, lambda_L1=10.0, lambda_KL=0.01, verbose=False)
    for k, v in overrides.items():
        setattr(opt, k, v)
    return opt


def _adam(params, lr, beta1):
    """torch.optim.Adam(lr, betas=(beta1, 0.999)) as the reference builds it (pix2pixHD_model.py:113-117); on the GPU the
    single-kernel (`fused`) implementation of the same update: 190 M parameters are one pass instead of ten."""
    if params and all(p.is_cuda for p in params):
        try:
            return torch.optim.Adam(params, lr=lr, betas=(beta1, 0.999), fused=True)
        except (TypeError, RuntimeError):
            pass
    return torch.optim.Adam(params, lr=lr, betas=(beta1, 0.999))


class ImagePool:
    """History buffer of generated images (util/image_pool.py); pool_size 0 (the default) passes images through."""

    def __init__(self, pool_size):
        self.pool_size = pool_size
        self.images = []

    def query(self, images):
        if self.pool_size == 0:
            return images
        out = []
        for img in images:
            img = img.unsqueeze(0)
            if len(self.images) < self.pool_size:
                self.images.append(img)
                out.append(img)
            elif np.random.uniform(0, 1) > 0.5:
                i = np.random.randint(0, self.pool_size)
                out.append(self.images[i].clone())
                self.images[i] = img
            else:
                out.append(img)
        return torch.cat(out, 0)


class Pix2PixHDModel(BaseModel):
    def name(self):
        return 'Pix2PixHDModel'

    # ------------------------------------------------------------------------------------------------ construction
    def initialize(self, opt):
        BaseModel.initialize(self, opt)
        self.isTrain = opt.isTrain
        self.use_features = bool(opt.instance_feat or opt.label_feat)
        self.gen_features = self.use_features and not opt.load_features
        self.no_global_encoder = opt.no_global_encoder
        if not opt.no_global_encoder:
            raise NotImplementedError('global encoder (the reference never defines netGlobalE either)')
        input_nc = opt.label_nc if opt.label_nc != 0 else 3
        g_in = input_nc + (0 if opt.no_instance else 1)
        if self.use_features:
            g_in += opt.feat_num
        if opt.feat_pose:
            g_in += opt.feat_pose_num_bins + 1 if opt.feat_pose_num_bins else 2
        if opt.feat_normal:
            g_in += 3
        if opt.feat_depth:
            g_in += 1
        self.netG_input_nc = g_in
        self.model_names = ['G']
        self.netG = networks.define_G(g_in, opt.output_nc, opt.ngf, opt.netG, opt.n_downsample_global,
                                      opt.n_blocks_global, opt.n_local_enhancers, opt.n_blocks_local, opt.norm,
                                      gpu_ids=self.gpu_ids)
        if self.isTrain:
            d_in = input_nc + opt.output_nc + (0 if opt.no_instance else 1)
            self.netD = networks.define_D(d_in, opt.ndf, opt.n_layers_D, opt.norm, opt.no_lsgan, opt.num_D,
                                          not opt.no_ganFeat_loss, gpu_ids=self.gpu_ids)
            self.model_names.append('D')
        if self.gen_features:
            self.netE = networks.define_G(opt.output_nc, opt.feat_num, opt.nef, 'encoder', opt.n_downsample_E,
                                          norm=opt.norm, gpu_ids=self.gpu_ids, isTrain=opt.isTrain)
            self.model_names.append('E')
        if getattr(opt, 'verbose', True):   # the reference always prints (pix2pixHD_model.py:66); default_options() turns it off
            self.print_networks(True)
        if not self.isTrain or opt.continue_train or opt.load_pretrain:
            path = '' if not self.isTrain else opt.load_pretrain
            self.load_network(self.netG, 'G', opt.which_epoch, path)
            if self.isTrain:
                self.load_network(self.netD, 'D', opt.which_epoch, path)
            if self.gen_features:
                self.load_network(self.netE, 'E', opt.which_epoch, path)
        if self.isTrain:
            if opt.pool_size > 0 and len(self.gpu_ids) > 1:
                raise NotImplementedError('Fake Pool Not Implemented for MultiGPU')
            self.fake_pool = ImagePool(opt.pool_size)
            self.old_lr = opt.lr
            self.criterionGAN = networks.GANLoss(use_lsgan=not opt.no_lsgan, tensor=self.Tensor)
            self.criterionFeat = networks.L1Loss()   # torch.nn.L1Loss() semantics, fused kernels for dense GPU pairs
            if not opt.no_vgg_loss:
                self.criterionVGG = networks.VGGLoss(self.gpu_ids)
            self.loss_names = ['G_GAN', 'G_GAN_Feat', 'G_VGG', 'D_real', 'D_fake', 'G_L1', 'E_VAE', 'E_regress']
            if opt.niter_fix_global > 0:
                params = []
                for key, value in dict(self.netG.named_parameters()).items():
                    lr = opt.lr if key.startswith('model' + str(opt.n_local_enhancers)) else 0.0
                    params.append({'params': [value], 'lr': lr})
            else:
                params = list(self.netG.parameters())
            if self.gen_features:
                params += list(self.netE.parameters())
            self.optimizer_G = _adam(params, opt.lr, opt.beta1)
            self.optimizer_D = _adam(list(self.netD.parameters()), opt.lr, opt.beta1)

    # ------------------------------------------------------------------------------------------------ input assembly
    def _device(self):
        return next(self.netG.parameters()).device

    def _one_hot(self, index_map, channels):
        n, _, h, w = index_map.shape
        out = torch.zeros(n, channels, h, w, dtype=torch.float32, device=index_map.device)
        return out.scatter_(1, index_map.long(), 1.0)

    def get_edges(self, t):
        """1 where an instance id differs from a 4-neighbour (pix2pixHD_model.py:343-349)."""
        edge = torch.zeros(t.shape, dtype=torch.bool, device=t.device)
        dx = t[:, :, :, 1:] != t[:, :, :, :-1]
        dy = t[:, :, 1:, :] != t[:, :, :-1, :]
        edge[:, :, :, 1:] |= dx
        edge[:, :, :, :-1] |= dx
        edge[:, :, 1:, :] |= dy
        edge[:, :, :-1, :] |= dy
        return edge.float()

    def encode_input(self, label_map, inst_map=None, real_image=None, feat_map=None, pose_map=None, normal_map=None,
                     depth_map=None, infer=False):
        dev = self._device()
        opt = self.opt
        with torch.no_grad():
            if opt.label_nc == 0:
                input_label = label_map.detach().to(dev)
            else:
                input_label = self._one_hot(label_map.detach().to(dev), opt.label_nc)
            if not opt.no_instance:
                inst_map = inst_map.detach().to(dev)
                input_label = torch.cat((input_label, self.get_edges(inst_map)), dim=1)
            if real_image is not None:
                real_image = real_image.detach().to(dev)
            if self.use_features:
                if opt.load_features:
                    feat_map = feat_map.detach().to(dev)
                if opt.feat_pose:
                    pose_map = pose_map.detach().to(dev)
                    if opt.feat_pose_num_bins:
                        pose_map = self._one_hot(pose_map, opt.feat_pose_num_bins + 1)
                if opt.feat_normal:
                    normal_map = normal_map.detach().to(dev)
                if opt.feat_depth:
                    depth_map = depth_map.detach().to(dev)
        return input_label, inst_map, real_image, feat_map, pose_map, normal_map, depth_map

    def _generator_input(self, input_label, feat_map, pose_map, normal_map, depth_map):
        parts = [input_label]
        if self.use_features:
            parts.append(feat_map)
            if self.opt.feat_pose:
                parts.append(pose_map)
            if self.opt.feat_normal:
                parts.append(normal_map)
            if self.opt.feat_depth:
                parts.append(depth_map)
        if getattr(self.netG, 'accepts_parts', False):
            return parts  # written side by side by the conv executor; gradients only for the parts that need them
        return torch.cat(parts, dim=1)

    def discriminate(self, input_label, test_image, use_pool=False):
        if getattr(self.netD, 'accepts_parts', False) and (not use_pool or self.fake_pool.pool_size == 0):
            # the parts go side by side into the discriminator's channels-last input buffer (no NCHW concatenation)
            return self.netD.forward([input_label, test_image.detach()])
        x = torch.cat((input_label, test_image.detach()), dim=1)
        if use_pool:
            x = self.fake_pool.query(x)
        return self.netD.forward(x)

    # ------------------------------------------------------------------------------------------------ training forward
    def forward(self, label, inst, image, feat, pose=None, normal=None, depth=None, infer=False):
        opt = self.opt
        input_label, inst_map, real_image, feat_map, pose_map, normal_map, depth_map = self.encode_input(
            label, inst, image, feat, pose, normal, depth)
        loss_E_VAE = 0
        if self.use_features and not opt.load_features:
            feat_map, loss_E_VAE = self.netE.forward(real_image, inst_map)
        fake_image = self.netG.forward(self._generator_input(input_label, feat_map, pose_map, normal_map, depth_map))

        shared = self.fake_pool.pool_size == 0 and hasattr(self.netD, 'forward_dual')
        if shared:
            # the pool passes the fake image through, so :192 and :210 score the same tensor with the same weights:
            # one pass, two autograd views (see the module docstring)
            pred_fake_pool, pred_fake, second_fake_pass = self.netD.forward_dual([input_label, fake_image])
        else:
            pred_fake_pool = self.discriminate(input_label, fake_image, use_pool=True)
        loss_D_fake = self.criterionGAN(pred_fake_pool, False)
        pred_real = self.discriminate(input_label, real_image)
        loss_D_real = self.criterionGAN(pred_real, True)
        # generator's view of the discriminator: gradients flow to fake_image only (see the module docstring)
        if shared:
            second_fake_pass()   # what :210's forward does to the InstanceNorm running statistics, in the reference's order
        elif getattr(self.netD, 'accepts_parts', False):
            pred_fake = self.netD.forward([input_label, fake_image], detach_weights=True)
        else:
            pred_fake = self.netD.forward(torch.cat((input_label, fake_image), dim=1), detach_weights=True)
        loss_G_GAN = self.criterionGAN(pred_fake, True)

        loss_G_GAN_Feat = 0
        if not opt.no_ganFeat_loss:
            wgt = (4.0 / (opt.n_layers_D + 1)) * (1.0 / opt.num_D) * opt.lambda_feat
            for i in range(opt.num_D):
                for j in range(len(pred_fake[i]) - 1):
                    loss_G_GAN_Feat = loss_G_GAN_Feat + wgt * self.criterionFeat(pred_fake[i][j], pred_real[i][j].detach())
        loss_G_VGG = 0
        if not opt.no_vgg_loss:
            loss_G_VGG = self.criterionVGG(fake_image, real_image) * opt.lambda_feat
        loss_G_L1 = 0
        if opt.lambda_L1 > 0:
            loss_G_L1 = self.criterionFeat(fake_image, real_image) * opt.lambda_L1
        loss_E_VAE = loss_E_VAE * opt.lambda_KL
        if isinstance(loss_E_VAE, float):
            loss_E_VAE = 0
        return [[loss_G_GAN, loss_G_GAN_Feat, loss_G_VGG, loss_D_real, loss_D_fake, loss_G_L1, loss_E_VAE, 0],
                fake_image if infer else None]

    __call__ = forward

    def train_step(self, label, inst, image, feat=None, pose=None, normal=None, depth=None):
        """textural/train.py:69-95: losses, generator update, discriminator update.  Returns the loss dict."""
        losses, _ = self.forward(label, inst, image, feat, pose, normal, depth)
        d = dict(zip(self.loss_names, [x if isinstance(x, int) else torch.mean(x) for x in losses]))
        loss_D = (d['D_fake'] + d['D_real']) * 0.5
        loss_G = d['G_GAN'] + d.get('G_GAN_Feat', 0) + d.get('G_VGG', 0) + d.get('G_L1', 0) + d.get('E_VAE', 0)
        self.optimizer_G.zero_grad()
        loss_G.backward()
        side = self._update_stream(loss_G.device)
        if side is None:
            self.optimizer_G.step()
            self.optimizer_D.zero_grad()
            loss_D.backward()
            self.optimizer_D.step()
            return d
        # r06: the generator / encoder update runs BESIDE the discriminator's backward pass.  loss_D was built from fake.detach()
        # (pix2pixHD_model.py:192), so its backward pass reads and writes nothing the generator's optimizer touches: Adam over
        # 190 M parameters and the re-pack of their bf16 (hi, lo) operand copies (~7 GB of HBM traffic, 3-4 ms at batch 4) used
        # to sit between the two backward passes and in front of the next forward pass; on a side stream they hide behind the
        # MFMA-bound discriminator pass.  Same arithmetic on the same values: the updates are those of train.py:88-95.
        # SDN_UPDATE_STREAM=0 restores the serial order.
        cur = torch.cuda.current_stream(loss_G.device)
        side.wait_stream(cur)                       # the generator's gradients are final
        with torch.cuda.stream(side):
            self.optimizer_G.step()
            from sdn_hip import conv as _hc
            _hc.eager_repack([self.netG] + ([self.netE] if self.gen_features else []))
        self.optimizer_D.zero_grad()
        loss_D.backward()
        self.optimizer_D.step()
        cur.wait_stream(side)                       # (also orders every later free of a gradient behind the side stream's reads)
        return d

    _update_streams = {}

    def _update_stream(self, dev):
        """the side stream of train_step's generator update (one per device and calling stream), None when switched off or on
        the CPU"""
        if dev.type != 'cuda' or os.environ.get('SDN_UPDATE_STREAM', '1') == '0':
            return None
        key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
        st = Pix2PixHDModel._update_streams.get(key)
        if st is None:
            st = Pix2PixHDModel._update_streams[key] = torch.cuda.Stream(device=dev)
        return st

    # ------------------------------------------------------------------------------------------------ inference
    def fake_inference(self, image, label, inst, feat=None, pose=None, normal=None, depth=None):
        input_label, inst_map, real_image, _, pose_map, normal_map, depth_map = self.encode_input(
            label, inst_map=inst, pose_map=pose, normal_map=normal, depth_map=depth, real_image=image, infer=True)
        with torch.no_grad():
            feat_map = None
            if self.use_features:
                feat_map = self.netE.forward(real_image, inst_map) if feat is None else feat.to(self._device())
                if isinstance(feat_map, tuple):
                    feat_map = feat_map[0]
            return self.netG.forward(self._generator_input(input_label, feat_map, pose_map, normal_map, depth_map))

    def inference(self, label, inst):
        input_label, inst_map, _, _, _, _, _ = self.encode_input(label, inst, infer=True)
        with torch.no_grad():
            x = input_label
            if self.use_features:
                x = torch.cat((input_label, self.sample_features(inst_map)), dim=1)
            return self.netG.forward(x)

    def sample_features(self, inst):
        """pix2pixHD_model.py:298-316: draw one stored feature cluster per instance."""
        path = os.path.join(self.opt.checkpoints_dir, self.opt.name, self.opt.cluster_path)
        clusters = np.load(path, allow_pickle=True).item()
        feat_map = torch.zeros(1, self.opt.feat_num, inst.shape[2], inst.shape[3], device=inst.device)
        for i in np.unique(inst.cpu().numpy().astype(int)):
            label = i if i < 5000 else i // 5000
            if label in clusters:
                feat = clusters[label]
                row = feat[np.random.randint(0, feat.shape[0])]
                mask = (inst[0, 0] == int(i))
                for k in range(self.opt.feat_num):
                    feat_map[0, k][mask] = float(row[k])
        return feat_map

    # ------------------------------------------------------------------------------------------------ bookkeeping
    def save(self, which_epoch):
        self.save_network(self.netG, 'G', which_epoch, self.gpu_ids)
        self.save_network(self.netD, 'D', which_epoch, self.gpu_ids)
        if self.gen_features:
            self.save_network(self.netE, 'E', which_epoch, self.gpu_ids)

    def update_fixed_params(self):
        params = list(self.netG.parameters())
        if self.gen_features:
            params += list(self.netE.parameters())
        self.optimizer_G = _adam(params, self.opt.lr, self.opt.beta1)

    def update_learning_rate(self):
        lr = self.old_lr - self.opt.lr / self.opt.niter_decay
        for group in self.optimizer_D.param_groups + self.optimizer_G.param_groups:
            group['lr'] = lr
        self.old_lr = lr
