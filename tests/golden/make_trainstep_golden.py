#!/usr/bin/env python3
"""Generate tests/golden/trainstep_golden.npz: the REFERENCE's textural train step, executed literally on the CPU in fp64.

Runs only in the build container (needs /root/reference).  Everything that computes is the reference's own code, imported
unmodified from where it lies:
  * /root/reference/textural/models/pix2pixHD_model.py  (Pix2PixHDModel.initialize / encode_input / forward, :16-246)
  * /root/reference/textural/models/networks.py         (GlobalGenerator, MultiscaleDiscriminator, Encoder, GANLoss)
  * /root/reference/textural/options/train_options.py   (the option parser and its defaults)
and the loop body below is textural/train.py:69-95 line for line (losses -> loss_D / loss_G -> optimizer_G.zero_grad();
loss_G.backward(); optimizer_G.step(); optimizer_D.zero_grad(); loss_D.backward(); optimizer_D.step()).

Adaptations, none of which touches arithmetic: `torchvision` / `dominate` are stubbed (absent; only Vgg19 and the HTML
visualiser use them), `Tensor.cuda()` / `Module.cuda()` are the identity and `torch.cuda.FloatTensor` allocates on the
CPU (there is no GPU here), numpy's `any` accepts a torch tensor (networks.py:320 under numpy 2), and the default dtype
is float64 so that the result is a yardstick rather than another fp32 rounding of it (a bias in front of an InstanceNorm
has a zero gradient in exact arithmetic; Adam divides by |g| + 1e-8, so fp32 round-off of that zero would be a visible
parameter update).  Initial weights are rounded to fp32 so the product can start from exactly the same values.

Stored for a small configuration of the 3D-SDN option set (--feat_pose --feat_normal --no_vgg_loss, 2-scale D), two
consecutive steps on two different batches: inputs, initial state_dicts, per step the 8 losses, the gradients each
optimizer consumed, the parameter updates (w_after - w_before, fp64 differences stored as fp32) and the InstanceNorm
running statistics after the step.

    python tests/golden/make_trainstep_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('SDN_REFERENCE_ROOT', '/root/reference')

ARGV = ['--name', 'trainstep_golden', '--no_vgg_loss', '--feat_pose', 'x', '--feat_normal', 'x', '--num_D', '2',
        '--label_nc', '5', '--feat_num', '2', '--feat_pose_num_bins', '4', '--ngf', '8', '--n_downsample_global', '2',
        '--n_blocks_global', '2', '--ndf', '8', '--nef', '4', '--n_downsample_E', '2', '--batchSize', '2']
N, H, W = 2, 32, 48
STEPS = 2


class _Stub(types.ModuleType):
    def __getattr__(self, key):
        if key.startswith('__'):
            raise AttributeError(key)
        return type(key, (object,), {'__init__': lambda self, *a, **k: None})


def reference_model(checkpoints_dir):
    """The reference's Pix2PixHDModel on its own networks, CPU, float64."""
    for name in ('torchvision', 'torchvision.models', 'dominate', 'dominate.tags'):
        sys.modules.setdefault(name, _Stub(name))
    sys.path.insert(0, os.path.join(REF, 'textural'))
    torch.set_default_dtype(torch.float64)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.FloatTensor = lambda size: torch.empty(tuple(size), dtype=torch.float64)
    torch.cuda.ByteTensor = lambda size: torch.empty(tuple(size), dtype=torch.uint8)
    from options.train_options import TrainOptions
    old = sys.argv
    sys.argv = ['train.py'] + ARGV + ['--checkpoints_dir', checkpoints_dir]
    try:
        p = TrainOptions()
        p.initialize()
        opt = p.parser.parse_args()
    finally:
        sys.argv = old
    opt.isTrain = True
    opt.gpu_ids = []
    import models.networks as ref_networks
    import models.pix2pixHD_model as ref_model
    assert ref_networks.__file__.startswith(REF) and ref_model.__file__.startswith(REF)

    class _Np:
        def __getattr__(self, name):
            return getattr(np, name)

        @staticmethod
        def any(a, *args, **kw):
            return bool(a.any()) if isinstance(a, torch.Tensor) else np.any(a, *args, **kw)
    ref_networks.np = _Np()
    torch.manual_seed(2024)
    model = ref_model.Pix2PixHDModel()
    model.initialize(opt)
    with torch.no_grad():
        for p_ in model.parameters():
            p_.copy_(p_.float().double())      # exactly representable in the product's fp32 parameters
    return model, opt


def batch(step, opt):
    g = torch.Generator().manual_seed(500 + step)
    label = torch.randint(0, opt.label_nc, (N, 1, H, W), generator=g).double()
    inst = torch.zeros(N, 1, H, W)
    for n in range(N):
        for k in range(4):
            y0, x0 = int(torch.randint(0, H - 8, (1,), generator=g)), int(torch.randint(0, W - 8, (1,), generator=g))
            h, w = int(torch.randint(4, 16, (1,), generator=g)), int(torch.randint(4, 20, (1,), generator=g))
            inst[n, 0, y0:y0 + h, x0:x0 + w] = 1000 * (k + 1) + n
    image = (torch.rand(N, 3, H, W, generator=g) * 2 - 1).float().double()
    pose = torch.randint(0, opt.feat_pose_num_bins + 1, (N, 1, H, W), generator=g).double()
    normal = (torch.rand(N, 3, H, W, generator=g) * 2 - 1).float().double()
    return {'label': label, 'inst': inst, 'image': image, 'pose': pose, 'normal': normal}


def nets(model):
    return (('G', model.netG), ('D', model.netD), ('E', model.netE))


def main():
    import tempfile
    model, opt = reference_model(tempfile.mkdtemp())
    import json
    plain = {k: v for k, v in vars(opt).items() if isinstance(v, (int, float, str, bool, list)) and k != 'checkpoints_dir'}
    out = {'meta/argv': np.array(ARGV), 'meta/shape': np.array([N, H, W]), 'meta/steps': np.array(STEPS),
           'meta/loss_names': np.array(model.loss_names), 'meta/opt_json': np.array(json.dumps(plain, sort_keys=True))}
    for name, net in nets(model):
        for k, v in net.state_dict().items():
            out['init/%s/%s' % (name, k)] = v.detach().numpy().astype(np.float32 if v.is_floating_point() else v.numpy().dtype)
    for step in range(STEPS):
        data = batch(step, opt)
        for k, v in data.items():
            out['step%d/in/%s' % (step, k)] = v.numpy().astype(np.float32)
        before = {name: {k: p.detach().clone() for k, p in net.named_parameters()} for name, net in nets(model)}
        # ---------------- textural/train.py:69-95, literally (pix2pix_model.module == model: no DataParallel on the CPU)
        losses, generated = model(data['label'], data['inst'].clone(), data['image'], None,
                                  data['pose'], data['normal'], None, infer=False)
        losses = [torch.mean(x) if not isinstance(x, int) else x for x in losses]
        loss_dict = dict(zip(model.loss_names, losses))
        loss_D = (loss_dict['D_fake'] + loss_dict['D_real']) * 0.5
        loss_G = loss_dict['G_GAN'] + loss_dict['G_GAN_Feat'] + loss_dict['G_VGG'] + loss_dict['G_L1'] + loss_dict['E_VAE']
        model.optimizer_G.zero_grad()
        loss_G.backward()
        grads_G = {name: {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p))
                          for k, p in net.named_parameters()} for name, net in nets(model) if name != 'D'}
        model.optimizer_G.step()
        model.optimizer_D.zero_grad()
        loss_D.backward()
        grads_D = {k: p.grad.detach().clone() for k, p in model.netD.named_parameters()}
        model.optimizer_D.step()
        # ----------------
        for k, v in loss_dict.items():
            out['step%d/loss/%s' % (step, k)] = np.float64(float(v))
        out['step%d/loss/g_total' % step] = np.float64(float(loss_G))
        out['step%d/loss/d_total' % step] = np.float64(float(loss_D))
        for name in ('G', 'E'):
            for k, g in grads_G[name].items():
                out['step%d/grad/%s/%s' % (step, name, k)] = g.numpy().astype(np.float32)
        for k, g in grads_D.items():
            out['step%d/grad/D/%s' % (step, k)] = g.numpy().astype(np.float32)
        for name, net in nets(model):
            for k, p in net.named_parameters():
                out['step%d/dw/%s/%s' % (step, name, k)] = (p.detach() - before[name][k]).numpy().astype(np.float32)
            for k, v in net.state_dict().items():
                if 'running_' in k:
                    out['step%d/running/%s/%s' % (step, name, k)] = v.numpy().astype(np.float32)
    path = os.environ.get('SDN_TRAINSTEP_GOLDEN_OUT') or os.path.join(HERE, 'trainstep_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote %s: %d arrays, %.1f KiB' % (path, len(out), os.path.getsize(path) / 1024))
    for step in range(STEPS):
        print('step', step, {k.split('/')[-1]: float(out[k]) for k in out if k.startswith('step%d/loss/' % step)})


if __name__ == '__main__':
    main()
