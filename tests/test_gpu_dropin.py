"""The reference's caller loops on cuda:0 through the product's public surface only (the GPU-side half of
tests/test_dropin_reference_callers.py, which imports the reference's own files in the build container; they do not
travel to the GPU box).  Each statement below is the caller's statement, with `.item()` where the reference wrote
`.data[0]` (torch 0.4 idiom that current torch rejects for 0-dim tensors -- the one edit a maintainer has to make to
textural/train.py regardless of the backend).

  test_textural_train_loop_body   textural/train.py:47-48, 69-95 on create_model(opt) -> nn.DataParallel
  test_geometric_step_batch       geometric/scripts/main.py:114-154 (BaseNet.step_batch) on DataParallel(Model()).cuda()
  test_geometric_test_time_loop   geometric/scripts/main.py:402-456 (_test: encoder, then Adam over the blob)
"""
import numpy as np
import pytest
import torch
from torch.nn import functional as F
from torch.nn.parallel import DataParallel

from sdn_hip import synth

pytestmark = pytest.mark.gpu


def _batch(n, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    label = torch.randint(1, 14, (n, 1, h, w), generator=g).float()
    inst = torch.zeros(n, 1, h, w)
    pose = torch.zeros(n, 1, h, w)
    for b in range(n):
        for k in range(4):
            y0, x0 = int(torch.randint(0, h - 24, (1,), generator=g)), int(torch.randint(0, w - 40, (1,), generator=g))
            inst[b, 0, y0:y0 + 20, x0:x0 + 36] = 1000 * (k + 1)
            pose[b, 0, y0:y0 + 20, x0:x0 + 36] = int(torch.randint(1, 25, (1,), generator=g))
    image = torch.rand(n, 3, h, w, generator=g) * 2 - 1
    normal = torch.rand(n, 3, h, w, generator=g) * 2 - 1
    return {'label': label, 'inst': inst, 'image': image, 'feat': 0, 'pose': pose, 'normal': normal, 'depth': 0}


def test_textural_train_loop_body():
    from models.models import create_model as create_pix2pix_model
    from models.pix2pixHD_model import default_options
    opt = default_options(gpu_ids=[0], batchSize=2, num_D=2, feat_pose='x', feat_normal='x', no_vgg_loss=True,
                          isTrain=True, ngf=16, n_blocks_global=2, n_downsample_global=2, ndf=16)
    torch.manual_seed(5)
    pix2pix_model = create_pix2pix_model(opt)                       # train.py:47
    assert isinstance(pix2pix_model, DataParallel)
    # weights only: a bias in front of InstanceNorm has an exactly zero gradient
    before = [p.detach().clone() for p in pix2pix_model.module.netG.parameters() if p.dim() > 1][:3]
    before_d = [p.detach().clone() for p in pix2pix_model.module.netD.parameters() if p.dim() > 1][:3]
    losses_G, losses_D = [], []
    for i in range(3):
        data = _batch(2, 64, 96, 10 + i)
        save_fake = i == 2
        # Forward Pass (train.py:69-71)
        losses, generated = pix2pix_model(data['label'], data['inst'], data['image'], data['feat'], data['pose'],
                                          data['normal'], data['depth'], infer=save_fake)
        # sum per device losses (:74-75)
        losses = [torch.mean(x) if not isinstance(x, int) else x for x in losses]
        loss_dict = dict(zip(pix2pix_model.module.loss_names, losses))
        # calculate final loss scalar (:79-80)
        loss_D = (loss_dict['D_fake'] + loss_dict['D_real']) * 0.5
        loss_G = loss_dict['G_GAN'] + loss_dict['G_GAN_Feat'] + loss_dict['G_VGG'] + loss_dict['G_L1'] + loss_dict['E_VAE']
        losses_D.append(loss_D.item())
        losses_G.append(loss_G.item())
        # Backward Pass (:87-95)
        pix2pix_model.module.optimizer_G.zero_grad()
        loss_G.backward()
        pix2pix_model.module.optimizer_G.step()
        pix2pix_model.module.optimizer_D.zero_grad()
        loss_D.backward()
        pix2pix_model.module.optimizer_D.step()
        assert (generated is not None) == save_fake
        if save_fake:
            assert generated.shape == (2, 3, 64, 96) and float(generated.abs().max()) <= 1.0
    assert all(np.isfinite(losses_G)) and all(np.isfinite(losses_D))
    after = [p.detach() for p in pix2pix_model.module.netG.parameters() if p.dim() > 1][:3]
    after_d = [p.detach() for p in pix2pix_model.module.netD.parameters() if p.dim() > 1][:3]
    assert all(not torch.equal(a, b) for a, b in zip(before, after))
    assert all(not torch.equal(a, b) for a, b in zip(before_d, after_d))
    pix2pix_model.module.update_learning_rate()                      # train.py:143-144


def _geometric_model(render_size=64):
    from derender3d import TargetType
    from derender3d.models import Derenderer3d, ShapenetObj
    objs = []
    for k in range(8):
        v, f = synth.car_like(2500, seed=40 + k)
        objs.append(ShapenetObj(vertices=v[:, [2, 1, 0]] * np.asarray([-1, 1, 1], np.float32), faces=f))
    torch.manual_seed(2)
    return Derenderer3d(mode=TargetType.extend, image_size=256, render_size=render_size, objs=objs), TargetType


def _frame(n, seed):
    rng = np.random.default_rng(seed)
    images = torch.tensor(rng.normal(size=(n, 3, 224, 224)).astype(np.float32)).cuda()
    c = rng.uniform(-0.2, 0.2, (n, 2))
    h, w = rng.uniform(40, 150, n) / 725.0, rng.uniform(60, 300, n) / 725.0
    rois = np.stack([c[:, 0] - h / 2, c[:, 1] - w / 2, c[:, 0] + h / 2, c[:, 1] + w / 2], 1).astype(np.float32)
    return images, torch.tensor(rois).cuda(), torch.full((n, 1), 725.0).cuda()


def _pad_like(x, ref, mode='constant'):
    """derender3d/datasets.py:26-33 Transforms.pad_like"""
    ph, pw = ref.shape[2] - x.shape[2], ref.shape[3] - x.shape[3]
    return F.pad(x, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2), mode=mode)


def test_geometric_step_batch():
    net, TargetType = _geometric_model()
    model = DataParallel(net).cuda()                                 # main.py:182
    model.train()
    mode, mask_weight, ffd_coeff_reg = TargetType.extend, 0.1, 1.0   # FLAGS
    n = 4
    images, roi_norms, focals = _frame(n, 3)
    targets = torch.full((n,), TargetType.full, dtype=torch.uint8).cuda()
    _blob = model(images, roi_norms, focals)                         # :115
    loss_dict = {}
    g = lambda k: _blob[k]
    if mode & TargetType.geometry:                                    # :119-135
        thetas = torch.zeros(n, 1).cuda()
        theta_deltas = torch.cat([torch.cos(thetas), torch.sin(thetas)], dim=1)
        loss_dict.update({
            'theta_delta_loss': F.mse_loss(g('_theta_deltas'), theta_deltas),
            'translation2d_loss': F.mse_loss(g('_translation2ds'), torch.zeros(n, 2).cuda()),
            'scale_loss': F.mse_loss(g('_log_scales'), torch.zeros(n, 3).cuda()),
            'depth_loss': F.mse_loss(g('_log_depths'), torch.ones(n, 1).cuda()),
        })
    if mode & TargetType.reproject:                                   # :137-152
        masks = _pad_like((torch.rand(n, 1, 48, 48) > 0.5).float().cuda(), g('_masks'))
        ignores = _pad_like(torch.zeros(n, 1, 48, 48).cuda(), g('_masks'), mode='replicate')
        mask_losses = (1 - ignores) * F.mse_loss(g('_masks'), masks, reduction='none')
        mask_losses = mask_weight * mask_losses.mean(dim=3).mean(dim=2).mean(dim=1)
        loss_dict.update({
            'class_reward': torch.mean(g('_class_log_probs') * mask_losses.detach()),
            'mask_loss': torch.mean(mask_losses),
            'ffd_coeff_reg': ffd_coeff_reg * torch.mean(g('_ffd_coeffs') ** 2),
        })
    loss = sum(loss_dict.values())
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-3)   # :188
    opt.zero_grad()
    loss.backward()
    opt.step()
    assert all(torch.isfinite(v) for v in loss_dict.values())
    assert torch.isfinite(model.module.derenderer.net.conv1.weight.grad).all()
    assert float(model.module.derenderer._fc3.weight.grad.abs().max()) > 0


def test_geometric_test_time_loop():
    net, TargetType = _geometric_model()
    model = DataParallel(net).cuda()
    model.eval()
    n, num_opts = 3, 6
    rgbs, roi_norms, focals = _frame(n, 4)
    masks = torch.zeros(n, 1, 64, 64).cuda()
    masks[:, :, 16:48, 8:56] = 1
    _mroi_norms = (roi_norms[:, 2:4] + roi_norms[:, 0:2]) / 2
    _droi_norms = roi_norms[:, 2:4] - roi_norms[:, 0:2]
    _blob = {'_roi_norms': roi_norms, '_mroi_norms': _mroi_norms, '_droi_norms': _droi_norms, '_focals': focals}
    with torch.no_grad():
        _blob_derendered = model.module.derenderer(rgbs, _mroi_norms, _droi_norms)   # :402
    _blob.update(_blob_derendered)
    model.train()                                                     # :422-423
    model.module._force_no_sample = True
    for (key, value) in _blob_derendered.items():                     # :425-427
        _blob[key] = value.clone().detach()
    _blob_derendered_optimize = {}
    for key in ['_theta_deltas', '_translation2ds', '_log_scales', '_ffd_coeffs']:   # :429-436
        _blob_derendered_optimize[key] = _blob[key].requires_grad_()
    optimizer = torch.optim.Adam(_blob_derendered_optimize.values(), lr=3e-2)
    history = []
    for num_opt in range(num_opts):                                   # :439-456
        optimizer.zero_grad()
        _blob.update(model.module.render(_blob))
        _masks = _blob['_masks']
        masks_padded = _pad_like(masks, _masks)
        loss = F.mse_loss(_masks, masks_padded, reduction='none') + 100 * torch.mean(_blob['_ffd_coeffs'] ** 2)
        loss = torch.mean(loss)
        loss.backward()
        optimizer.step()
        history.append(loss.item())
    model.eval()
    model.module._force_no_sample = False
    assert all(np.isfinite(history)) and min(history[2:]) < history[0]
    for k in ('_masks', '_normals', '_depth_maps', '_zooms', '_alphas', '_translations'):
        assert k in _blob and torch.isfinite(_blob[k]).all()


def test_checkpoint_round_trip_and_test_mode_inference(tmp_path):
    """textural/train.py:115-134 + test.py / edit_vkitti.py: `save('latest')` writes {epoch}_net_{G,D,E}.pth with the
    reference's keys; a TEST-mode model (isTrain False) loads them (base_model.py:55-95) and reproduces the training
    model's `fake_inference` bit for bit; `inference` draws instance features from the clustered-feature file
    (pix2pixHD_model.py:298-316); `update_fixed_params` (train.py:139-140) rebuilds the generator optimizer."""
    from models.pix2pixHD_model import Pix2PixHDModel, default_options
    common = dict(gpu_ids=[0], batchSize=1, feat_pose='x', feat_normal='x', no_vgg_loss=True, ngf=16, n_blocks_global=2,
                  n_downsample_global=2, ndf=16, nef=8, n_downsample_E=2, checkpoints_dir=str(tmp_path), name='ckpt')
    torch.manual_seed(21)
    train = Pix2PixHDModel()
    train.initialize(default_options(isTrain=True, num_D=2, **common))
    d = _batch(1, 32, 64, 5)
    train.train_step(d['label'], d['inst'].clone(), d['image'], None, d['pose'], d['normal'])   # weights move off init
    train.save('latest')
    for name in ('G', 'D', 'E'):
        sd = torch.load(str(tmp_path / 'ckpt' / ('latest_net_%s.pth' % name)))
        assert list(sd.keys()) == list(getattr(train, 'net' + name).state_dict().keys())
        assert all(not v.is_cuda for v in sd.values())
    test = Pix2PixHDModel()
    test.initialize(default_options(isTrain=False, which_epoch='latest', **common))
    assert not hasattr(test, 'netD') and not hasattr(test, 'optimizer_G')
    a = train.fake_inference(d['image'], d['label'], d['inst'].clone(), pose=d['pose'], normal=d['normal'])
    b = test.fake_inference(d['image'], d['label'], d['inst'].clone(), pose=d['pose'], normal=d['normal'])
    # same weights, same kernels: the atomic split-K sums are the only source of differences (two calls of the SAME model
    # differ by ~2e-5 on this tanh output); before the r02 fix of the packed-weight cache this was 0.47
    assert a.shape == b.shape == (1, 3, 32, 64) and float((a - b).abs().max()) <= 1e-4

    # inference() with sampled instance features: a model without pose / normal inputs, clusters stored as the
    # reference's encode_features.py writes them (dict: label -> [k, feat_num] array)
    plain = dict(common, feat_pose='', feat_normal='')
    torch.manual_seed(22)
    t2 = Pix2PixHDModel()
    t2.initialize(default_options(isTrain=True, num_D=2, **plain))
    t2.save('latest')
    t3 = Pix2PixHDModel()
    opt3 = default_options(isTrain=False, which_epoch='latest', **plain)
    opt3.cluster_path = 'features_clustered_010.npy'
    t3.initialize(opt3)
    rng = np.random.default_rng(3)
    clusters = {int(k): rng.normal(size=(10, opt3.feat_num)).astype(np.float32) for k in range(1, 15)}
    np.save(str(tmp_path / 'ckpt' / opt3.cluster_path), clusters, allow_pickle=True)
    out = t3.inference(d['label'], d['label'].clone())           # instance map = label map (no instances)
    assert out.shape == (1, 3, 32, 64) and torch.isfinite(out).all() and float(out.abs().max()) <= 1.0

    # train.py:139-140
    before = train.optimizer_G
    train.update_fixed_params()
    assert train.optimizer_G is not before
    n_params = sum(len(g['params']) for g in train.optimizer_G.param_groups)
    assert n_params == len(list(train.netG.parameters())) + len(list(train.netE.parameters()))
