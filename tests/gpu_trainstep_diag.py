"""Diagnostic (not a test): gradients of the individual generator-loss terms of Pix2PixHDModel.forward on cuda:0 against the
fp64 oracle, on the first batch of tests/golden/trainstep_golden.npz."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.environ.get('SDN_PKG_ROOT', ROOT)
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(PKG, '3d-sdn_amd'), os.path.join(PKG, '3d-sdn_amd', 'textural')):
    sys.path.insert(0, p)
os.environ.setdefault('SDN_DETERMINISTIC', '1')
os.environ.setdefault('SDN_ALLOW_RANDOM_INIT', '1')
from oracle import textural_oracle as to  # noqa: E402
from test_gpu_trainstep import GOLD, _model  # noqa: E402
for p in (os.path.join(PKG, '3d-sdn_amd', 'textural'), os.path.join(PKG, '3d-sdn_amd')):   # (the import above put ROOT's first)
    sys.path.insert(0, p)

z = np.load(GOLD)
opt = json.loads(str(z['meta/opt_json']))


def oracle_grads():
    def sd(n):
        return {k[len('init/%s/' % n):]: (torch.from_numpy(z[k]).double() if z[k].dtype.kind == 'f' else torch.from_numpy(z[k]))
                for k in z.files if k.startswith('init/%s/' % n)}
    ps = {}

    def leaves(s, tag):
        out = dict(s)
        for k, v in s.items():
            if k.endswith('weight') or k.endswith('bias'):
                out[k] = v.clone().requires_grad_(True)
                ps[tag + '/' + k] = out[k]
        return out
    G, D, E = leaves(sd('G'), 'G'), leaves(sd('D'), 'D'), leaves(sd('E'), 'E')
    d = {k: torch.from_numpy(z['step0/in/' + k]).double() for k in ('label', 'inst', 'image', 'pose', 'normal')}
    N, _, H, W = d['label'].shape
    one_hot = torch.zeros(N, opt['label_nc'], H, W, dtype=torch.float64).scatter_(1, d['label'].long(), 1.0)
    ins = d['inst']
    edge = torch.zeros(N, 1, H, W, dtype=torch.bool)
    edge[:, :, :, 1:] |= ins[:, :, :, 1:] != ins[:, :, :, :-1]
    edge[:, :, :, :-1] |= ins[:, :, :, 1:] != ins[:, :, :, :-1]
    edge[:, :, 1:, :] |= ins[:, :, 1:, :] != ins[:, :, :-1, :]
    edge[:, :, :-1, :] |= ins[:, :, 1:, :] != ins[:, :, :-1, :]
    input_label = torch.cat([one_hot, edge.double()], 1)
    feat = to.encoder(E, d['image'], ins, opt['n_downsample_E'])
    pose_oh = torch.zeros(N, opt['feat_pose_num_bins'] + 1, H, W, dtype=torch.float64).scatter_(1, d['pose'].long(), 1.0)
    fake = to.global_generator(G, torch.cat([input_label, feat, pose_oh, d['normal']], 1), opt['n_downsample_global'],
                               opt['n_blocks_global'])
    fake.retain_grad()
    pf = to.multiscale_discriminator(D, torch.cat([input_label, fake], 1), opt['num_D'])
    pr = to.multiscale_discriminator(D, torch.cat([input_label, d['image']], 1), opt['num_D'])
    mse = lambda t, v: ((t - v) ** 2).mean()   # noqa: E731
    fw = (4.0 / (opt['n_layers_D'] + 1)) * (1.0 / opt['num_D']) * opt['lambda_feat']
    terms = {'G_GAN': sum(mse(s[-1], 1.0) for s in pf),
             'G_GAN_Feat': sum(fw * (a - b.detach()).abs().mean() for sf, sr in zip(pf, pr) for a, b in zip(sf[:-1], sr[:-1])),
             'G_L1': (fake - d['image']).abs().mean() * opt['lambda_L1']}
    out = {}
    for name, t in terms.items():
        for p in ps.values():
            p.grad = None
        fake.grad = None
        t.backward(retain_graph=True)
        out[name] = ({k: p.grad.clone() for k, p in ps.items() if p.grad is not None and k[0] in 'GE'}, fake.grad.clone())
    out['_inputs'] = (input_label.detach(), fake.detach(), d['image'])
    return out


def d_only(ref):
    """the discriminator's image gradient in isolation, four ways"""
    import tempfile
    import sdn_hip
    print('package:', sdn_hip.__file__)
    lab, fake, real = [t.float().cuda() for t in ref['_inputs']]
    fref = ref['G_GAN'][1]

    def gan(res):
        return sum(((s[-1] - 1.0) ** 2).mean() for s in res)

    def report(tag, g):
        g = g.double().cpu()
        print('   D-only %-34s rel %.3e cosine %.6f' % (tag, float((g - fref).norm() / fref.norm()),
                                                       float((g * fref).sum() / (g.norm() * fref.norm()))))
    m, _ = _model(z, tempfile.mkdtemp())
    D = m.netD
    x = torch.cat((lab, fake), 1).requires_grad_(True)
    gan(D(x)).backward()
    report('full input', x.grad[:, lab.shape[1]:])
    img = fake.clone().requires_grad_(True)
    gan(D([lab, img], detach_weights=True)).backward()
    report('parts, detach_weights', img.grad)
    img = fake.clone().requires_grad_(True)
    rw, rx, second = D.forward_dual([lab, img])
    gan(rx).backward()
    report('dual', img.grad)
    img = fake.clone().requires_grad_(True)
    rw, rx, second = D.forward_dual([lab, img])
    D([lab, real])
    second()
    gan(rx).backward()
    report('dual, real pass in between', img.grad)
    img = fake.clone().requires_grad_(True)
    rw, rx, second = D.forward_dual([lab, img])
    lossw = sum(((s[-1]) ** 2).mean() for s in rw)
    D([lab, real])
    second()
    gan(rx).backward()
    report('dual, real pass, both views live', img.grad)
    lossw.backward()


def main():
    import tempfile
    ref = oracle_grads()
    d_only(ref)
    for name in ('G_L1', 'G_GAN', 'G_GAN_Feat'):
        m, _ = _model(z, tempfile.mkdtemp())
        data = {k: torch.from_numpy(z['step0/in/%s' % k]).cuda() for k in ('label', 'inst', 'image', 'pose', 'normal')}
        losses, fake = m.forward(data['label'], data['inst'].clone(), data['image'], None, data['pose'], data['normal'], infer=True)
        fake.retain_grad()
        d = dict(zip(m.loss_names, losses))
        d[name].backward()
        gref, fref = ref[name]
        ef = float((fake.grad.double().cpu() - fref).norm() / fref.norm())
        print('%-11s d/dfake rel %.3e  |hip| %.4e |ref| %.4e' % (name, ef, float(fake.grad.norm()), float(fref.norm())))
        for key in ('G/model.1.weight', 'G/model.4.weight', 'G/model.19.weight', 'E/model.1.weight'):
            net, k = key.split('/')
            p = dict(getattr(m, 'net' + net).named_parameters())[k]
            g = p.grad.double().cpu()
            r = gref[key]
            print('   %-20s rel %.3e  ratio of norms %.4f  cosine %.6f' % (
                key, float((g - r).norm() / r.norm()), float(g.norm() / r.norm()), float((g * r).sum() / (g.norm() * r.norm()))))


if __name__ == '__main__':
    main()
