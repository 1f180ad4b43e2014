"""GPU: Pix2PixHDModel.train_step against the REFERENCE's train loop.

tests/golden/trainstep_golden.npz holds two consecutive iterations of textural/train.py:69-95 executed literally by the
reference's own Pix2PixHDModel / networks.py on the CPU in float64 (tests/golden/make_trainstep_golden.py).  The product
model is built from the options the reference's parser produced (stored in the file), starts from the same weights and
sees the same two batches; compared per step:

  * the eight losses                                                  (gate 2e-4 relative; measured ~1e-5)
  * the gradients each optimizer consumes (captured by step pre-hooks) (gate 2e-3 relative L2 per tensor; measured
    below -- the L1 feature-matching / reconstruction losses differentiate to sign(a - b), so a forward difference of
    1e-6 flips a few signs: the same finite effect as the ReLU pattern of test_full_generator_activations_vs_oracle)
  * the parameter updates w_after - w_before.  Adam's first steps are lr * g / (|g| + 1e-8): an element whose
    gradient is smaller than the gradient error can move by the full +-lr in either direction, so the comparison is made
    (a) over the elements whose reference gradient is above 1e-3 of the tensor's rms: 1e-3 relative L2 (measured ~1e-5),
    (b) over the whole tensor: the fraction of elements whose update differs by more than 1 % of lr stays below 1 %;
    parameters whose reference gradient is exactly zero in exact arithmetic (a bias in front of an InstanceNorm) must
    not move at all
  * the InstanceNorm running statistics after each step (three discriminator passes per step in the reference's order
    fake / real / fake, pix2pixHD_model.py:192,194,210)                (gate 1e-4)
  * step 2's numbers only agree if step 1's updates reached every packed-weight cache (the r02 stale-cache bug).

What the product does differently from the reference loop -- 7 discriminator passes instead of 9, one dual-view pass for the
fake image, side streams -- is therefore pinned against the reference's own sequence, not against itself."""
import json
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, '3d-sdn_amd'), os.path.join(ROOT, '3d-sdn_amd', 'textural'), os.path.join(ROOT, 'tests')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, 'tests', 'golden', 'trainstep_golden.npz')


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-300))


def _model(z, tmp_path):
    from models.pix2pixHD_model import Pix2PixHDModel
    opt = SimpleNamespace(**json.loads(str(z['meta/opt_json'])))
    opt.gpu_ids, opt.checkpoints_dir, opt.verbose = [0], str(tmp_path), False
    m = Pix2PixHDModel()
    m.initialize(opt)
    for name in ('G', 'D', 'E'):
        net = getattr(m, 'net' + name)
        sd = {k[len('init/%s/' % name):]: torch.from_numpy(z[k]) for k in z.files if k.startswith('init/%s/' % name)}
        assert set(sd) == set(net.state_dict()), name
        net.load_state_dict(sd)
    from sdn_hip import conv as hc
    hc.invalidate_weight_caches()
    return m, opt


# per-step gates: (loss, gradient rel L2, update rel L2 on well-conditioned elements, fraction moving differently, running
# statistics).  Step 0 starts from identical weights: what is measured is the arithmetic.  Step 1 starts from each side's
# OWN step-0 result: Adam's first update is +-lr for every element, so the elements whose step-0 gradient is below the
# gradient error (a fraction ~1e-3) sit 2 lr apart on the two sides, and this small network amplifies that weight
# difference (measured: gradients 7e-3; the second Adam update, which mixes both steps' gradients, differs on 11 % of the
# encoder's elements); the gates of step 1 still catch a forward pass that missed the step-0 update -- stale packed weights
# change its losses by 0.7 ... 6 % and its gradients by more.
GATES = [dict(loss=2e-4, grad=2e-3, dw_big=1e-3, dw_frac=0.01, running=1e-4),
         dict(loss=2e-3, grad=3e-2, dw_big=0.1, dw_frac=0.2, running=1e-3)]


@pytest.mark.parametrize('streams', ['side_streams', 'single_stream'])
def test_train_step_reproduces_the_reference_loop(streams, monkeypatch, tmp_path):
    monkeypatch.setenv('SDN_DETERMINISTIC', '1')
    if streams == 'single_stream':
        monkeypatch.setenv('SDN_D_STREAMS', '0')
        monkeypatch.setenv('SDN_WGRAD_STREAM', '0')
    z = np.load(GOLD)
    m, opt = _model(z, tmp_path)
    lr = opt.lr
    taken = {}

    def grab(tag, nets):
        def hook(optimizer, args, kwargs):
            taken[tag] = {n: {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p))
                              for k, p in getattr(m, 'net' + n).named_parameters()} for n in nets}
        return hook
    m.optimizer_G.register_step_pre_hook(grab('G', ('G', 'E')))
    m.optimizer_D.register_step_pre_hook(grab('D', ('D',)))
    failures = []
    for step in range(int(z['meta/steps'])):
        gate = GATES[step]
        worst = {k: (0.0, '') for k in gate}

        def see(kind, value, what):
            if value > worst[kind][0]:
                worst[kind] = (value, what)
        data = {k: torch.from_numpy(z['step%d/in/%s' % (step, k)]).cuda() for k in ('label', 'inst', 'image', 'pose', 'normal')}
        before = {n: {k: p.detach().clone() for k, p in getattr(m, 'net' + n).named_parameters()} for n in 'GDE'}
        d = m.train_step(data['label'], data['inst'].clone(), data['image'], None, data['pose'], data['normal'])
        torch.cuda.synchronize()
        for k in ('G_GAN', 'G_GAN_Feat', 'D_real', 'D_fake', 'G_L1'):
            want = float(z['step%d/loss/%s' % (step, k)])
            see('loss', abs(float(d[k].detach()) - want) / abs(want), k)
        for n in 'GDE':
            net = getattr(m, 'net' + n)
            grads = taken['D' if n == 'D' else 'G'][n]
            for k, p in net.named_parameters():
                g_ref = torch.from_numpy(z['step%d/grad/%s/%s' % (step, n, k)]).double()
                dw_ref = torch.from_numpy(z['step%d/dw/%s/%s' % (step, n, k)]).double()
                g = grads[k].double().cpu()
                dw = (p.detach() - before[n][k]).double().cpu()
                what = 'net%s %s' % (n, k)
                if float(g_ref.abs().max()) < 1e-12:
                    # exact-arithmetic zero (bias in front of InstanceNorm): no gradient, no movement
                    if not (float(g.abs().max()) == 0.0 and float(dw.abs().max()) <= 1e-9):
                        failures.append('step %d %s: a bias in front of InstanceNorm moved' % (step, what))
                    continue
                see('grad', rel_l2(g, g_ref), what)
                big = g_ref.abs() > 1e-3 * g_ref.pow(2).mean().sqrt()
                if int(big.sum()):
                    see('dw_big', float((dw[big] - dw_ref[big]).norm() / dw_ref[big].norm()), what)
                if dw.numel() >= 200:
                    see('dw_frac', float(((dw - dw_ref).abs() > 0.01 * lr).double().mean()), what)
            for k, v in net.state_dict().items():
                if 'running_' in k:
                    want = torch.from_numpy(z['step%d/running/%s/%s' % (step, n, k)])
                    see('running', float((v.detach().cpu() - want).abs().max() / (want.abs().max() + 1e-30)), 'net%s %s' % (n, k))
        print('train step %d vs the reference loop (%s): ' % (step, streams)
              + ', '.join('%s %.2e (%s)' % (k, v[0], v[1]) for k, v in worst.items()))
        for k, (v, what) in worst.items():
            if v > gate[k]:
                failures.append('step %d %s: %s %.3e > %.1e' % (step, what, k, v, gate[k]))
    assert not failures, failures
