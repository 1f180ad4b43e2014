"""Diagnostic: repeat the generator-side backward of the train-step golden's first batch in one process and report which
tensors differ between repetitions (a bistable result was seen in tests/test_gpu_trainstep.py)."""
import json, os, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, '3d-sdn_amd'), os.path.join(ROOT, '3d-sdn_amd', 'textural')):
    sys.path.insert(0, p)
os.environ.setdefault('SDN_DETERMINISTIC', '1'); os.environ.setdefault('SDN_ALLOW_RANDOM_INIT', '1')
from test_gpu_trainstep import GOLD, _model
z = np.load(GOLD)
runs = []
junk = []
for rep in range(8):
    if rep % 2:
        junk.append(torch.randn(1 << (18 + rep % 5), device='cuda'))   # perturb the allocator between repetitions
    else:
        junk.clear()
    m, _ = _model(z, tempfile.mkdtemp())
    data = {k: torch.from_numpy(z['step0/in/%s' % k]).cuda() for k in ('label', 'inst', 'image', 'pose', 'normal')}
    losses, fake = m.forward(data['label'], data['inst'].clone(), data['image'], None, data['pose'], data['normal'], infer=True)
    fake.retain_grad()
    d = dict(zip(m.loss_names, losses))
    (d['G_GAN'] + d['G_GAN_Feat'] + d['G_L1']).backward()
    rec = {'fake': fake.detach().clone(), 'fake.grad': fake.grad.clone()}
    for n in 'GE':
        for k, p in getattr(m, 'net' + n).named_parameters():
            rec['%s/%s' % (n, k)] = p.grad.clone() if p.grad is not None else torch.zeros_like(p)
    runs.append(rec)
ref = runs[0]
for i, r in enumerate(runs[1:], 1):
    diffs = []
    for k in ref:
        a, b = ref[k].double(), r[k].double()
        e = float((a - b).norm() / (a.norm() + 1e-300))
        if e > 1e-6:
            diffs.append((k, e))
    print('run %d vs run 0: %d tensors differ' % (i, len(diffs)), diffs[:12])
