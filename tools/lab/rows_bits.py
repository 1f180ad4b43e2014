"""Lab: is the frame gradient bit-identical between the product library and a lab build (LAB_LIB)?  Each in its own process; the vertex
sink is switched off (SDN_VERTEX_SINK=0: its float atomics are the only unordered sums of the step) and the dense face gradient compared
through a SHA-256 of its bytes."""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r'''
import hashlib, os, sys
sys.path[:0] = [%r, %r, %r]
import sdn_hip
if os.environ.get('LAB_LIB'):
    sdn_hip.LIB_PATH = os.path.abspath(os.environ['LAB_LIB'])
import numpy as np, torch
from sdn_hip import ops, synth
from util import posed_mesh
h = hashlib.sha256()
for seed in (3, 4, 5):
    v, f = synth.cad_like(30000, seed=seed)
    pv, ang = posed_mesh(v, f, render_size=256)
    faces = torch.tensor(pv[0][f.astype(np.int64)], device='cuda:0')[None].contiguous().requires_grad_(True)
    rgb, alpha, depth = ops.RasterizeMaps.apply(faces, None, 256, True, 0.1, 100.0, 1e-4, (0, 0, 0), False, True, False, None, False)
    g = torch.Generator(device='cuda').manual_seed(seed)
    (alpha * torch.randn(alpha.shape, generator=g, device='cuda')).sum().backward()
    h.update(faces.grad.cpu().numpy().tobytes())
print('GRAD', h.hexdigest())
''' % (ROOT, os.path.join(ROOT, '3d-sdn_amd'), os.path.join(ROOT, 'tests'))
out = {}
for name, lib in (('product', ''), ('lab', sys.argv[1])):
    env = dict(os.environ, LAB_LIB=lib)
    r = subprocess.run([sys.executable, '-c', CODE], env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith('GRAD ')]
    print(name, line[0] if line else r.stdout[-300:] + r.stderr[-1200:])
    out[name] = line[0] if line else None
print('IDENTICAL' if out['product'] and out['product'] == out['lab'] else 'DIFFERENT')
