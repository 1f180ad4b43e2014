#!/usr/bin/env python3
"""bench.py -- rendered-objects/sec of the geometric hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One STEP = one VKITTI frame's worth of objects (16, the cap of geometric/scripts/main.py:812) per GPU, each object
going through exactly what one iteration of the reference's test-time optimisation loop does
(geometric/scripts/main.py:439-456 over derender3d/models/__init__.py:161-224):
    FFD decode -> PerspectiveTransform (zoom-to-fit) -> silhouette + normal + depth at render_size 384
    (2x anti-aliasing: 768^2 internal), loss = MSE(mask, target) + 100 * mean(ffd^2), backward to the pose / FFD
    parameters.
Meshes: 8 procedural car-class templates of ~43k triangles (85.7k faces after fill_back) -- the ShapeNet files and
every dataset are absent on the benchmark box (sdn_hip/synth.py).  Inputs are resident in HBM before the timed
region; nothing leaves the device inside it.  With N > 1 the objects are sharded over the ranks (weak scaling: 16 per
GPU) and the rendered maps are exchanged with ONE RCCL all_gather per step, the only exchange this path has.

Extra objects in the JSON line:
  roofline      dominant kernel k_raster_tiles: algorithmic bytes per launch (SURVEY.md 8(d) per-object figure:
                12 V + 12 F0 + 20 S^2 + 20 R^2) / its mean launch time, measured with hipEvents on the launch
                stream inside the timed region (sdn_timing_*), against 8 TB/s.
  cpu_baseline  the CPU oracle (port of the reference kernels, OpenMP over pixels) on ONE object of the same
                workload -- a single rgb+alpha+depth rasterisation + the silhouette backward -- on this host's cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, '3d-sdn_amd'), os.path.join(ROOT, '3d-sdn_amd', 'geometric'),
           os.path.join(ROOT, 'tests')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

OBJECTS_PER_FRAME = 16
RENDER_SIZE = 384
FOCAL = 725.0  # VKITTI camera (geometric/derender3d/datasets.py:207-213)
N_TRIS = 45000


def build_scene(device, seed):
    """8 templates + 16 object poses, everything resident on `device`."""
    from derender3d.models.transforms import FFD, FFDBank, PerspectiveTransform
    from sdn_hip import synth
    rng = np.random.default_rng(seed)
    ffds, faces, sizes = [], [], []
    for k in range(8):
        v, f = synth.car_like(N_TRIS, seed=100 + k)
        v = v[:, [2, 1, 0]] * np.asarray([-1, 1, 1], np.float32)  # ShapenetObj axis convention
        ffds.append(FFD(torch.tensor(v), constraints=[
            FFD.Constraint.symmetry(axis=FFD.Constraint.Axis.z),
            FFD.Constraint.homogeneity(axis=FFD.Constraint.Axis.y, index=[0, 1])]))
        faces.append(torch.tensor(f))
        sizes.append((v.shape[0], f.shape[0]))
    bank = FFDBank(ffds, faces).to(device)
    n = OBJECTS_PER_FRAME
    cls = rng.integers(0, 8, n)
    theta = rng.uniform(-np.pi, np.pi, n)
    params = {
        'ffd': torch.tensor(rng.normal(0, 0.02, (n, 192)).astype(np.float32), device=device, requires_grad=True),
        'log_scale': torch.tensor(np.log(np.array([3.9, 1.5, 1.6], np.float32))[None].repeat(n, 0) +
                                  rng.normal(0, 0.05, (n, 3)).astype(np.float32), device=device, requires_grad=True),
        'theta': torch.tensor(theta.astype(np.float32)[:, None], device=device, requires_grad=True),
        'translation': torch.tensor(np.stack([rng.uniform(-8, 8, n), rng.uniform(0.5, 2, n), -rng.uniform(8, 40, n)],
                                             1).astype(np.float32), device=device, requires_grad=True),
    }
    targets = torch.zeros(n, 1, RENDER_SIZE, RENDER_SIZE, device=device)
    targets[:, :, 120:270, 40:340] = 1
    return bank, sizes, cls, params, targets, PerspectiveTransform()


def make_step(device, bank, cls, params, targets, ptf, backward=True):
    """The per-object work of derender3d/models/__init__.py:161-224 + the loss of scripts/main.py:445-453, for the 16
    objects of a frame in one batch of launches."""
    from derender3d.models.renderer import Renderer
    n = OBJECTS_PER_FRAME
    renderer = Renderer(image_size=RENDER_SIZE)
    renderer.viewing_angle = [np.arctan(RENDER_SIZE / (2.0 * FOCAL)) / np.pi * 180] * n
    zoom_to = torch.full((n, 1), RENDER_SIZE / (2.0 * FOCAL), device=device)
    zeros = torch.zeros(n, 1, device=device)
    cls_t = torch.tensor(cls, device=device, dtype=torch.int64)

    def step():
        verts, faces = bank.decode(params['ffd'], cls_t)
        th = params['theta']
        rot = torch.cat([torch.cos(th / 2), zeros, torch.sin(th / 2), zeros], dim=1)
        tr = params['translation']
        verts, _ = ptf(verts, scales=torch.exp(params['log_scale']), rotations=rot, translations=tr,
                       perspective_translations=tr, zoom_tos=zoom_to)
        mask, normal, depth = renderer.render_maps(verts, faces)
        if backward:
            loss = ((mask - targets) ** 2).mean(dim=(1, 2, 3)).sum() + 100 * (params['ffd'] ** 2).mean(dim=1).sum()
            for p in params.values():
                p.grad = None
            loss.backward()
        return torch.cat([mask, normal, depth], dim=1)

    return step


def cpu_baseline():
    """Oracle on one object of the workload (bounded: one pass, a few tens of seconds of CPU work at most)."""
    from oracle import nr_oracle as no
    from oracle import raster_np as rn
    from sdn_hip import synth
    from util import posed_mesh
    v, f = synth.car_like(N_TRIS, seed=100)
    pv, ang = posed_mesh(v, f)
    r = no.NRRenderer()
    r.image_size = RENDER_SIZE
    r.viewing_angle = ang
    r.camera_mode = 'look'
    r.eye = torch.zeros(1, 3)
    r.camera_direction = torch.tensor([[0., 0., -1.]])
    r.up = torch.tensor([[0., 1., 0.]])
    vt = (torch.tensor(pv) * torch.tensor([-1., 1., 1.])).requires_grad_(True)
    fi = torch.tensor(f[None])
    t0 = time.time()
    filled = r._fill_back(fi)
    normals = r.face_normals(vt, filled)
    tex = normals[:, :, None, None, None, :].repeat(1, 1, 2, 2, 2, 1)
    faces9 = no.vertices_to_faces(r._camera(vt), filled)
    out = no.rasterize_rgbad(faces9, tex, RENDER_SIZE, True, 0.1, 100, 1e-3, (0, 0, 0), True, True, True)
    target = torch.zeros(1, RENDER_SIZE, RENDER_SIZE)
    target[:, 120:270, 40:340] = 1
    ((out['alpha'] - target) ** 2).mean().backward()
    dt = time.time() - t0
    return {'value': 1.0 / dt, 'unit': 'objects/s', 'cores': rn.num_threads(), 'kind': 'port',
            'sample': '1 object (%d faces, 768^2): ONE rgb+alpha+depth rasterisation + silhouette backward, %.1f s; '
                      'the reference would rasterise three times' % (2 * f.shape[0], dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--forward-only', action='store_true')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the HIP path has no CPU fallback')
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world)
    if args.gpus != world and rank == 0 and world > 1:
        print('warning: --gpus %d but WORLD_SIZE %d' % (args.gpus, world), file=sys.stderr)

    import sdn_hip
    sdn_hip.lib()
    bank, sizes, cls, params, targets, ptf = build_scene(device, seed=1234 + rank)
    step = make_step(device, bank, cls, params, targets, ptf, backward=not args.forward_only)
    gathered = None
    if world > 1:
        gathered = torch.empty(world * OBJECTS_PER_FRAME, 5, RENDER_SIZE, RENDER_SIZE, device=device)

    def full_step():
        maps = step()
        if world > 1:
            dist.all_gather_into_tensor(gathered, maps.detach().contiguous())
        return maps

    for _ in range(args.warmup):
        full_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    sdn_hip.timing_enable(True)
    sdn_hip.timing_read()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kern_ms, launches = sdn_hip.timing_read()
    sdn_hip.timing_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        objects = world * OBJECTS_PER_FRAME * args.steps
        vmean = float(np.mean([sizes[c][0] for c in cls]))
        fmean = float(np.mean([sizes[c][1] for c in cls]))
        S = 2 * RENDER_SIZE
        objs_per_launch = OBJECTS_PER_FRAME  # the whole frame is one k_raster_tiles launch (bs = 16)
        alg_bytes = objs_per_launch * (12 * vmean + 12 * fmean + 20 * S * S + 20 * RENDER_SIZE * RENDER_SIZE)
        kern_s = kern_ms / 1e3 / max(launches, 1)
        achieved = alg_bytes / kern_s / 1e9 if launches else 0.0
        traffic = None
        pmc = os.path.join(ROOT, 'profiles', 'pmc_raster_tiles.json')
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get('hbm_bytes_per_launch')
            except Exception:
                traffic = None
        line = {
            'metric': 'rendered-objects/sec (FFD decode + transform + silhouette/normal/depth @384, fwd+bwd)'
            if not args.forward_only else 'rendered-objects/sec (forward only)',
            'value': objects / elapsed,
            'unit': 'objects/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32',
            'data': 'synthetic',
            'config': {'workload': 'configs[1]: car-class mesh (%.0f tris, %.0f faces with fill_back) render fwd+bwd, '
                                   '16 objects of a 375x1242 VKITTI frame per step per GPU, render_size 384 (768^2 '
                                   'internal)' % (fmean, 2 * fmean),
                       'objects_per_step_per_gpu': OBJECTS_PER_FRAME, 'render_size': RENDER_SIZE,
                       'parallelism': 'objects sharded over %d rank(s)%s' % (
                           world, ', one RCCL all_gather of [16,5,384,384] maps per step' if world > 1 else '')},
            'roofline': {'bound': 'hbm', 'kernel': 'k_raster_tiles', 'achieved': achieved, 'peak': 8000.0,
                         'unit': 'GB/s', 'frac': achieved / 8000.0, 'traffic': traffic,
                         'algorithmic_bytes_per_launch': alg_bytes, 'launches': launches,
                         'avg_launch_us': kern_s * 1e6,
                         'objects_per_launch': objs_per_launch,
                         'note': 'one launch = the 16 objects of a frame; the kernel is ALU/latency-bound, see DESIGN.md'},
            'textural_gan_fwd_bwd_ms': None,
        }
        if not args.no_cpu_baseline and world == 1:
            try:
                line['cpu_baseline'] = cpu_baseline()
            except Exception as e:  # the baseline must never take the GPU number down with it
                line['cpu_baseline'] = {'value': None, 'unit': 'objects/s', 'cores': 0, 'kind': 'port',
                                        'sample': 'failed: %r' % (e,)}
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
