"""Which Python lines launch the torch-side kernels of one textural GAN step?  (GPU only; a diagnosis tool, not a test.)

The kernel statistics of the textural leg (profiles/*_tex_kernel_stats.csv) show a few hundred small fills / copies /
element-wise launches per step besides the sdn:: kernels.  This profiles ONE train step with torch.profiler (Python
stacks on) and prints, for every aten op that launched device work, the call count, the device time and the innermost
repo frames -- so each launch can be traced to a line of sdn_hip/conv.py, textural/models/*.py or torch.optim.

    python tools/attribute_torch_ops.py [out.txt] [tex|geo|opt]
"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (sets up the package path)


def geo_step(dev):
    class A:
        forward_only = False
    bank, sizes, cls, params, targets, ptf = bench.build_scene(dev, seed=1234)
    return bench.make_step(dev, bank, cls, params, targets, ptf, backward=True)


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else None
    which = sys.argv[2] if len(sys.argv) > 2 else 'tex'
    dev = torch.device('cuda', 0)
    if which == 'geo':
        step = geo_step(dev)
    elif which == 'opt':
        step = opt_step(dev)
    else:
        step = tex_step(dev)
    report(step, out_path, which)


def opt_step(dev):
    """ONE iteration of the configs[2] test-time optimisation loop (bench.derender3d_loop, geometric/scripts/main.py:433-456)
    on the drop-in Derenderer3d"""
    import numpy as np
    from derender3d import TargetType
    from derender3d.models import Derenderer3d, ShapenetObj
    from sdn_hip import synth
    objs = []
    for k in range(8):
        v, f = synth.cad_like(46000, seed=100 + k)
        objs.append(ShapenetObj(vertices=v[:, [2, 1, 0]] * np.asarray([-1, 1, 1], np.float32), faces=f))
    torch.manual_seed(7)
    model = Derenderer3d(mode=TargetType.extend, image_size=256, render_size=bench.RENDER_SIZE, objs=objs).to(dev)
    n = bench.OBJECTS_PER_FRAME
    rng = np.random.default_rng(1236)
    images = torch.tensor(rng.normal(size=(n, 3, 224, 224)).astype(np.float32), device=dev)
    c = np.stack([rng.uniform(-0.15, 0.15, n), rng.uniform(-0.6, 0.6, n)], 1)
    h, w = rng.uniform(40, 150, n) / bench.FOCAL, rng.uniform(60, 300, n) / bench.FOCAL
    rois = torch.tensor(np.stack([c[:, 0] - h / 2, c[:, 1] - w / 2, c[:, 0] + h / 2, c[:, 1] + w / 2], 1).astype(np.float32), device=dev)
    focals = torch.full((n, 1), bench.FOCAL, device=dev)
    masks = torch.zeros(n, 1, bench.RENDER_SIZE, bench.RENDER_SIZE, device=dev)
    masks[:, :, 120:270, 40:340] = 1
    model.eval()
    with torch.no_grad():
        blob = model(images, rois, focals)
    model.train()
    model._force_no_sample = True
    b = {k: (v.clone().detach() if isinstance(v, torch.Tensor) else v) for k, v in blob.items()}
    params = {k: b[k].requires_grad_() for k in ('_theta_deltas', '_translation2ds', '_log_scales', '_ffd_coeffs')}
    opt = torch.optim.Adam(params.values(), lr=3e-2)

    def step():
        opt.zero_grad()
        b.update(model.render(b))
        loss = torch.nn.functional.mse_loss(b['_masks'], masks, reduction='none') + 100 * torch.mean(b['_ffd_coeffs'] ** 2)
        loss.mean().backward()
        opt.step()
    return step


def tex_step(dev):
    sys.path.insert(0, os.path.join(ROOT, '3d-sdn_amd', 'textural'))
    from models.pix2pixHD_model import Pix2PixHDModel, default_options
    opt = default_options(gpu_ids=[0], batchSize=bench.TEX_BATCH, num_D=3, feat_pose='1', feat_normal='1',
                          no_vgg_loss=True, isTrain=True)
    torch.manual_seed(4321)
    model = Pix2PixHDModel()
    model.initialize(opt)
    label, inst, image, pose, normal = bench.textural_batch(model, dev, 77)

    def step():
        return model.train_step(label, inst.clone(), image, None, pose, normal)
    return step


def report(step, out_path, which):
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    try:   # without the verbose switch this torch build records empty Python stacks
        cfg = torch._C._profiler._ExperimentalConfig(verbose=True)
    except Exception:
        cfg = None
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True,
                 experimental_config=cfg) as prof:
        step()
        torch.cuda.synchronize()
    lines = []
    # per (op, innermost repo frames): count and device time
    agg = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        dt = getattr(ev, 'self_device_time_total', None)
        if dt is None:
            dt = getattr(ev, 'self_cuda_time_total', 0.0)
        if not dt or not ev.name.startswith('aten::'):
            continue
        frames = []
        for fr in (ev.stack or []):
            if '/torch/' in fr and 'optim' not in fr:
                continue
            if 'attribute_torch_ops' in fr:
                continue
            frames.append(fr.replace(ROOT + '/', ''))
            if len(frames) == 3:
                break
        shp = str(getattr(ev, 'input_shapes', '') or '')[:60]
        key = (ev.name, ' <- '.join(frames) + '   ' + shp)
        agg[key][0] += 1
        agg[key][1] += dt
    tot = sum(v[1] for v in agg.values())
    lines.append('aten ops with device time in one ' + which + ' step: %d launches-worth, %.2f ms' % (sum(v[0] for v in agg.values()), tot / 1e3))
    for (name, where), (cnt, dt) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:120]:
        lines.append('%8.1f us %5d x  %-22s %s' % (dt, cnt, name, where))
    lines.append('')
    lines.append('by op:')
    byop = collections.defaultdict(lambda: [0, 0.0])
    for (name, _), (cnt, dt) in agg.items():
        byop[name][0] += cnt
        byop[name][1] += dt
    for name, (cnt, dt) in sorted(byop.items(), key=lambda kv: -kv[1][1]):
        lines.append('%8.1f us %5d x  %s' % (dt, cnt, name))
    text = '\n'.join(lines)
    print(text)
    if out_path:
        with open(out_path, 'w') as f:
            f.write(text + '\n')


if __name__ == '__main__':
    main()
