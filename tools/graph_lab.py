"""Can the 16-object frame step (decode, transform, render, loss, backward) be captured in a HIP graph and replayed?
Development aid (GPU).  usage: graph_lab.py <stage>   stage: fwd | grad | all (runs the stages in subprocesses)"""
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def make(dev, backward):
    from derender3d.models.renderer import Renderer
    bank, sizes, cls, params, targets, ptf = bench.build_scene(dev, seed=1234)
    n = bench.OBJECTS_PER_FRAME
    renderer = Renderer(image_size=bench.RENDER_SIZE)
    renderer.viewing_angle = [np.arctan(bench.RENDER_SIZE / (2.0 * bench.FOCAL)) / np.pi * 180] * n
    zoom_to = torch.full((n, 1), bench.RENDER_SIZE / (2.0 * bench.FOCAL), device=dev)
    zeros = torch.zeros(n, 1, device=dev)
    cls_t = torch.tensor(cls, device=dev, dtype=torch.int64)
    plist = [params[k] for k in sorted(params)]

    def step():
        verts, faces = bank.decode(params['ffd'], cls_t)
        th = params['theta']
        rot = torch.cat([torch.cos(th / 2), zeros, torch.sin(th / 2), zeros], dim=1)
        tr = params['translation']
        verts, _ = ptf(verts, scales=torch.exp(params['log_scale']), rotations=rot, translations=tr,
                       perspective_translations=tr, zoom_tos=zoom_to)
        if backward == 'fwd_gradmode':
            mask, normal, depth = renderer.render_maps(verts, faces)
            return torch.cat([mask, normal, depth], dim=1), []
        if backward == 'g_verts':
            return verts, list(torch.autograd.grad((verts ** 2).sum(), plist))
        if backward == 'g_decode':
            v0, _ = bank.decode(params['ffd'], cls_t)
            return v0, list(torch.autograd.grad((v0 ** 2).sum(), [params['ffd']]))
        mask, normal, depth = renderer.render_maps(verts, faces)
        maps = torch.cat([mask, normal, depth], dim=1)
        if not backward:
            return maps, []
        if backward == 'g_ffd':
            loss = 100 * (params['ffd'] ** 2).mean(dim=1).sum()
            return maps, list(torch.autograd.grad(loss, [params['ffd']]))
        if backward == 'g_depth':
            return maps, list(torch.autograd.grad((depth ** 2).mean(), plist))
        if backward == 'g_normal':
            return maps, list(torch.autograd.grad((normal ** 2).mean(), plist))
        if backward == 'g_mask':
            return maps, list(torch.autograd.grad(((mask - targets) ** 2).mean(), plist))
        loss = ((mask - targets) ** 2).mean(dim=(1, 2, 3)).sum() + 100 * (params['ffd'] ** 2).mean(dim=1).sum()
        return maps, list(torch.autograd.grad(loss, plist))
    return step, params


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def stage(which):
    dev = torch.device('cuda', 0)
    if which == 'torch_only':
        p = torch.randn(16, 192, device=dev, requires_grad=True)

        def f():
            return torch.autograd.grad(100 * (p ** 2).mean(dim=1).sum(), [p])[0]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                f()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        print('[torch_only] capturing', flush=True)
        with torch.cuda.graph(g):
            out = f()
        g.replay()
        torch.cuda.synchronize()
        print('[torch_only] ok, grad norm %.4f vs %.4f' % (float(out.norm()), float((200 * p / 192).norm())), flush=True)
        return
    step, params = make(dev, False if which == 'fwd' else (True if which == 'grad' else which))
    if which == 'fwd':
        inner = step

        def step():
            with torch.no_grad():
                return inner()
    ref, ref_g = step()
    ref, ref_g = ref.clone(), [g.clone() for g in ref_g]
    print('[%s] eager  %.3f ms per frame' % (which, timed(step)), flush=True)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    print('[%s] capturing' % which, flush=True)
    # the autograd engine's device thread launching into the capturing stream crashes hipStreamEndCapture on this
    # ROCm build: keep the backward pass on the calling thread
    with torch.autograd.set_multithreading_enabled(False):
        with torch.cuda.graph(g):
            out, grads = step()
    print('[%s] captured' % which, flush=True)
    g.replay()
    torch.cuda.synchronize()
    print('[%s] replay vs eager: maps max abs diff %.3e' % (which, float((out - ref).abs().max())))
    for a, b in zip(grads, ref_g):
        print('   grad rel L2 %.3e' % float((a - b).norm() / (b.norm() + 1e-30)))
    with torch.no_grad():
        params['translation'][:, 0] += 0.5
    g.replay()
    torch.cuda.synchronize()
    moved = out.clone()
    eager, _ = step()
    print('[%s] after moving the objects: replay vs eager max abs diff %.3e (moved by %.3e)'
          % (which, float((moved - eager).abs().max()), float((moved - ref).abs().max())))
    print('[%s] graph  %.3f ms per frame' % (which, timed(g.replay)), flush=True)


if __name__ == '__main__':
    w = sys.argv[1] if len(sys.argv) > 1 else 'all'
    if w == 'all':
        for st in sys.argv[2:] or ('fwd', 'grad'):
            r = subprocess.run([sys.executable, '-X', 'faulthandler', __file__, st], capture_output=True, text=True)
            print(r.stdout[-3000:])
            if r.returncode:
                err = [l for l in r.stderr.splitlines() if 'Warning' not in l and 'Extension modules' not in l]
                print('stage %s: exit %d\n%s' % (st, r.returncode, '\n'.join(err[-40:])[:4000]))
    else:
        stage(w)
