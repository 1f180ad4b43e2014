"""Host-side cost of one textural GAN step WITHOUT a GPU (development aid).

The GAN step issues ~1800 launches; on the MI355X box the host needs longer to issue them than the GPU to run them
(bench.py: `host_enqueue_ms_per_step` ~ `ms_per_step`).  This runs the same Python -- Pix2PixHDModel.train_step over
sdn_hip.conv.ConvChain -- on CPU tensors with libsdn_hip.so replaced by a stub whose entry points return 0, so that
cProfile shows where the host time goes (plan building, ctypes marshalling, allocations, autograd glue).  The tensors
hold garbage: only the host path is being measured.

    python tools/host_overhead.py [--profile] [--size H W] [--steps K]
"""
import argparse
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, '3d-sdn_amd'), os.path.join(ROOT, '3d-sdn_amd', 'textural')):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault('SDN_ALLOW_RANDOM_INIT', '1')


class _StubLib:
    calls = 0

    def __getattr__(self, name):
        def fn(*a):
            _StubLib.calls += 1
            return 0
        fn.__name__ = name
        setattr(self, name, fn)
        return fn


def install_stub():
    import ctypes
    import sdn_hip
    from sdn_hip import conv, ops
    real = sdn_hip.lib      # the real library declares argtypes: marshal through them so the ctypes cost is included
    try:
        L = real()
        stub = _StubLib()
        for name in sdn_hip.exported_symbols():
            f = getattr(L, name)
            proto = ctypes.CFUNCTYPE(f.restype, *(f.argtypes or []))
            setattr(stub, name, proto(lambda *a: 0))
    except Exception:
        stub = _StubLib()
    for mod in (sdn_hip, conv, ops):
        mod.lib = lambda: stub
        mod.stream = lambda: None
    # tensors are on the CPU here
    torch.Tensor.is_cuda = property(lambda self: True)

    class _Stream:
        cuda_stream = 0

        def wait_stream(self, other):
            pass
    torch.cuda.current_stream = lambda *a, **k: _Stream()
    os.environ['SDN_WGRAD_STREAM'] = os.environ['SDN_D_STREAMS'] = '0'
    return stub


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--profile', action='store_true')
    ap.add_argument('--size', type=int, nargs=2, default=[32, 48])
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--ngf', type=int, default=8)
    args = ap.parse_args()
    install_stub()
    from models.pix2pixHD_model import Pix2PixHDModel, default_options
    opt = default_options(gpu_ids=[], batchSize=2, num_D=3, feat_pose='1', feat_normal='1', no_vgg_loss=True, isTrain=True,
                          ngf=args.ngf, ndf=args.ngf, nef=8)
    torch.manual_seed(0)
    model = Pix2PixHDModel()
    model.initialize(opt)
    h, w = args.size
    label = torch.randint(1, 14, (2, 1, h, w)).float()
    inst = torch.zeros(2, 1, h, w)
    inst[:, :, 4:20, 8:30] = 1000
    image = torch.rand(2, 3, h, w)
    pose = torch.zeros(2, 1, h, w)
    normal = torch.rand(2, 3, h, w)

    def step():
        return model.train_step(label, inst.clone(), image, None, pose, normal)
    step()
    step()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    dt = (time.perf_counter() - t0) / args.steps
    print('host time per step: %.1f ms' % (dt * 1e3))
    if args.profile:
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(args.steps):
            step()
        pr.disable()
        st = pstats.Stats(pr)
        st.sort_stats('cumulative').print_stats(45)
        st.sort_stats('tottime').print_stats(35)


if __name__ == '__main__':
    main()
