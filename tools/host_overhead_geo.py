"""Host-side cost of one geometric frame step (bench.make_step) WITHOUT a GPU: CPU tensors, libsdn_hip.so replaced by
entry points that return 0 through the same ctypes signatures.  Development aid for the issue-bound case (on a slow
host the 16-object frame step is bound by the ~90 launches the host has to issue)."""
import cProfile
import ctypes
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def install_stub():
    import sdn_hip
    from sdn_hip import ops
    L = sdn_hip.lib()

    class Stub:
        pass
    stub = Stub()
    keep = []
    for name in sdn_hip.exported_symbols():
        f = getattr(L, name)
        if name in ('sdn_raster_workspace_bytes', 'sdn_raster_bwd_workspace_bytes', 'sdn_last_error', 'sdn_version'):
            setattr(stub, name, f)      # pure host functions: the real ones
            continue
        proto = ctypes.CFUNCTYPE(f.restype, *(f.argtypes or []))
        cb = proto(lambda *a: 0)
        keep.append(cb)
        setattr(stub, name, cb)
    stub._keep = keep
    import derender3d.models.transforms as tr
    for mod in (sdn_hip, ops, tr):
        if hasattr(mod, 'lib'):
            mod.lib = lambda: stub
        if hasattr(mod, 'stream'):
            mod.stream = lambda: None
    torch.Tensor.is_cuda = property(lambda self: True)


def main():
    install_stub()
    dev = torch.device('cpu')
    if '--tiny' in sys.argv:   # small meshes and maps: what is left is the per-launch host cost
        bench.N_TRIS = 300
        bench.RENDER_SIZE = 32
    bank, sizes, cls, params, targets, ptf = bench.build_scene(dev, seed=1234)
    step = bench.make_step(dev, bank, cls, params, targets, ptf, backward=True)
    for _ in range(3):
        step()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    print('host time per frame step: %.3f ms' % ((time.perf_counter() - t0) / n * 1e3))
    if '--profile' in sys.argv:
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(n):
            step()
        pr.disable()
        st = pstats.Stats(pr)
        st.sort_stats('cumulative').print_stats(40)
        st.sort_stats('tottime').print_stats(30)


if __name__ == '__main__':
    main()
