"""One process = K identical frame steps (bench.py's configs[1] step: 16 objects, decode + transform + three maps + silhouette
loss, forward and backward), nothing else: the target of `rocprofv3 --kernel-trace --stats` for a per-step kernel table
(tools/gpu_prof_geo.sh divides every total by the K this script prints).

    python tools/prof_geo.py [--steps 20] [--mesh car_like|cad_like]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, '3d-sdn_amd'), os.path.join(ROOT, '3d-sdn_amd', 'geometric')]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--mesh', default='car_like')
    ap.add_argument('--timing', action='store_true', help='also print the mean k_raster_tiles / edge kernel durations (hipEvents)')
    ap.add_argument('--k1', action='store_true', help='the K1 coverage mode (neural_renderer.use_unsafe_rasterizer(True))')
    ap.add_argument('--lib', default=None, help='a lab build of libsdn_hip.so (tools/build_lab_variant.sh) instead of the product library')
    a = ap.parse_args()
    if a.lib:
        import sdn_hip as _sh
        _sh.LIB_PATH = os.path.abspath(a.lib)
    import torch
    import bench
    device = torch.device('cuda', 0)
    if a.k1:
        from sdn_hip import ops
        ops.set_k1_coverage(True)
    bank, sizes, cls, params, targets, ptf = bench.build_scene(device, seed=1234, mesh=a.mesh)
    step = bench.make_step(device, bank, cls, params, targets, ptf, backward=True, pack=False)
    step()
    torch.cuda.synchronize()
    if a.timing:
        import sdn_hip
        sdn_hip.timing_enable(True)
        for slot in (sdn_hip.SLOT_RASTER_TILES, sdn_hip.SLOT_EDGE_SCAN):
            sdn_hip.timing_read_slot(slot)
    t0 = time.perf_counter()
    for _ in range(a.steps - 1):
        step()
    torch.cuda.synchronize()
    if a.timing:
        rt = sdn_hip.timing_read_slot(sdn_hip.SLOT_RASTER_TILES)
        es = sdn_hip.timing_read_slot(sdn_hip.SLOT_EDGE_SCAN)
        print('PROF_GEO_TIMING mesh %s  k_raster_tiles %.1f us  edge kernels %.1f us  (lib %s)' % (
            a.mesh, rt[0] * 1e3 / max(rt[1], 1), es[0] * 1e3 / max(es[1], 1), a.lib or 'product'))
    print('PROF_GEO steps %d  ms_per_step %.3f  mesh %s' % (a.steps, (time.perf_counter() - t0) / max(a.steps - 1, 1) * 1e3, a.mesh))


if __name__ == '__main__':
    main()
