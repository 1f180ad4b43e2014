#!/usr/bin/env python3
"""Copy a round's evidence (gpurun_out/<tag>_*, written by tools/gpu_round.sh) into profiles/ and write
profiles/<tag>_summary.md.   usage: tools/round_summary.py <tag> [previous bench json for the comparison column]"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
prev = json.load(open(sys.argv[2]))['parsed'] if len(sys.argv) > 2 and os.path.exists(sys.argv[2]) else None
G, P = os.path.join(ROOT, 'gpurun_out'), os.path.join(ROOT, 'profiles')
names = ['bench.json', 'geo_kernel_stats.csv', 'tex_kernel_stats.csv', 'pmc_geo_FETCH_SIZE.json', 'pmc_geo_WRITE_SIZE.json',
         'pmc_tex_FETCH_SIZE.json', 'pmc_tex_WRITE_SIZE.json', 'smoke.log']
for n in names:
    src = os.path.join(G, '%s_%s' % (tag, n))
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, '%s_%s' % (tag, n)))
for kind, pre in (('geo', 'pmc_'), ('tex', 'pmc_tex_')):   # the files bench.py reads
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        shutil.copy(os.path.join(G, '%s_pmc_%s_%s.json' % (tag, kind, c)), os.path.join(P, '%s%s.json' % (pre, c)))
tests_tail = [l.strip() for l in open(os.path.join(G, tag + '_tests.log')) if ' passed' in l or ' failed' in l]


def steps_of(path, marker, per_step, fallback):
    """number of steps in a kernel-stats file = launches of a kernel that is known to run `per_step` times per step"""
    for r in csv.DictReader(open(path)):
        if marker in r['Name']:
            return max(1, round(int(r['Calls']) / per_step))
    return fallback


MFMA = ('k_conv_gemm', 'k_conv_tile', 'k_conv_halo', 'k_wgrad_tile', 'k_conv_wgrad<')


def non_mfma_ms(path, steps):
    rows = list(csv.DictReader(open(path)))
    return sum(float(r['TotalDurationNs']) for r in rows if not any(m in r['Name'] for m in MFMA)) / 1e6 / steps


def top(path, steps, n=14):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    out = ['| `%s` | %.1f | %.1f | %.1f %% |' % (r['Name'].split('(')[0][:60], float(r['Calls']) / steps,
                                               float(r['TotalDurationNs']) / 1e3 / steps,
                                               float(r['TotalDurationNs']) / tot * 100) for r in rows[:n]]
    return out, tot / 1e6 / steps


# divisors from the launch counts: k_edge_reduce runs once per frame step (backward), the generator's head-layer
# weight gradient k_wgrad_narrow_row<2> once per GAN step (1 warm-up + 2 timed, then 1 + 2 with the side streams off = 6)
gsteps = steps_of(os.path.join(P, tag + '_geo_kernel_stats.csv'), 'k_edge_reduce', 1, 7)
tsteps = steps_of(os.path.join(P, tag + '_tex_kernel_stats.csv'), 'k_wgrad_narrow_row<2>', 1, 6)
geo, gt = top(os.path.join(P, tag + '_geo_kernel_stats.csv'), gsteps)
tex, tt = top(os.path.join(P, tag + '_tex_kernel_stats.csv'), tsteps, n=24)
nonmfma = non_mfma_ms(os.path.join(P, tag + '_tex_kernel_stats.csv'), tsteps)
d = json.load(open(os.path.join(P, tag + '_bench.json')))
f, w = (json.load(open(os.path.join(P, '%s_pmc_geo_%s.json' % (tag, c)))) for c in ('FETCH_SIZE', 'WRITE_SIZE'))
tf, tw = (json.load(open(os.path.join(P, '%s_pmc_tex_%s.json' % (tag, c)))) for c in ('FETCH_SIZE', 'WRITE_SIZE'))
mb = lambda F, W, k: (2 * F[k]['FETCH_SIZE']['mean'] + W[k]['WRITE_SIZE']['mean']) * 1024 / 1e6
L = ['# %s -- validation of the tree at the commit that adds this file (one lease, `tools/gpu_round.sh %s`)\n' % (tag, tag),
     'GPU tests: `%s`; `__graft_entry__.smoke()`: see `%s_smoke.log`\n' % (tests_tail[-1] if tests_tail else '?', tag),
     '## Bench line (`%s_bench.json`)\n' % tag, '| | previous round (driver) | this run |\n|---|---|---|']
pv = (lambda k, fmt='%.1f': (fmt % prev[k]) if prev and k in prev else '--')
r, rf, ra, rt = d['roofline'], d['roofline_raster_fwd'], d['roofline_alu'], d['roofline_textural']
L.append('| rendered objects/s (16-object frame, fwd+bwd) | %s | **%.0f** (%.2f ms per frame) |' % (pv('value', '%.0f'), d['value'], d['ms_per_step']))
L.append('| textural GAN step, bs 4, 384x1248 | %s ms | %.1f ms |' % (pv('textural_gan_fwd_bwd_ms'), d['textural_gan_fwd_bwd_ms']))
L.append('| `%s` per launch | %s | %.0f us = %.0f GB/s algorithmic = %.1f %% of 8 TB/s; counters %.0f MB = %.0f %% of peak |' % (
    r['kernel'], ('%.0f us' % prev['roofline']['avg_launch_us']) if prev and 'avg_launch_us' in prev.get('roofline', {}) else '--', r['avg_launch_us'], r['achieved'], 100 * r['frac'],
    r['traffic'] / 1e6, 100 * r['traffic_frac_of_peak']))
L.append('| `k_raster_tiles` per launch | %s | %.0f us = %.0f GB/s algorithmic = %.1f %%; counters %.0f MB = %.0f %% of peak |' % (
    ('%.0f us' % prev['roofline_raster_fwd']['avg_launch_us']) if prev and 'roofline_raster_fwd' in prev else '--', rf['avg_launch_us'], rf['achieved'], 100 * rf['frac'],
    rf['traffic'] / 1e6, 100 * rf['traffic_frac_of_peak']))
L.append('| ALU view of `k_raster_tiles` | -- | %.1f M candidate pixel tests, %.1f M covered, %.2f GFLOP per launch = %.2f TFLOP/s = %.1f %% of the 157.3 TFLOP/s fp32 vector peak |' % (
    ra['candidate_pixel_tests'] / 1e6, ra['tests_passed'] / 1e6, ra['flops_per_launch'] / 1e9, ra['achieved'], 100 * ra['frac']))
L.append('| MFMA forward / data-gradient group (`k_conv_gemm` + `k_conv_tile` + `k_conv_halo`) | %s | %.1f TFLOP/s algorithmic = %.1f %% of 2.5 PFLOP/s (issued %.1f %%); HBM traffic %.0f MB per launch |' % (
    ('%.1f TFLOP/s' % prev['roofline_textural']['achieved']) if prev and 'roofline_textural' in prev else '--', rt['achieved'], 100 * rt['frac'], 100 * rt['issued_frac'], (rt['traffic'] or 0) / 1e6))
L.append('| MFMA weight gradients (`k_wgrad_tile` + `k_conv_wgrad`) | %s | %.1f TFLOP/s |' % (('%.1f TFLOP/s' % prev['roofline_textural']['wgrad']['achieved']) if prev and 'roofline_textural' in prev else '--', rt['wgrad']['achieved']))
if 'single_stream' in rt:
    ss = rt['single_stream']
    L.append('| the same kernels with the side streams off (every kernel alone on the chip; step %.1f ms) | -- | forward / data gradient %.1f TFLOP/s = %.1f %% (issued %.1f %%), weight gradients %.1f TFLOP/s |' % (
        ss['ms_per_step'], ss['achieved'], 100 * ss['frac'], 100 * ss['issued_frac'], ss['wgrad_achieved']))
d3, ep = d['derender3d_loop'], d['edit_pipeline']
L.append('| configs[2] (16 objects): encoder fwd / inference / 20-iteration optimisation / train step | -- | %.2f / %.2f / %.1f (%.2f per iteration, %.0f objects/s) / %.1f ms |' % (
    d3['encoder_fwd_ms'], d3['inference_ms'], d3['optimisation_ms'], d3['optimisation_ms_per_iteration'], d3['optimisation_objects_per_s'], d3['train_step_ms']))
sec = 'car_like' if 'car_like' in d else 'cad_like'
if d.get(sec) and 'objects_per_s' in d[sec]:
    c = d[sec]
    L.append('| the same frame step on `synth.' + sec + '` templates (secondary family) | -- | %.0f objects/s (%.2f ms; `k_raster_tiles` %.0f us, edge kernels %.0f us) |' % (
        c['objects_per_s'], c['ms_per_step'], c['k_raster_tiles_us'], c['edge_kernels_us']))
L.append('| host time to issue one step into an empty queue: frame / GAN step | -- | %.2f / %.1f ms |' % (
    d.get('host_issue_ms_one_step', float('nan')), d.get('textural', {}).get('host_issue_ms_one_step', float('nan'))))
L.append('| configs[4] (64 frames x 10 objects, one GPU) | -- | %.1f frames/s (%.1f ms per frame) |' % (ep['frames_per_s'], ep['ms_per_frame_per_gpu']))
L.append('| CPU oracle (%d threads), one object fwd+bwd | -- | %.1f s per object (%.3f objects/s; %s timed) |' % (d['cpu_baseline']['cores'], 1 / d['cpu_baseline']['value'], d['cpu_baseline']['value'], d['cpu_baseline'].get('timed_objects', '?')))
if 'cpu_baseline_textural' in d:
    L.append('| CPU oracle, textural G/D/E train step bs 1 192x624 | -- | %.1f s |' % (d['cpu_baseline_textural']['value'] / 1e3))
L.append('\n## Geometric leg, kernel time per step (`%s_geo_kernel_stats.csv`, %.2f ms summed; %d steps + one counting launch, headline mesh)\n' % (tag, gt, gsteps))
L.append('| kernel | launches / step | us / step | share |\n|---|---|---|---|')
L += geo
L.append('\n## Textural leg, kernel time per step (`%s_tex_kernel_stats.csv`, %.1f ms summed over %d steps -- half of them with, half without the side streams: summed durations exceed the step time where kernels overlap; kernels other than the MFMA conv kernels: **%.1f ms per step**)\n' % (tag, tt, tsteps, nonmfma))
L.append('| kernel | launches / step | us / step | share |\n|---|---|---|---|')
L += tex
L.append('\n## HBM counters (separate `--pmc FETCH_SIZE` / `WRITE_SIZE` passes; bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB per dispatch)\n')
L.append('| kernel | MB per launch |\n|---|---|')
for k in ('sdn::k_raster_tiles', 'sdn::k_edge_scan_sil', 'sdn::k_edge_rows', 'sdn::k_chunk_sum', 'sdn::k_edge_reduce', 'sdn::k_tile_fill',
          'sdn::k_edge_plan', 'sdn::k_compact_rows'):
    if k in f and k in w:
        L.append('| `%s` | %.0f |' % (k, mb(f, w, k)))
for k in ('sdn::k_conv_gemm', 'sdn::k_conv_tile', 'sdn::k_conv_halo', 'sdn::k_wgrad_tile', 'sdn::k_conv_wgrad'):
    if k in tf and k in tw:
        L.append('| `%s` (mean over all launches of a step) | %.0f |' % (k, mb(tf, tw, k)))
shapes = set(tf.get('_by_shape', {})) & set(tw.get('_by_shape', {}))
if shapes:
    L.append('\n### MFMA conv kernels by layer shape (template arguments + launch grid; the 12 with the most traffic)\n')
    L.append('| kernel instance | dispatches | MB per launch |\n|---|---|---|')
    rows = sorted(((mb(tf['_by_shape'], tw['_by_shape'], k), k) for k in shapes), reverse=True)[:12]
    for v, k in rows:
        L.append('| `%s` | %d | %.0f |' % (k.replace('sdn::', '')[:110], tf['_by_shape'][k]['FETCH_SIZE']['dispatches'], v))
open(os.path.join(P, tag + '_summary.md'), 'w').write('\n'.join(L) + '\n')
print('\n'.join(L[:16]))
