#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output: per-kernel mean of every PMC counter (counter_collection.csv) or the kernel stats.
usage: pmc_summary.py <dir> [name-filter] [tag]   -> JSON on stdout (with the sha256 of the libsdn_hip.so it ran on).  Used on the GPU box so only the summary travels back.
Kernel names are normalised to `namespace::name` (no `void `, template or argument lists); `_tag` records the run."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

import hashlib

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
out = {'_tag': sys.argv[3]} if len(sys.argv) > 3 else {}
# which build the counters belong to: bench.py flags the numbers `traffic_stale` when the library has changed since
_lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), '3d-sdn_amd', 'lib', 'libsdn_hip.so')
if os.path.exists(_lib):
    out['_lib_sha256'] = hashlib.sha256(open(_lib, 'rb').read()).hexdigest()


def norm(k):
    k = k[5:] if k.startswith('void ') else k
    for ch in '<(':
        k = k.split(ch)[0]
    return k.strip()


def shape_key(row):
    """the MFMA conv kernels serve every layer: template arguments + launch grid tell the layer shapes apart"""
    k = row.get('Kernel_Name', '')
    k = k[5:] if k.startswith('void ') else k
    return '%s grid %s' % (k.split('(')[0].strip(), row.get('Grid_Size', '?'))


for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    shapes = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get('Kernel_Name', '')
            if flt and flt not in k:
                continue
            a = acc[norm(k)][row['Counter_Name']]
            a[0] += float(row['Counter_Value'])
            a[1] += 1
            if any(m in k for m in ('k_conv_gemm', 'k_conv_wgrad', 'k_conv_tile', 'k_conv_halo', 'k_wgrad_tile')):
                a = shapes[shape_key(row)][row['Counter_Name']]
                a[0] += float(row['Counter_Value'])
                a[1] += 1
    if shapes:   # the 24 shapes with the most counter volume
        top = sorted(shapes.items(), key=lambda kv: -max(v[0] for v in kv[1].values()))[:24]
        out.setdefault('_by_shape', {})
        for k, cs in top:
            out['_by_shape'].setdefault(k, {}).update({c: {'mean': v[0] / v[1], 'dispatches': v[1]} for c, v in cs.items()})
    for k, cs in acc.items():
        out.setdefault(k, {}).update({c: {'mean': v[0] / v[1], 'dispatches': v[1]} for c, v in cs.items()})
print(json.dumps(out, indent=1))
