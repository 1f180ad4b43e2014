#!/usr/bin/env python3
"""Aggregates a tests/gpu_layer_times.py log: one line per (network, pass, layer shape) with count, summed ms and TFLOP/s."""
import collections
import re
import sys


def main(path, min_ms=0.25):
    agg, sect = collections.OrderedDict(), None
    for ln in open(path):
        if ln.startswith('===='):
            sect = ln.split()[1]
            continue
        m = re.match(r'(\S+)\s+(.*?)\s+([\d.]+) ms\s*(?:([\d.]+) TFLOP/s)?', ln)
        if not m:
            if ln.startswith('totals'):
                print(sect, ln.strip())
            continue
        a = agg.setdefault((sect, m.group(1), m.group(2).strip()), [0, 0.0, 0.0])
        a[0] += 1
        a[1] += float(m.group(3))
        a[2] += float(m.group(3)) * float(m.group(4)) if m.group(4) else 0
    for k, (n, ms, fl) in agg.items():
        if ms < min_ms:
            continue
        print('%-2s %-8s %-46s x%-3d %7.3f ms %s' % (k[0], k[1], k[2], n, ms, '%6.1f TF' % (fl / ms) if fl else ''))


if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.25)
