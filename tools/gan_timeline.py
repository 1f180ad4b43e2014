#!/usr/bin/env python3
"""Where is the GPU idle, and how much do kernels overlap, inside one multi-stream GAN step?  Reads a rocprofv3 kernel TRACE
(kernel_trace.csv: start / end timestamps per dispatch) of tools/prof_gan.py and prints, for the last step: wall time, the
union of the busy intervals, the idle gaps attributed to the kernel that ends them, and the time during which >= 2 kernels run.

    python tools/gan_timeline.py <kernel_trace.csv> <steps>
"""
import collections
import csv
import sys


def main(path, steps):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:60], r.get('Queue_Id', '?')))
    rows.sort()
    # the steps launch identical kernel sequences: take the last 1/steps of the dispatches
    n = len(rows) // steps
    last = rows[-n:]
    t0, t1 = last[0][0], max(r[1] for r in last)
    print('dispatches per step %d; wall time of the last step %.2f ms' % (n, (t1 - t0) / 1e6))
    busy, cur_end, gaps, overlap = 0, t0, collections.Counter(), 0
    events = []
    for s, e, name, q in last:
        events.append((s, 1))
        events.append((e, -1))
        if s > cur_end:
            if s - cur_end > 2000:
                gaps[name] += s - cur_end
            busy += 0
            cur_end = s
        if e > cur_end:
            busy += e - max(s, cur_end)
            cur_end = e
    events.sort()
    depth, prev = 0, events[0][0]
    depth_time = collections.Counter()
    for t, d in events:
        depth_time[depth] += t - prev
        prev = t
        depth += d
    print('busy (union) %.2f ms; idle %.2f ms' % (busy / 1e6, (t1 - t0 - busy) / 1e6))
    print('time with k kernels in flight: ' + ', '.join('%d: %.2f ms' % (k, v / 1e6) for k, v in sorted(depth_time.items())))
    print('idle gaps > 2 us, by the kernel that ends them (top 15):')
    for name, v in gaps.most_common(15):
        print('  %8.1f us  %s' % (v / 1e3, name))
    queues = collections.Counter()
    for s, e, name, q in last:
        queues[q] += e - s
    print('kernel time per queue: ' + ', '.join('%s: %.1f ms' % (q, v / 1e6) for q, v in queues.most_common()))
    # r06: a coarse Gantt chart -- per millisecond of the step and queue the kernel that holds most of the bin (share of the bin)
    qs = [q for q, _ in queues.most_common()]
    nb = int((t1 - t0) / 1e6) + 1
    bins = [[collections.Counter() for _ in qs] for _ in range(nb)]
    for s, e, name, q in last:
        b0, b1 = int((s - t0) / 1e6), int((e - t0) / 1e6)
        for b in range(b0, min(b1, nb - 1) + 1):
            lo, hi = max(s, t0 + b * 1000000), min(e, t0 + (b + 1) * 1000000)
            if hi > lo:
                bins[b][qs.index(q)][name.replace('void sdn::', '').replace('sdn::', '')[:26]] += hi - lo
    print('ms   ' + ' | '.join('queue %-29s' % q for q in qs))
    for b in range(nb):
        cells = []
        for c in bins[b]:
            if c:
                name, v = c.most_common(1)[0]
                cells.append('%-26s %3d%% %2d%%' % (name, 100 * v / 1e6, 100 * sum(c.values()) / 1e6))
            else:
                cells.append(' ' * 35)
        print('%3d  ' % b + ' | '.join(cells))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]))
