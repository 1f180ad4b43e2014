"""kernel_stats.csv of tools/prof_gan.py (K identical steps) -> per-step table, kernels grouped by what they are for.
usage: gan_step_table.py <kernel_stats.csv> <K>"""
import csv
import sys

GROUPS = [
    ('MFMA forward / data gradient', ('k_conv_gemm', 'k_conv_tile')),
    ('MFMA weight gradient', ('k_conv_wgrad', 'k_wgrad_tile')),
    ('fp32 head layers (narrow)', ('k_conv_narrow', 'k_wgrad_narrow')),
    ('InstanceNorm forward', ('k_in_finalize', 'k_in_apply')),
    ('InstanceNorm / activation backward', ('k_in_bwd', 'k_act_bwd')),
    ('operand planes (standalone split)', ('k_split_planes',)),
    ('weight pack / gradient unpack', ('k_pack_weights', 'k_unpack_grad')),
    ('split-K tails, reflect fold, add, colsum', ('k_split_reduce', 'k_bias_act_stats', 'k_reflect_fold', 'k_add', 'k_colsum')),
    ('losses, pooling, segment ops', ('k_l1', 'k_avgpool', 'k_segment', 'k_maxpool')),
    ('copies / fills (runtime)', ('copyBuffer', 'fillBuffer')),
]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    K = int(sys.argv[2])
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print('kernel time per step (sum over kernels, single stream): %.2f ms; %d launches per step' % (
        tot / 1e6 / K, sum(int(r['Calls']) for r in rows) // K))
    print()
    print('| group | launches / step | ms / step | share |')
    print('|---|---|---|---|')
    seen = set()
    agg = []
    for name, keys in GROUPS:
        t = c = 0
        for r in rows:
            if any(k in r['Name'] for k in keys) and r['Name'] not in seen:
                seen.add(r['Name'])
                t += float(r['TotalDurationNs'])
                c += int(r['Calls'])
        agg.append((name, c, t))
    t = sum(float(r['TotalDurationNs']) for r in rows if r['Name'] not in seen)
    c = sum(int(r['Calls']) for r in rows if r['Name'] not in seen)
    agg.append(('everything else (torch element-wise, Adam, ...)', c, t))
    for name, c, t in agg:
        print('| %s | %.1f | %.2f | %.1f %% |' % (name, c / K, t / 1e6 / K, 100 * t / tot))
    print()
    print('| kernel | launches / step | us / launch | ms / step |')
    print('|---|---|---|---|')
    for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:32]:
        print('| `%s` | %.1f | %.1f | %.2f |' % (r['Name'][:70], int(r['Calls']) / K, float(r['AverageNs']) / 1e3,
                                                 float(r['TotalDurationNs']) / 1e6 / K))


if __name__ == '__main__':
    main()
