#!/usr/bin/env python
"""Condense rocprofv3 output directories into the small text/JSON summaries kept under profiles/.

    python scripts/summarize_prof.py stats  <dir> <out.txt>     # --kernel-trace --stats run
    python scripts/summarize_prof.py pmc    <fetch_dir> <write_dir> <out.json> [<traffic.json>]

stats: per-kernel calls / total / average duration from *kernel_trace.csv (or *kernel_stats.csv).
pmc:   per-kernel mean FETCH_SIZE / WRITE_SIZE per dispatch -> HBM bytes per launch.  gfx950
       corrections from /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counters are in KiB;
       FETCH_SIZE reads exactly half of a wide coalesced stream, so it is doubled; WRITE_SIZE is
       taken as is (uncalibrated, stated in the output).
"""
import glob
import json
import os
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    m = re.search(r'conv_igemm_kernel<dir::(\w+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\w+), (\w+)>', name)
    if m:   # <DT, BM, BN, WGM, WGN, NST, BK, CIN16, SPLITK> -> the variant names of csrc/conv_igemm.hip
        dt, bm, bn, wm, wn, nst, bk, c16, sk = m.groups()
        return 'conv_igemm<%sx%s_w%sx%s%s%s%s>%s[%s]' % (bm, bn, wm, wn, '_s' + nst if nst != '2' else '',
                                                        '_k' + bk if bk != '64' else '',
                                                        '/splitk' if sk == 'true' else '',
                                                        '/stem' if c16 == 'true' else '', dt.lower())
    m = re.search(r'conv1x1_persist_kernel<dir::(\w+)>', name)
    if m:
        return 'conv_igemm<256x256_persist1x1>[%s]' % m.group(1).lower()
    m = re.search(r'conv_patch3x3_kernel<dir::(\w+), (\d+), (\d+)', name)
    if m:
        return 'conv_igemm<256x%s_patch3x3>[%s]' % (m.group(2), m.group(1).lower())
    m = re.search(r'conv1x1_wreg_kernel<dir::(\w+), (\d+)>', name)
    if m:
        return 'conv_igemm<64x512_wreg1x1>[%s]' % m.group(1).lower()
    m = re.search(r'dir::(\w+)', name)
    if m:
        return m.group(1)
    m = re.search(r'(\w+)(<|\()', name.replace('void ', ''))
    return ('torch:' + m.group(1))[:50] if m else name[:50]


def db(d):
    r = glob.glob(os.path.join(d, '**', '*.db'), recursive=True)
    if not r:
        sys.exit('no rocprofv3 .db under ' + d)
    return sqlite3.connect(r[0])


def stats(d, out):
    agg = defaultdict(lambda: [0, 0.0])
    for name, dur in db(d).execute('select name, duration from kernels'):
        k = short(name)
        agg[k][0] += 1
        agg[k][1] += dur / 1e3
    tot = sum(v[1] for v in agg.values())
    with open(out, 'w') as fh:
        fh.write('# rocprofv3 --kernel-trace --stats summary (rocpd "kernels" view of %s)\n' % d)
        fh.write('%-52s %8s %12s %10s %6s\n' % ('kernel', 'calls', 'total_us', 'avg_us', '%'))
        for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            fh.write('%-52s %8d %12.1f %10.2f %6.2f\n' % (k, n, us, us / n, 100 * us / tot))
    print(open(out).read())


def pmc_means(d, counter):
    agg = defaultdict(lambda: [0, 0.0])
    for name, val in db(d).execute('select name, counter_value from pmc_events where counter_name = ?', (counter,)):
        k = short(name)
        agg[k][0] += 1
        agg[k][1] += float(val)
    return {k: (n, v / n) for k, (n, v) in agg.items()}


def pmc(fd, wd, out, traffic=None):
    fe, wr = pmc_means(fd, 'FETCH_SIZE'), pmc_means(wd, 'WRITE_SIZE')
    res = {}
    for k in sorted(set(fe) | set(wr)):
        f = fe.get(k, (0, 0.0))
        w = wr.get(k, (0, 0.0))
        res[k] = {'dispatches': max(f[0], w[0]), 'FETCH_SIZE_KiB_mean': f[1], 'WRITE_SIZE_KiB_mean': w[1],
                  'hbm_read_bytes_per_launch': 2 * f[1] * 1024, 'hbm_write_bytes_per_launch': w[1] * 1024,
                  'hbm_bytes_per_launch': (2 * f[1] + w[1]) * 1024}
    json.dump({'note': 'FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B, MI355X_MICROARCH.md HBM); '
                       'WRITE_SIZE uncalibrated; Infinity-Cache hits are counted, not excluded',
               'kernels': res}, open(out, 'w'), indent=1)
    if traffic:
        # bench.py names kernels "conv_igemm<VARIANT>" without dtype / stem suffix
        t = {}
        for k, v in res.items():
            t[re.sub(r'(/stem)?\[\w+\]$', '', k)] = v['hbm_bytes_per_launch']
        json.dump(t, open(traffic, 'w'), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['dispatches'])[:14]:
        print('%-52s n=%5d  read %8.1f MB  write %8.1f MB' % (k, v['dispatches'], v['hbm_read_bytes_per_launch'] / 1e6,
                                                              v['hbm_write_bytes_per_launch'] / 1e6))


if __name__ == '__main__':
    if sys.argv[1] == 'stats':
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(*sys.argv[2:])
