#!/usr/bin/env python
"""Per-kernel means of every counter found in one or more rocprofv3 --pmc output dirs.
    python scripts/pmc_table.py <dir> [<dir> ...]
Counters with several instances per dispatch (per-XCC/SE rows) are summed per dispatch first."""
import glob
import os
import sqlite3
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_prof import short  # noqa: E402

tab = defaultdict(dict)
for d in sys.argv[1:]:
    db = glob.glob(os.path.join(d, '**', '*.db'), recursive=True)[0]
    con = sqlite3.connect(db)
    per = defaultdict(float)
    kname = {}
    dur = {}
    for name, disp, cn, val, du in con.execute(
            'select name, dispatch_id, counter_name, counter_value, duration from pmc_events'):
        per[(disp, cn)] += float(val)
        kname[disp] = short(name)
        dur[disp] = du
    agg = defaultdict(lambda: [0, 0.0])
    for (disp, cn), v in per.items():
        a = agg[(kname[disp], cn)]
        a[0] += 1
        a[1] += v
    dd = defaultdict(lambda: [0, 0.0])
    for disp, du in dur.items():
        dd[kname[disp]][0] += 1
        dd[kname[disp]][1] += du
    for (k, cn), (n, v) in agg.items():
        tab[k][cn] = v / n
    for k, (n, v) in dd.items():
        tab[k]['n'] = n
        tab[k]['us(' + os.path.basename(d.rstrip('/')) + ')'] = v / n / 1e3

cols = sorted({c for v in tab.values() for c in v})
for k in sorted(tab, key=lambda k: -tab[k].get('n', 0)):
    if not k.startswith('conv_igemm') and not k.endswith('_kernel'):
        continue
    print(k)
    for c in cols:
        if c in tab[k]:
            print('    %-34s %16.1f' % (c, tab[k][c]))
