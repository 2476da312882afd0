#!/usr/bin/env python
"""Condense rocprofv3 output directories into the small text/JSON summaries kept under profiles/.

    python scripts/summarize_prof.py stats  <dir> <out.txt>     # --kernel-trace --stats run
    python scripts/summarize_prof.py pmc    <fetch_dir> <write_dir> <out.json> [<traffic.json>]
    python scripts/summarize_prof.py table  <launches.json> <stats_dir> <sq_dir> <fetch_dir> <write_dir> \
                                            <out.txt> [<traffic.json>]

stats: per-kernel calls / total / average duration from *kernel_trace.csv (or *kernel_stats.csv).
table: the per-kernel ROOFLINE table.  launches.json (bench.py --dump-launches) is the launch
       sequence of one forward with algorithmic flops / bytes per launch; rocprofv3 dispatches of the
       engine's kernels are aligned with it by position (every forward issues the same sequence), so
       each (kernel, layer shape) group gets its own duration, HBM traffic and MFMA-busy figures:
         MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCCs * 256 CUs * 4 SIMDs)   (busy
         cycles are summed over all SIMDs, 32 per 32x32x16 MFMA; GRBM_GUI_ACTIVE is summed over the
         8 XCCs - both checked against SQ_INSTS_MFMA and the kernel duration), clock = GUI_ACTIVE per
         XCC / duration.
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


def csrc_hash():
    """sha256 over the kernel sources (csrc/*.hip, *.h, include/dir_engine.h): profiles/traffic.json carries it as
    '_build', and bench.py drops `roofline.traffic` when the library it runs was built from other sources - PMC
    bytes of an older kernel are not a measurement of this one."""
    import glob
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(root, 'deep-image-retrieval_amd', 'csrc', '*.hip')) +
                   glob.glob(os.path.join(root, 'deep-image-retrieval_amd', 'csrc', '*.h')) +
                   [os.path.join(root, 'include', 'dir_engine.h')])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def short(name):
    m = re.search(r'conv_igemm_kernel<dir::(\w+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\w+), (\w+)(?:, (\w+))?>', name)
    if m:   # <DT, BM, BN, WGM, WGN, NST, BK, CIN16, SPLITK[, DUAL]> -> the variant names of csrc/conv_igemm.hip
        dt, bm, bn, wm, wn, nst, bk, c16, sk, dual = m.groups()
        return 'conv_igemm<%sx%s_w%sx%s%s%s%s%s>%s[%s]' % (bm, bn, wm, wn, '_s' + nst if nst != '2' else '',
                                                          '_k' + bk if bk != '64' else '',
                                                          '/splitk' if sk == 'true' else '',
                                                          '/dual' if dual == 'true' else '',
                                                          '/stem' if c16 == 'true' else '', dt.lower())
    m = re.search(r'conv_pair_kernel<(\d+), (\d+), (\d+), (\d+), (\w+)(?:, (\w+))?>', name)
    if m:   # the paired-fp16 head of DIR_FP16P (csrc/conv_pair.hip): <BM, BN, WGM, WGN, XP[, DUAL]>
        return 'conv_pair<%sx%s_%sw%s>' % (m.group(1), m.group(2), 'x' if m.group(5) == 'true' else '',
                                           '/dual' if m.group(6) == 'true' else '')
    if 'conv_pair_patch64_kernel' in name:   # layer1's 3x3 on a patch pair (csrc/conv_pair.hip)
        return 'conv_pair<128x64_patch3x3_xw>'
    m = re.search(r'conv_f32_kernel<(\d+)>', name)
    if m:   # the strict fp32 path (csrc/conv_f32.hip)
        return 'conv_f32<128x%s>' % m.group(1)
    m = re.search(r'conv_c3c1_kernel<dir::(\w+), (\d+), (\w+), (\d+)(?:, (\w+), (\w+))?>', name)
    if m:   # <DT, P, DS, P2[, WP3, WP1]>: wp = paired weights (DIR_FP16P)
        return 'conv_c3c1<%s%s%s>[%s]' % (m.group(2), ',ds' if m.group(3) == 'true' else '',
                                          ',wp' if m.group(5) == 'true' else '', m.group(1).lower())
    m = re.search(r'conv_c3c1ds_lc_kernel<dir::(\w+), (\w+)>', name)
    if m:   # the DS seam with loader / consumer roles (csrc/conv_c3c1lc.hip): <DT, WP>
        return 'conv_c3c1<64,ds%s>[%s]' % (',wp' if m.group(2) == 'true' else '', m.group(1).lower())
    m = re.search(r'conv_seam3_kernel<dir::(\w+)>', name)
    if m:   # the layer3 seam (csrc/conv_seam3.hip), opt-in
        return 'conv_seam3<256>[%s]' % m.group(1).lower()
    m = re.search(r'conv_patch64_lc_kernel<dir::(\w+)>', name)
    if m:
        return 'conv_igemm<256x64_patchlc3x3>[%s]' % m.group(1).lower()
    m = re.search(r'conv1x1_ring_kernel<dir::(\w+)>', name)
    if m:
        return 'conv_igemm<128x256_ring1x1>[%s]' % m.group(1).lower()
    m = re.search(r'conv1x1_persist_kernel<dir::(\w+)(?:, (\w+), (\w+))?(?:, (\w+))?>', name)
    if m:   # <DT, XDEEP, RES[, DUAL]>
        return 'conv_igemm<256x256_persist1x1%s%s>[%s]' % ('_x3' if m.group(2) == 'true' else '',
                                                           '/dual' if m.group(4) == 'true' else '', m.group(1).lower())
    m = re.search(r'conv_patch3x3s2_kernel<dir::(\w+)>', name)   # the stride-2 patch kernel (csrc/conv_patchs2.hip, round 6)
    if m:
        return 'conv_igemm<256x128_patchs2>[%s]' % m.group(1).lower()
    m = re.search(r'conv_patch3x3s_kernel<dir::(\w+), (\d+)>', name)
    if m:
        return 'conv_igemm<256x256_patch3x3s>[%s]' % m.group(1).lower()
    m = re.search(r'conv_patch3x3w(?:_lc)?_kernel<dir::(\w+)>', name)   # (one-role form and, round 5, the loader / consumer form)
    if m:
        return 'conv_igemm<512x128_patch3x3w>[%s]' % m.group(1).lower()
    m = re.search(r'conv_patch3x3_kernel<dir::(\w+), (\d+), (\d+)', name)
    if m:
        return 'conv_igemm<256x%s_patch3x3>[%s]' % (m.group(2), m.group(1).lower())
    m = re.search(r'conv1x1_wreg_kernel<dir::(\w+), (\d+)>', name)
    if m:
        return 'conv_igemm<64x512_wreg1x1>[%s]' % m.group(1).lower()
    m = re.search(r'conv_small_kernel<dir::(\w+), (\d+), (\d+)>', name)
    if m:   # small maps: 64 x 64 tiles, loader / consumer waves on LDS counters (csrc/conv_small.hip): <DT, NST, KPS>
        return 'conv_igemm<64x64_small_s%s%s>[%s]' % (m.group(2), 'k2' if m.group(3) == '2' else '', m.group(1).lower())
    m = re.search(r'conv1x1_lc_kernel<dir::(\w+), (\w+)>', name)
    if m:   # the deep-X ring with loader / consumer roles (csrc/conv_persistlc.hip): <DT, DUAL>
        return 'conv_igemm<256x256_lc1x1%s>[%s]' % ('/dual' if m.group(2) == 'true' else '', m.group(1).lower())
    m = re.search(r'conv1x1_wregd_kernel<dir::(\w+), (\d+), (\d+)>', name)
    if m:   # the two-source register-stationary GEMM of layer2's first block (csrc/conv_wregd.hip)
        return 'conv_igemm<64x256_wregd1x1/dual>[%s]' % m.group(1).lower()
    m = re.search(r'stem_pool_u8_kernel<(\d+), (\w+)(?:, (\w+))?>', name)
    if m:   # <NRP, RAW[, XPAIR]> (csrc/stem_u8.hip): XPAIR = the generic paired stem on the walking kernel
        return 'stem_pool_pair_kernel' if m.group(3) == 'true' else 'stem_pool_u8_kernel'
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


# ---- per-kernel roofline table ---------------------------------------------------------------------
ENGINE_PREFIX = ('conv_igemm<', 'conv_c3c1<', 'conv_seam3<', 'conv_f32<', 'conv_pair<', 'stem_pool', 'prep_input', 'global_pool', 'gemm_nt', 'maxpool', 'upsample_add')
PEAK_TF, PEAK_GBS, NXCC, NSIMD = 2500.0, 8000.0, 8, 1024
FOLLOW_UP = ('gemm_splitk_finalize_kernel', 'conv_splitk_finalize_kernel')
MEASURED_TF, MEASURED_GBS = 1582.0, 6305.0   # scripts/probes/*_ceiling.hip on a pool box (profiles/r02_*_ceiling.txt)


def bench_kernel_name(k):
    """rocprof short name -> the name bench.py's profile records carry."""
    k = re.sub(r'(/stem)?\[\w+\]$', '', k)
    return {'stem_pool_kernel': 'stem_pool', 'stem_pool_persist_kernel': 'stem_pool', 'prep_input_kernel': 'prep_input',
            'global_pool_kernel': 'global_pool',
            'gemm_nt_small_kernel': 'gemm_nt_f32', 'gemm_nt_f32_kernel': 'gemm_nt_f32',
            'maxpool_kernel': 'maxpool_3x3s2', 'upsample_add_kernel': 'upsample_add',
            'prep_input_f32_kernel': 'prep_input_f32', 'maxpool_f32_kernel': 'maxpool_f32',
            'global_pool_f32_kernel': 'global_pool_f32', 'upsample_add_f32_kernel': 'upsample_add_f32',
            'stem_pool_pair_kernel': 'stem_pool_pair', 'stem_pool_pair_persist_kernel': 'stem_pool_pair',
            'prep_input_pair_kernel': 'prep_input_pair',
            'stem_pool_u8_kernel': 'stem_pool_u8', 'prep_input_u8_kernel': 'prep_input_u8'}.get(k, k)


def layer_group(name):
    parts = name.split('.')
    if len(parts) == 3 and parts[0].startswith('layer') and parts[1] != '0':
        return parts[0] + '.' + parts[2]
    return name


def aligned(d, seq, counters=None):
    """[(launch index in seq, duration ns, {counter: value})] for every engine dispatch of database d."""
    con = db(d)
    rows = []
    for disp, name, dur in con.execute('select dispatch_id, name, duration from kernels order by dispatch_id'):
        k = bench_kernel_name(short(name))
        if k.startswith(ENGINE_PREFIX):
            rows.append((disp, k, dur))
    vals = defaultdict(dict)
    if counters:
        for disp, cn, v in con.execute('select dispatch_id, counter_name, counter_value from pmc_events'):
            vals[disp][cn] = vals[disp].get(cn, 0.0) + float(v)    # per-XCC / per-SE instances: summed
    out, pos = [], 0
    L = len(seq)
    for disp, k, dur in rows:
        want = seq[pos % L]['kernel']
        if k in FOLLOW_UP and out:     # second kernel of a launch record (split-K finalize): its time joins the record
            i, d0, v0 = out[-1]
            out[-1] = (i, d0 + dur, v0)
            continue
        if k != want:
            # tolerate kernels outside the recorded sequence (e.g. a warm-up autotune); resync on the head
            if k == seq[0]['kernel']:
                pos = (pos // L + 1) * L if pos % L else pos
            else:
                continue
        out.append((pos % L, dur, vals.get(disp, {})))
        pos += 1
    return out


def table(launches, stats_d, sq_d, fetch_d, write_d, out, traffic=None):
    seq = json.load(open(launches))
    groups = defaultdict(lambda: defaultdict(float))

    def key(i):
        r = seq[i]
        return (r['kernel'], layer_group(r['name']), r['flops'], r['bytes'])

    for i, dur, _ in aligned(stats_d, seq):
        g = groups[key(i)]
        g['n'] += 1
        g['us'] += dur / 1e3
    for d in (sq_d, fetch_d, write_d):
        if not d or not os.path.isdir(d):
            continue
        tag = os.path.basename(d.rstrip('/'))
        for i, dur, v in aligned(d, seq, True):
            g = groups[key(i)]
            g['n_' + tag] += 1
            g['us_' + tag] += dur / 1e3
            for cn, x in v.items():
                g[cn] += x
                g['n_' + cn] += 1
    nstep = max(1, min(int(g['n']) for g in groups.values() if g['n']))
    tot_us = sum(g['us'] for g in groups.values())
    tot_fl = sum(k[2] * g['n'] for k, g in groups.items())
    lines = ['# per-kernel roofline table: rocprofv3 --kernel-trace durations + --pmc passes aligned with bench.py\'s',
             '# launch sequence (scripts/summarize_prof.py table).  bound: AI vs 2.5 PF / 8 TB/s = 312 FLOP/B;',
             '# frac = achieved / that roof; traffic = PMC HBM bytes (2 x FETCH_SIZE + WRITE_SIZE, KiB) / algorithmic bytes;',
             '# MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs); clk = GRBM_GUI_ACTIVE/8 / duration.',
             '%-34s %-20s %5s %8s %6s %9s %9s %8s %8s %5s %6s %8s %8s %6s' % (
                 'kernel', 'layers', 'n/st', 'avg_us', 'share', 'GFLOP', 'MB', 'TF/s', 'GB/s', 'bound', 'frac',
                 'traffic', 'MfmaUtil', 'clkGHz')]
    tr = defaultdict(dict)
    for k, g in sorted(groups.items(), key=lambda kv: -kv[1]['us']):
        kern, grp, fl, by = k
        if not g['n']:
            continue
        us = g['us'] / g['n']
        tf = fl / us / 1e6 if fl else 0.0
        gbs = by / us / 1e3 if by else 0.0
        hbm = (fl / by) < (PEAK_TF * 1e12) / (PEAK_GBS * 1e9) if (fl and by) else True
        frac = gbs / PEAK_GBS if hbm else tf / PEAK_TF
        traffic_b = None
        if g['n_FETCH_SIZE'] and g['n_WRITE_SIZE']:
            traffic_b = (2 * g['FETCH_SIZE'] / g['n_FETCH_SIZE'] + g['WRITE_SIZE'] / g['n_WRITE_SIZE']) * 1024
            tr[kern][grp] = traffic_b
        mfma = clk = None
        if g['n_GRBM_GUI_ACTIVE'] and g['n_SQ_VALU_MFMA_BUSY_CYCLES']:
            gui = g['GRBM_GUI_ACTIVE'] / g['n_GRBM_GUI_ACTIVE'] / NXCC
            mfma = g['SQ_VALU_MFMA_BUSY_CYCLES'] / g['n_SQ_VALU_MFMA_BUSY_CYCLES'] / (gui * NSIMD)
            tagus = g['us_' + os.path.basename(sq_d.rstrip('/'))] / max(1, g['n_' + os.path.basename(sq_d.rstrip('/'))])
            clk = gui / (tagus * 1e3)
        lines.append('%-34s %-20s %5.1f %8.1f %6.3f %9.2f %9.1f %8.1f %8.1f %5s %6.3f %8s %8s %6s' % (
            kern, grp, g['n'] / nstep, us, g['us'] / tot_us, fl / 1e9, by / 1e6, tf, gbs, 'hbm' if hbm else 'mfma', frac,
            '%.2fx' % (traffic_b / by) if traffic_b and by else '-', '%.3f' % mfma if mfma is not None else '-',
            '%.2f' % clk if clk is not None else '-'))
    lines.append('# step: %.1f us of engine kernels per forward, %.1f TFLOP/s = %.3f of the 2.5 PF dense MFMA peak' % (
        tot_us / nstep, tot_fl / tot_us / 1e6, tot_fl / tot_us / 1e6 / PEAK_TF))
    # the same step against per-kernel floors max(flops / MFMA rate, bytes / HBM rate): at the guide's peaks and at the
    # rates this pool's boxes deliver to kernels that do nothing else (profiles/r02_mfma_ceiling.txt, r02_hbm_ceiling.txt)
    def floors(tf_rate, gb_rate):
        return sum(g['n'] * max(k[2] / (tf_rate * 1e6), k[3] / (gb_rate * 1e3)) for k, g in groups.items() if g['n']) / nstep
    f_peak, f_meas = floors(PEAK_TF, PEAK_GBS), floors(MEASURED_TF, MEASURED_GBS)
    lines.append('# sum of per-kernel floors: %.1f us at 2.5 PF / 8 TB/s = %.3f of the measured step; %.1f us at the measured ceilings '
                 '%.2f PF (random operands) / %.1f TB/s (read) = %.3f of the measured step' % (
                     f_peak, f_peak / (tot_us / nstep), f_meas, MEASURED_TF / 1e3, MEASURED_GBS / 1e3, f_meas / (tot_us / nstep)))
    open(out, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))
    if traffic:
        tr = dict(tr)
        tr['_build'] = csrc_hash()
        json.dump(tr, open(traffic, 'w'), indent=1)


if __name__ == '__main__':
    if sys.argv[1] == 'table':
        table(*sys.argv[2:])
    elif sys.argv[1] == 'stats':
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(*sys.argv[2:])
