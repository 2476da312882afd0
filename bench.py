#!/usr/bin/env python
"""bench.py - images/sec of descriptor extraction (ResNet101-GeM, 1024x1024) on N MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (dir_forward: prep -> trunk -> GeM -> FC -> L2) over one batch
of synthetic normalised images that is already resident in HBM.  Database images are sharded
image-parallel over the ranks (no data-path collective inside a step); after the K steps each rank
all-gathers its shard's descriptor block once over RCCL/xGMI (the one exchange step the path has,
inside the timed region for N > 1).  Weights are the deterministic synthetic checkpoint of
tests/synth.py (there is no network for real ones); oracle/ is imported by the cpu_baseline leg only.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      the kernel with the largest time share (bound / achieved / peak / frac / traffic) AND
                `kernels`: the same figures for every (kernel, layer shape) group of the step, plus the
                step-level MFMA fraction - algorithmic FLOPs and bytes / event-timed launch durations
                collected inside this process over the timed steps
  cpu_baseline  the CPU oracle (a port of the reference forward) timed on this box's host cores
                on a bounded sample of the same workload (rank 0, N == 1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, 'deep-image-retrieval_amd'), os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_TFLOPS = {'bf16': 2500.0, 'fp16': 2500.0}   # dense MFMA, MI355X_MICROARCH.md (no sparsity)
PEAK_HBM_GBS = 8000.0                            # HBM3E spec (6.29 TB/s measured copy)
GFLOP_PER_IMG = {('resnet101', 1024): 325.99, ('resnet50', 224): 8.183}


def cpu_allotted():
    """CPUs this process may actually use: the cgroup quota when there is one, else the affinity mask."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            avail = min(avail, max(1, int(round(float(quota) / float(period)))))
    except (OSError, ValueError):
        pass
    return avail


def cpu_baseline(arch, size, budget_s):
    """CPU oracle forward, batch 1 (the reference's default path, test_dir.py:52-55), all cores."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import dir_oracle as O          # the only oracle import of this file: the timed CPU port
    import synth
    sd = synth.synth_state_dict(arch, seed=7)
    # thread count: the box may expose more logical CPUs than its cgroup lets run; pick the
    # fastest of a few counts on a probe at half the image side instead of trusting nproc
    allotted = cpu_allotted()
    try:
        visible = len(os.sched_getaffinity(0))
    except AttributeError:
        visible = os.cpu_count() or 1
    probe = synth.synth_images(11, 1, max(size // 2, 64), max(size // 2, 64))
    best = (float('inf'), 1)
    for nt in sorted({min(visible, c) for c in (allotted // 2 or 1, allotted, 2 * allotted)}):
        torch.set_num_threads(nt)
        O.rmac_forward(sd, arch, probe)
        t0 = time.perf_counter()
        O.rmac_forward(sd, arch, probe)
        dt = time.perf_counter() - t0
        if dt < best[0]:
            best = (dt, nt)
    torch.set_num_threads(best[1])
    x = synth.synth_images(11, 1, size, size)
    O.rmac_forward(sd, arch, x)   # warm-up (allocator, oneDNN primitive cache)
    n, t0 = 0, time.perf_counter()
    while True:
        O.rmac_forward(sd, arch, x)
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 64:
            break
    return {'value': round(n / el, 4), 'unit': 'images/sec', 'cores': torch.get_num_threads(),
            'cpu_allotted': allotted, 'cpu_visible': visible, 'kind': 'port',
            'sample': '%d x %s fp32 %dx%d forward, batch 1, oracle/dir_oracle.py (%.1f s)' % (n, arch, size, size, el)}


def layer_group(name):
    """'layer3.7.conv2' -> 'layer3.conv2' (the identical blocks 1.. of a stage share a row); block 0 of a
    stage differs in shape (stride / input width) and keeps its own name 'layer3.0.conv2'."""
    parts = name.split('.')
    if len(parts) == 3 and parts[0].startswith('layer') and parts[1] != '0':
        return parts[0] + '.' + parts[2]
    return name


def kernel_table(prof, nprof, peak_tf, traffic):
    """One row per (kernel, layer group, algorithmic flops, bytes): launches per step, average duration,
    share of the step, the roof that bounds it (arithmetic intensity vs 2.5 PF / 8 TB/s = 312 FLOP/B), the
    fraction of that roof and - where profiles/traffic.json has it - PMC HBM bytes / algorithmic bytes.
    Returned compact ({"cols": [...], "rows": [[...], ...]}, rows below 1 % of the step folded into one) so
    that the one JSON line stays a few KB; `--layers` prints every launch."""
    groups = {}
    total = sum(r['ms'] for r in prof)
    for r in prof:
        key = (r['kernel'], layer_group(r['name']), r['flops'], r['bytes'])
        g = groups.setdefault(key, [0.0, 0])
        g[0] += r['ms']
        g[1] += 1
    rows, rest = [], [0.0, 0]
    balance = peak_tf * 1e12 / (PEAK_HBM_GBS * 1e9)
    for (kern, grp, fl, by), (ms, n) in sorted(groups.items(), key=lambda kv: -kv[1][0]):
        if ms / total < 0.01:
            rest[0] += ms
            rest[1] += n
            continue
        avg = ms / n
        tf = fl / (avg * 1e-3) / 1e12 if fl else 0.0
        gbs = by / (avg * 1e-3) / 1e9 if by else 0.0
        hbm = (fl / by) < balance if (fl and by) else True
        t = (traffic or {}).get(kern)
        if isinstance(t, dict):          # per-shape PMC traffic keyed by layer group (scripts/summarize_prof.py)
            t = t.get(grp)
        rows.append([kern.replace('conv_igemm<', '').rstrip('>') if kern.startswith('conv_igemm<') else kern, grp,
                     round(n / nprof, 1), round(avg, 4), round(ms / total, 3), 'hbm' if hbm else 'mfma',
                     round(gbs / PEAK_HBM_GBS if hbm else tf / peak_tf, 3), round(t / by, 2) if (t and by) else None])
    if rest[1]:
        rows.append(['(others)', '', round(rest[1] / nprof, 1), round(rest[0] / rest[1], 4), round(rest[0] / total, 3),
                     None, None, None])
    return {'cols': ['kernel', 'layers', 'launches_per_step', 'avg_ms', 'share', 'bound', 'frac_of_bound',
                     'pmc_traffic_ratio'], 'rows': rows}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=32, help='images per GPU per step')
    ap.add_argument('--arch', default='resnet101')
    ap.add_argument('--size', type=int, default=1024)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp16'])
    ap.add_argument('--autotune', action='store_true',
                    help='time every admissible tile variant per layer first (default: the built-in tile heuristic, '
                         'which the tuner no longer beats at this shape)')
    ap.add_argument('--no-autotune', action='store_true', help='(default; kept for old command lines)')
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='0 disables the CPU baseline leg')
    ap.add_argument('--profile-every', type=int, default=4,
                    help='record per-launch HIP events on every n-th timed step (1 = all steps)')
    ap.add_argument('--layers', action='store_true', help='also print the per-layer profile to stderr')
    ap.add_argument('--dump-launches', default='',
                    help='write the launch sequence of one forward (name, kernel, flops, bytes, avg ms) as JSON: '
                         'scripts/summarize_prof.py aligns rocprofv3 dispatches with it')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit('bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)' % args.gpus)
        args.gpus = world
    assert torch.cuda.is_available(), 'bench.py needs an MI355X'
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or 'RANK' in os.environ:      # under torch.distributed.run the RCCL path runs even at N = 1
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))   # RCCL

    import synth
    from dirtorch_amd import nets
    net = nets.create_model(args.arch + '_rmac', pretrained='')
    net.load_state_dict(synth.synth_state_dict(args.arch, seed=7))
    net.compute_dtype = args.dtype
    net.cuda().eval()

    B, S, K, Wm = args.batch, args.size, args.steps, args.warmup
    g = torch.Generator(device='cuda').manual_seed(1234 + rank)
    x = torch.randn(B, 3, S, S, generator=g, device='cuda')       # normalised-image statistics
    if os.environ.get('DIRTORCH_AMD_BENCH_CONST_INPUT'):            # experiment only (profiles/README.md): constant image ->
        x.zero_()                                                   # spatially constant activations, minimal operand toggling
    D = net.out_dim
    shard = torch.empty(K * B, D, device='cuda')                  # this rank's descriptor block

    net.autotune = args.autotune and not args.no_autotune
    net(x)                                                        # build engine, (autotune), first touch
    net.autotune = False
    for _ in range(Wm):
        net(x)
    torch.cuda.synchronize()

    net.set_profiling(256 * (K + 1))    # event pairs pre-created: nothing is allocated while timing
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        net.pause_profiling(k % args.profile_every != 0)   # event records only on sampled steps
        shard[k * B:(k + 1) * B] = net(x)
    if dist is not None:
        allb = torch.empty(world * K * B, D, device='cuda')
        dist.all_gather_into_tensor(allb, shard)                  # one exchange step (RCCL)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    prof = net.get_profile()
    net.set_profiling(False)

    if rank == 0:
        nprof = len([k for k in range(K) if k % args.profile_every == 0])
        fam = {}
        for r in prof:
            f = fam.setdefault(r['kernel'], [0.0, 0.0, 0.0, 0])
            f[0] += r['ms']
            f[1] += r['flops']
            f[2] += r['bytes']
            f[3] += 1
        conv_ms = sum(v[0] for k, v in fam.items() if k.startswith('conv_igemm'))
        conv_fl = sum(v[1] for k, v in fam.items() if k.startswith('conv_igemm'))
        dom = max((k for k in fam if k.startswith('conv_igemm')), key=lambda k: fam[k][0])
        dms, dfl, dby, dn = fam[dom]
        peak_tf = PEAK_TFLOPS[args.dtype]
        tflops = dfl / (dms * 1e-3) / 1e12
        gbs = dby / (dms * 1e-3) / 1e9
        # which roof bounds the dominant kernel: its arithmetic intensity vs the machine balance
        # (2.5 PF dense / 8 TB/s = 312 FLOP/B; MI355X_MICROARCH.md)
        hbm_bound = (dfl / dby) < (peak_tf * 1e12) / (PEAK_HBM_GBS * 1e9)
        traffic_all = None
        tfile = os.path.join(ROOT, 'profiles', 'traffic.json')   # PMC-derived HBM bytes / launch (scripts/summarize_prof.py)
        if os.path.isfile(tfile):
            try:
                traffic_all = json.load(open(tfile))
            except Exception:
                traffic_all = None
        traffic = (traffic_all or {}).get(dom)
        if isinstance(traffic, dict):        # per-shape entries: average over the kernel's launch mix of this step
            per = {}
            for r in prof:
                if r['kernel'] == dom:
                    per[layer_group(r['name'])] = per.get(layer_group(r['name']), 0) + 1
            tot = sum(per.values())
            traffic = sum(traffic.get(g, 0.0) * n for g, n in per.items()) / tot if tot and all(g in traffic for g in per) else None
        all_ms = sum(v[0] for v in fam.values())
        all_fl = sum(v[1] for v in fam.values())
        roof = {'bound': 'hbm' if hbm_bound else 'mfma', 'kernel': dom,
                'achieved': round(gbs if hbm_bound else tflops, 2),
                'peak': PEAK_HBM_GBS if hbm_bound else peak_tf,
                'unit': 'GB/s' if hbm_bound else 'TFLOP/s',
                'frac': round(gbs / PEAK_HBM_GBS if hbm_bound else tflops / peak_tf, 4),
                'traffic': traffic,
                'note': 'frac = the largest-time-share kernel against ITS roof; whole step = step_mfma_frac of the MFMA peak',
                # measured once on a pool box (profiles/r02_mfma_ceiling.txt): bare register-only MFMA loop, 8 waves/CU
                'mfma_ceiling_measured_tflops': {'random_operands': 1582, 'zero_operands': 2285},
                'hbm_ceiling_measured_gbs': {'read': 6305, 'copy_1to1': 5147},   # profiles/r02_hbm_ceiling.txt
                'launches': dn, 'avg_launch_ms': round(dms / dn, 5),
                'flops_per_launch': dfl / dn, 'algorithmic_bytes_per_launch': dby / dn,
                'kernel_tflops': round(tflops, 2), 'kernel_gbs': round(gbs, 1),
                'all_conv_tflops': round(conv_fl / (conv_ms * 1e-3) / 1e12, 2),
                'all_conv_mfma_frac': round(conv_fl / (conv_ms * 1e-3) / 1e12 / peak_tf, 4),
                'step_mfma_frac': round(all_fl / (all_ms * 1e-3) / 1e12 / peak_tf, 4),
                'profiled_steps': nprof,
                'all_conv_ms_per_step': round(conv_ms / nprof, 4),
                'all_kernels_ms_per_step': round(all_ms / nprof, 4),
                'kernels': kernel_table(prof, nprof, peak_tf, traffic_all)}
        if args.dump_launches:
            first = [r for r in prof]
            L = next((i for i in range(1, len(first)) if first[i]['name'] == first[0]['name']), len(first))
            seq = []
            for i in range(L):
                same = [prof[j] for j in range(i, len(prof), L)]
                seq.append({'name': prof[i]['name'], 'kernel': prof[i]['kernel'], 'flops': prof[i]['flops'],
                            'bytes': prof[i]['bytes'], 'ms': sum(r['ms'] for r in same) / len(same)})
            json.dump(seq, open(args.dump_launches, 'w'), indent=0)
        if args.layers:
            agg = {}
            for r in prof:
                a = agg.setdefault((r['name'], r['kernel']), [0.0, r['flops'], r['bytes'], 0])
                a[0] += r['ms']
                a[3] += 1
            for (name, kern), (ms, fl, by, n) in agg.items():
                ms /= n
                print('%-24s %-36s %8.3f ms %8.1f TF/s %8.1f GB/s' % (
                    name, kern, ms, fl / ms / 1e9, by / ms / 1e6), file=sys.stderr)
        cpu = None
        if world == 1 and args.cpu_seconds > 0:
            cpu = cpu_baseline(args.arch, S, args.cpu_seconds)
        value = world * B * K / el
        out = {
            'metric': 'images/sec descriptor extraction (%s-GeM, %dx%d)' % (args.arch, S, S),
            'value': round(value, 2), 'unit': 'images/sec', 'n_gpus': world, 'steps': K, 'warmup': Wm,
            'ms_per_step': round(el / K * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': 'configs[1]: %s_rmac (AP-GeM head) single-scale %dx%d descriptor extraction, '
                                   '1 process per GPU, synthetic weights + images' % (args.arch, S, S),
                       'batch_per_gpu': B, 'global_batch': world * B, 'input': 'fp32 NCHW resident in HBM',
                       'gflop_per_image': GFLOP_PER_IMG.get((args.arch, S)),
                       'tflops_per_gpu': round(value / world * GFLOP_PER_IMG.get((args.arch, S), 0) / 1e3, 1),
                       'parallelism': 'image-parallel shards, 1 all-gather of descriptors' if world > 1 else 'single GPU'},
            'roofline': roof, 'cpu_baseline': cpu,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
