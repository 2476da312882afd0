#!/usr/bin/env python
"""bench.py - images/sec of descriptor extraction (ResNet101-GeM, 1024x1024) on N MI355X.

    python bench.py                                   (= --gpus 1 --steps 100 --warmup 3; the driver passes its own K / W)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (dir_forward: prep -> trunk -> GeM -> FC -> L2) over one batch
of synthetic normalised images that is already resident in HBM.  The default dtype is fp16p (fp16 with the paired head):
the fast format that meets the north-star 1e-4 cosine tolerance on a conditioned network - bf16, which BASELINE configs[1]
names, cannot; `config.one_minus_cos` / `tolerance` / `meets_tolerance` / `bf16_images_per_sec` /
`bf16_one_minus_cos` are flat scalars of the line, measured in the same run.  `python bench.py --gpus N` without
torch.distributed.run launches its own N ranks (self_spawn; --dry-launch = that plumbing on the CPU over gloo).  Database images are sharded
image-parallel over the ranks (no data-path collective inside a step); after the K steps each rank
all-gathers its shard's descriptor block once over RCCL/xGMI (the one exchange step the path has,
inside the timed region for N > 1; the same collective has run once before the clock starts).  Weights are the deterministic synthetic checkpoint of
tests/synth.py (there is no network for real ones); oracle/ is imported by the cpu_baseline leg only.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      the kernel with the largest time share (bound / achieved / peak / frac / traffic) AND
                `kernels`: the same figures for every (kernel, layer shape) group of the step, plus the
                step-level MFMA fraction - algorithmic FLOPs and bytes / event-timed launch durations
                collected inside this process over the timed steps
  cpu_baseline  the CPU oracle (a port of the reference forward) timed on this box's host cores
                on a bounded sample of the same workload (rank 0, N == 1 only)
and, inside `config`, measured outside the timed region (N = 1 only):
  precision     the rates of the same step in the other formats (bf16 / fp16 / strict fp32: 20 steps each, median + min / max
                over groups of 5) and every dtype's distance from the CPU oracle (descriptors on the calibrated checkpoint, mAP)
  workloads     BASELINE configs[3] and configs[4] on this GPU, 20 steps each + min / median / max over 4 more groups of 5
                (the code of the two workloads below)
The timed loops run with the Python cycle collector off (a full collection in the first timed step idles the GPU for 35-55 ms).

Two more workloads keep the same flags and JSON contract (the default above is BASELINE configs[1]):
  --workload distractors   configs[3]: 70 x 1 006 322 x 2048 similarity + device rank / AP, the database sharded over
                           the ranks, ONE all-gather of descriptor (or --exchange scores: score) blocks; the descriptors
                           are L2-normalised, so the similarity runs on two fp16 planes (dir_similarity_unit; the range is
                           established once per database, outside the timed steps; --sim-general = the six-product bf16 kernel)
  --workload multiscale    configs[4]: three scales of resident uint8 1200^2 images, fp16, device-side resize
"""
import argparse
import gc
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

PEAK_TFLOPS = {'bf16': 2500.0, 'fp16': 2500.0, 'fp16p': 2500.0}   # dense MFMA, MI355X_MICROARCH.md (no sparsity)
PEAK_HBM_GBS = 8000.0                            # HBM3E spec (6.29 TB/s measured copy)
GFLOP_PER_IMG = {('resnet101', 1024): 325.99, ('resnet50', 224): 8.183}


IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)     # dirtorch/nets/backbones/resnet.py:110-111


def to_uint8_nhwc(x, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """Normalised fp32 NCHW pictures (tests/synth.py) -> the raw uint8 NHWC images they would have been decoded from
    (PIL -> np.uint8 HWC, dirtorch/utils/transforms.py:617-623 run backwards, rounded and clamped)."""
    m, s_ = torch.tensor(mean).view(1, 3, 1, 1), torch.tensor(std).view(1, 3, 1, 1)
    return ((x * s_ + m) * 255.0).round().clamp_(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()


def normalise_uint8(u8, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """What the REFERENCE feeds its network for a uint8 image: ToTensor (/ 255) then Normalize, fp32 NCHW."""
    m, s_ = torch.tensor(mean).view(1, 3, 1, 1), torch.tensor(std).view(1, 3, 1, 1)
    return ((u8.permute(0, 3, 1, 2).float() / 255.0) - m) / s_


def bench_state_dict(arch, size, kind, dist=None, rank=0):
    """The checkpoint the timed step runs: 'calibrated' = tests/synth.py's He-init weights with BatchNorm statistics calibrated
    on two synthetic images of the bench size (one fp32 CPU forward; the conditioned network the parity numbers are quoted on),
    'he' = the plain He-init checkpoint of rounds 1-5.  Under torch.distributed rank 0 calibrates and the others load its file."""
    import synth
    if kind == 'he':
        return synth.synth_state_dict(arch, seed=7)
    path = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'dir_bench_calibrated_%s_%d_%d.pt' % (arch, size, os.getuid()))
    if dist is None or rank == 0:
        sd = synth.calibrated_state_dict(arch, synth.synth_images(99, 2, size, size), seed=7)
        if dist is not None:
            torch.save(sd, path)
    if dist is not None:
        dist.barrier()
        if rank != 0:
            sd = torch.load(path, weights_only=True)
    return sd


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


def reference_cli_scalars(arch, size):
    """The REFERENCE's own CLI (`python -m dirtorch.extract_features --gpu -1`, dirtorch/extract_features.py:82-124) cannot
    run on the GPU box (no /root/reference there); oracle/time_reference_cli.py timed it in the build container and
    oracle/reference_cli_timing.json holds the result.  Flat scalars for the bench line, labelled with where they are from."""
    try:
        t = json.load(open(os.path.join(ROOT, 'oracle', 'reference_cli_timing.json')))
        for k, v in t.items():
            if isinstance(v, dict) and v.get('arch') == arch and v.get('size') == size:
                return {'reference_cli_img_s': v['cli_images_per_s'], 'reference_cli_forward_only_img_s': v['forward_only_images_per_s'],
                        'reference_cli_cores': v['cpus_visible'], 'reference_cli_host': 'build container (%s), not this box; %d images, batch 1' % (
                            t.get('host', '?'), v['images'])}
    except (OSError, ValueError, KeyError):
        pass
    return {}


def cpu_baseline(arch, size, budget_s, sd=None, images=None):
    """CPU oracle forward, batch 1 (the reference's default path, test_dir.py:52-55), all cores.
    sd / images: time the forwards on THESE inputs and hand their descriptors back (second return value) -
    the parity leg needs the oracle's output for the same images anyway, and the time of an fp32 forward
    does not depend on the weight values."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import dir_oracle as O          # the only oracle import of this file: the timed CPU port / the parity checker
    import synth
    if sd is None:
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
    x = synth.synth_images(11, 1, size, size) if images is None else images
    O.rmac_forward(sd, arch, x[:1])   # warm-up (allocator, oneDNN primitive cache)
    outs = []
    n, t0 = 0, time.perf_counter()
    while True:
        d = O.rmac_forward(sd, arch, x[n % len(x):n % len(x) + 1])
        if n < len(x):
            outs.append(d.reshape(1, -1))
        n += 1
        el = time.perf_counter() - t0
        if (el >= budget_s and n >= len(x)) or n >= 64:
            break
    out = {'value': round(n / el, 4), 'unit': 'images/sec', 'cores': torch.get_num_threads(),
           'cpu_allotted': allotted, 'cpu_visible': visible, 'kind': 'port',
           'sample': '%d x %s fp32 %dx%d forward, batch 1, oracle/dir_oracle.py (%.1f s)' % (n, arch, size, size, el)}
    out.update(reference_cli_scalars(arch, size))
    return out, torch.cat(outs).numpy()


def step_spread(step, groups=4, per_group=5):
    """ms per step of `step()` over `groups` separately timed groups of `per_group` steps (after the contract's K timed
    steps, outside them): {min, median, max} so that a side number carries its spread."""
    ms = []
    for _ in range(groups):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(per_group):
            step()
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t0) / per_group * 1e3)
    ms.sort()
    return {'min': round(ms[0], 4), 'median': round(ms[len(ms) // 2], 4), 'max': round(ms[-1], 4), 'groups': groups,
            'steps_per_group': per_group}


def grouped_rate(step, items_per_step, steps=20, group=5, warm=2):
    """items/s of `step()` as (median, min, max) over steps // group timed groups of `group` back-to-back steps
    (one synchronize per group: the spread of a side number without a host round trip per step)."""
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    rates = []
    for _ in range(max(1, steps // group)):
        t0 = time.perf_counter()
        for _ in range(group):
            step()
        torch.cuda.synchronize()
        rates.append(items_per_step * group / (time.perf_counter() - t0))
    rates.sort()
    return round(rates[len(rates) // 2], 1), round(rates[0], 1), round(rates[-1], 1)


def precision_leg(arch, size, batch, x_bench, cpu_seconds, headline_dtype='fp16p', feed='f32', sd_cal=None, sd_rate=None):
    """Outside the timed region, rank 0 at N = 1: what the three storage formats cost and what they lose.

      images_per_sec   the same step as the headline in the other storage formats - bf16 (what BASELINE configs[1] names),
                       fp16, fp16p (fp16 with the paired head of conv_pair.hip: the fast mode that meets the 1e-4 bar on
                       a conditioned network; the headline's own format is skipped) and the strict fp32 mode
                       (conv_f32.hip): 20 steps each, the median / min / max rate over groups of 5 steps
      one_minus_cos    engine vs the fp32 CPU oracle (oracle/dir_oracle.py) on the BatchNorm-calibrated synthetic
                       checkpoint - the conditioned network on which 16-bit storage is visible - for two images at
                       the bench size travelling INSIDE a batch of `batch` (so the timed kernel mix computes them)
      map              |mAP(engine) - mAP(oracle)| through extraction -> PCA whitening -> similarity -> revisitop AP
                       (the ROxford protocol: easy / hard / junk lists) on a small synthetic retrieval set

    Returns (dict for config['precision'], the cpu_baseline dict) - the CPU forwards that produce the parity
    reference are the timed cpu_baseline sample."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import dir_oracle as O
    import synth
    from dirtorch_amd import nets
    from dirtorch_amd.utils import common

    def engine(sd, dtype, arch=arch):
        net = nets.create_model(arch + '_rmac', pretrained='')
        net.load_state_dict(sd)
        net.compute_dtype = dtype
        return net.cuda().eval()

    out = {'images_per_sec': {}, 'images_per_sec_spread': {}, 'one_minus_cos': {}, 'map': {}}
    # ---- throughput of the OTHER formats on the headline workload (>= 20 steps each, rate per group of 5) ----------
    sd0 = sd_rate if sd_rate is not None else synth.synth_state_dict(arch, seed=7)
    others = [(d, batch, 20) for d in ('bf16', 'fp16', 'fp16p') if d != headline_dtype] + [('f32', max(1, batch // 4), 20)]
    for dtype, b, steps in others:
        net = engine(sd0, dtype)
        xb = x_bench[:b]
        med, lo, hi = grouped_rate(lambda: net(xb), b, steps)
        out['images_per_sec'][dtype] = med
        out['images_per_sec'][dtype + '_batch'] = b
        out['images_per_sec_spread'][dtype] = {'min': lo, 'median': med, 'max': hi, 'steps': steps, 'group': 5}
        del net
        torch.cuda.empty_cache()
    # ---- descriptors vs the CPU oracle at the bench size, calibrated checkpoint --------------------------
    sd = sd_cal if sd_cal is not None else synth.calibrated_state_dict(arch, synth.synth_images(99, 2, size, size), seed=7)
    xp = synth.synth_images(4, 2, size, size)
    if feed == 'u8':      # the engine gets the raw uint8 pictures, the oracle what the reference makes of them (ToTensor + Normalize)
        xp8 = to_uint8_nhwc(xp)
        xp = normalise_uint8(xp8)
    cpu, ref = cpu_baseline(arch, size, cpu_seconds, sd=sd, images=xp)
    xin = x_bench.clone()
    xin[:2] = (xp8 if feed == 'u8' else xp).cuda()
    for dtype in ('bf16', 'fp16', 'fp16p', 'f32'):
        net = engine(sd, dtype)
        b = batch if dtype != 'f32' else max(2, batch // 4)
        got = net(xin[:b])[:2].cpu().numpy()
        out['one_minus_cos'][dtype] = float('%.3g' % (1 - O.cosine(got, ref)).max())
        del net
        torch.cuda.empty_cache()
    out['one_minus_cos']['checkpoint'] = 'BatchNorm-calibrated synthetic (tests/synth.py), 2 images inside the batch, %s feed' % (
        'uint8 NHWC' if feed == 'u8' else 'fp32 NCHW')
    # ---- mAP through the whole post-extraction path ----------------------------------------------------------
    # (ResNet-18: with random weights the deeper trunks are chaotic - noisy copies of an image decorrelate, the
    # oracle's own mAP sits at chance (0.03 for R101, 0.09 for R50) and a dmAP there measures nothing; on R18 the
    # planted structure is retrievable: mAP-easy 1.0, medium 0.68, hard 0.29.  Same kernels, same post-processing.)
    r = np.random.RandomState(11)
    march, N, Q, S = 'resnet18', 160, 12, 224
    imgs = synth.synth_images(21, N, S, S).numpy()
    gnd = []
    perm = r.permutation(np.arange(Q, N))
    for q in range(Q):
        idx = perm[q * 9:(q + 1) * 9]           # disjoint positives / junk per query, 40 pure distractors
        for j, sigma in zip(idx[:6], (0.05, 0.1, 0.15, 0.3, 0.45, 0.6)):
            imgs[j] = imgs[q] + sigma * r.standard_normal(imgs[q].shape).astype(np.float32)
        gnd.append({'easy': sorted(idx[:3].tolist()), 'hard': sorted(idx[3:6].tolist()), 'junk': sorted([q] + idx[6:].tolist())})
    xs = torch.from_numpy(imgs)
    sdm = synth.calibrated_state_dict(march, synth.synth_images(99, 8, S, S), seed=7)
    refd = torch.cat([O.rmac_forward(sdm, march, xs[i:i + 40]).reshape(-1, 2048) for i in range(0, N, 40)]).numpy()
    P = O.fit_pca(refd[Q:])
    kw = dict(whitenp=0.25, whitenv=32)
    ref_w = O.whiten_features(refd, P, **kw)
    m_ref = O.mean_ap(O.matmul(ref_w[:Q], ref_w), gnd)
    out['map']['set'] = ('%s, %d synthetic %dx%d images, %d queries with planted near-duplicates, revisitop easy/hard/junk '
                         'protocol, PCA whitening to 32-d (calibrated checkpoint)' % (march, N, S, S, Q))
    out['map']['oracle'] = {k: round(v, 5) for k, v in m_ref.items()}
    for dtype in ('bf16', 'fp16', 'fp16p', 'f32'):
        net = engine(sdm, dtype, march)
        got = torch.cat([net(xs[i:i + 40].cuda()) for i in range(0, N, 40)]).cpu().numpy()
        got_w = common.whiten_features(got, P, **kw)
        m = O.mean_ap(common.matmul(got_w[:Q], got_w), gnd)
        out['map']['d_map_' + dtype] = float('%.3g' % max(abs(m[k] - m_ref[k]) for k in m_ref))
        del net
        torch.cuda.empty_cache()
    return out, cpu


def other_workloads(args, x_bench):
    """BASELINE configs[3] and configs[4] on one GPU, bounded to a few seconds, for the driver's one line: the
    distractor protocol (70 x 1 006 322 x 2048 similarity + device rank / AP) and the three-scale extraction of
    1200^2 images.  Same code as --workload distractors / multiscale, fewer steps; failures are reported, not raised
    (the headline number is already measured)."""
    import copy
    out = {}
    del x_bench
    torch.cuda.empty_cache()
    for name, fn, over in (('distractors', bench_distractors, {'steps': 20, 'warmup': 2, 'cpu_seconds': 0.0}),
                           ('multiscale', bench_multiscale, {'steps': 20, 'warmup': 2})):      # (keeps --cpu-seconds: its parity leg)
        a = copy.copy(args)
        for k, v in over.items():
            setattr(a, k, v)
        try:
            r = fn(a, 1, 0, None)
            if name == 'distractors':
                out[name] = {'db_rows': a.db_rows, 'queries': a.queries, 'steps': a.steps, 'ms_per_step': r['ms_per_step'],
                             'ms_per_step_spread': r['ms_per_step_spread'], 'sim_ms_min_median_max': r['roofline']['launch_ms_min_median_max'],
                             'db_rows_per_sec': r['value'], 'sim_ms': r['roofline']['avg_launch_ms'],
                             'sim_hbm_frac': r['roofline']['frac'], 'sim_kernel': r['roofline']['kernel'], 'rank_ap_ms': r['roofline']['rank_ap_ms'],
                             'mAP_medium': r['config']['mAP_medium'], 'whiten_ms': r['roofline'].get('whiten_ms'),
                             'whiten': r['roofline'].get('whiten')}
            else:
                out[name] = {'images_per_sec_3scale': r['value'], 'steps': a.steps, 'ms_per_step': r['ms_per_step'],
                             'ms_per_step_spread': r['ms_per_step_spread'], 'batch': a.ms_batch,
                             'size': a.ms_size, 'step_mfma_frac': r['roofline']['frac'], 'dtype': r['dtype'],
                             **{k: v_ for k, v_ in r['config'].items() if k.startswith(('one_minus_cos', 'tolerance', 'meets_tolerance',
                                                                                        'parity_sample', 'fp16_images_per_sec', 'scale_streams'))}}
        except Exception as e:      # noqa: BLE001 - report and go on
            out[name] = {'error': '%s: %s' % (type(e).__name__, e)}
        torch.cuda.empty_cache()
    return out


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


def exchange_costs(N, Q, D, W):
    """The one exchange step of configs[3] in its two layouts, for W ranks: descriptor blocks [ceil(N/W), D] fp32 (what
    north_star names) or score blocks [Q, ceil(N/W)] fp32 (every rank scores its own rows first)."""
    rows = -(-N // W)
    out = {'world': W}
    for name, b in (('descriptors', (W - 1) * rows * D * 4), ('scores', (W - 1) * Q * rows * 4)):
        out[name] = {'bytes_received_per_rank': b, 'ring_ms_at_one_link': round(b / 153e9 * 1e3, 3),
                     'mesh_ms_at_w_minus_1_links': round(b / (153e9 * (W - 1)) * 1e3, 3)}
    return out


def bench_distractors(args, world, rank, dist):
    """BASELINE configs[3]: RParis6K + 1M distractors.  Rank r owns the contiguous row range
    shard_range(N, r, W) of the [N, 2048] fp32 database (generated on the device: unit-norm rows), the Q query
    descriptors are replicated.  A step is the post-extraction path of dirtorch/test_dir.py:145-167 over the whole
    database:

      --exchange descriptors  (north_star) ONE all_gather_into_tensor of the padded [ceil(N/W), 2048] blocks
                              (8.24 GB at N = 1 006 322), then on every rank: Q x N similarity (sim_split.hip) ->
                              device rank counts + revisitop AP (ranking.hip)
      --exchange scores       every rank scores only ITS rows (Q x N/W), ONE all-gather of the [Q, ceil(N/W)]
                              score blocks (N*Q*4 B = 0.28 GB), then rank + AP: the same APs bit for bit with
                              29x fewer bytes on xGMI and 1/W of the similarity per GPU

    value = database rows ranked per second, whole job (N * steps / time); scaling 'strong' (N is fixed)."""
    from dirtorch_amd import distributed as ddist
    from dirtorch_amd import ops, ranking
    N, Q, D, K, Wm = args.db_rows, args.queries, 2048, args.steps, args.warmup
    lo, hi = ddist.shard_range(N, rank, world)
    rows = max(h - l for l, h in (ddist.shard_range(N, r, world) for r in range(world)))
    g = torch.Generator(device='cuda').manual_seed(77 + rank)
    local = torch.zeros(rows, D, device='cuda')                      # padded to the common block size
    for i in range(0, hi - lo, 65536):                               # (chunked: no second 8 GB temporary)
        n = min(65536, hi - lo - i)
        local[i:i + n] = torch.nn.functional.normalize(torch.randn(n, D, generator=g, device='cuda'), dim=1)
    gq = torch.Generator(device='cuda').manual_seed(5)
    qs = torch.nn.functional.normalize(torch.randn(Q, D, generator=gq, device='cuda'), dim=1)

    class _DB(object):   # revisitop-format relevance lists (easy / hard / junk), same on every rank
        relevants = None
    r = np.random.RandomState(1)
    db = _DB()
    db.nimg, db.nquery, db.easy, db.hard, db.junk = N, Q, [], [], []
    for q in range(Q):
        idx = r.choice(N, 200, replace=False)
        db.easy.append(sorted(idx[:80].tolist()))
        db.hard.append(sorted(idx[80:160].tolist()))
        db.junk.append(sorted(idx[160:].tolist()))
    tables = ranking.build_probe_tables(db)
    # L2-normalised descriptors: the fp16-pair form of the large-database similarity (csrc/sim_split.hip PAIR) applies; the
    # range is established ONCE per database, outside the timed steps, the way a retrieval service would (ranking.is_unit_range)
    unit = not args.sim_general and ranking.is_unit_range(qs, local)
    sim = lambda q, b: ops.similarity(q, b, unit_range=unit)   # noqa: E731
    xch = dist is not None      # a process group exists (also a ONE-rank one, under torch.distributed.run): the exchange step runs through RCCL
    full = torch.empty(world * rows, D, device='cuda') if args.exchange == 'descriptors' and xch else None
    sc_all = torch.empty(world, Q, rows, device='cuda') if args.exchange == 'scores' and xch else None
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    t_x, t_s, t_r = [], [], []

    def step(record):
        if record:
            ev[0].record()
        if args.exchange == 'descriptors':
            if xch:
                ddist.allgather_blocks(full, local)                   # the one exchange step (RCCL over xGMI; DIRTORCH_AMD_EXCHANGE=mesh: direct sends)
            if record:
                ev[1].record()
            # (shards of unequal length - 1 006 322 % 8 = 2 - arrive padded and are scored block by block:
            # dirtorch_amd.distributed.score_gathered, tests/test_ranking_gpu.py::test_sharded_scoring_*)
            scores = ddist.score_gathered(qs, full, N, world, sim) if xch else sim(qs, local[:N])
        else:
            mine = sim(qs, local)                                     # [Q, rows] (padding rows score 0)
            if record:
                ev[1].record()
            if xch:
                ddist.allgather_blocks(sc_all.view(world * Q, rows), mine)
                scores = ddist.merge_score_blocks(sc_all, N, world)
            else:
                scores = mine[:, :N].contiguous()
        if record:
            ev[2].record()
        aps = ranking.eval_aps_device(db, scores, tables)
        if record:
            ev[3].record()
            torch.cuda.synchronize()
            a, b, c = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), ev[2].elapsed_time(ev[3])
            # descriptors: exchange, then similarity; scores: similarity, then exchange
            (t_x if args.exchange == 'descriptors' else t_s).append(a)
            (t_s if args.exchange == 'descriptors' else t_x).append(b)
            t_r.append(c)
        return aps

    # ---- a8 at this scale, outside the timed steps: PCA-whitening the rank's database rows before scoring (test_dir.py:136-138,
    # common.py:221-239) - synthetic PCA (mean of the rows, orthonormal components, variances over four decades, whitenp 0.25) ----
    whiten = None
    if rank == 0 and not getattr(args, 'no_whiten', False):
        try:
            nloc = hi - lo
            mean = local[:8192].mean(dim=0).contiguous()
            comps = torch.linalg.qr(torch.randn(D, D, device='cuda', generator=gq, dtype=torch.float32))[0].t().contiguous()
            alpha = (1.0 / torch.logspace(-1, -5, D, device='cuda', dtype=torch.float64).pow(0.25)).float()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

            def timed(fn, reps):
                fn()
                ms = []
                for _ in range(reps):
                    e0.record()
                    out_ = fn()
                    e1.record()
                    torch.cuda.synchronize()
                    ms.append(e0.elapsed_time(e1))
                    del out_
                return sorted(ms)
            w_ms = timed(lambda: ops.pca_whiten(local[:nloc], comps, mean, alpha, l2norm=True, unit_range=True), 5)
            nx = min(nloc, 131072)
            x_ms = timed(lambda: ops.pca_whiten(local[:nx], comps, mean, alpha, l2norm=True, unit_range=False), 3)
            fl = 2.0 * nloc * D * D
            whiten = {'rows': nloc, 'width': D, 'ms': round(w_ms[len(w_ms) // 2], 3), 'ms_min_max': [round(w_ms[0], 3), round(w_ms[-1], 3)],
                      'kernel': 'whiten_split_kernel (two fp16 planes per operand, 3 plane products) + l2norm_rows_kernel',
                      'algorithmic_tflops': round(fl / (w_ms[len(w_ms) // 2] * 1e-3) / 1e12, 1),
                      'frac_of_fp32_mfma_peak_157': round(fl / (w_ms[len(w_ms) // 2] * 1e-3) / 1e12 / 157.0, 3),
                      'issued_frac_of_fp16_peak': round(3.0 * fl / (w_ms[len(w_ms) // 2] * 1e-3) / 1e12 / PEAK_TFLOPS['fp16'], 3),
                      'exact_fp32_chain_ms_scaled': round(x_ms[len(x_ms) // 2] * nloc / nx, 2),
                      'exact_fp32_chain_sample': '%d rows: %.3f ms (gemm_nt_f32, fp32 MFMA) scaled to %d rows' % (nx, x_ms[len(x_ms) // 2], nloc)}
            del comps, mean, alpha
        except Exception as e:      # noqa: BLE001 - report and go on: the timed steps below do not depend on it
            whiten = {'error': '%s: %s' % (type(e).__name__, e)}
        torch.cuda.empty_cache()

    for _ in range(max(Wm, 1)):
        aps = step(False)
    torch.cuda.synchronize()
    gc.collect()
    gc.disable()          # (no cycle collection inside the timed region: see the extract workload)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        aps = step(False)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    el = time.perf_counter() - t0
    gc.enable()
    if dist is not None:
        t = torch.tensor([el], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    spread = step_spread(lambda: step(False)) if world == 1 else None
    for _ in range(20):     # per-phase durations (events on torch's current stream, where every kernel above runs)
        step(True)
    if rank != 0:
        return None
    mean = lambda v: sum(v) / len(v) if v else 0.0   # noqa: E731
    med = lambda v: sorted(v)[len(v) // 2] if v else 0.0   # noqa: E731
    sim_ms, x_ms, r_ms = mean(t_s), mean(t_x), mean(t_r)
    sim_rows = N if args.exchange == 'descriptors' else rows
    sim_bytes = sim_rows * D * 4 + Q * sim_rows * 4          # database rows once + the score block
    x_bytes = ((world - 1) * rows * D * 4) if args.exchange == 'descriptors' else ((world - 1) * Q * rows * 4)
    out = {
        'metric': 'database descriptors ranked/sec (Q x N similarity + revisitop AP, %d x %d x %d)' % (Q, N, D),
        'value': round(N * K / el, 1), 'unit': 'db_rows/sec', 'n_gpus': world, 'steps': K, 'warmup': Wm,
        'ms_per_step': round(el / K * 1e3, 3), 'ms_per_step_spread': spread, 'higher_is_better': True, 'scaling': 'strong',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'configs[3]: RParis6K + 1M synthetic distractors (N = %d unit-norm 2048-d rows, Q = %d), '
                               'database sharded over the ranks, one all-gather of %s, similarity + device rank/AP'
                               % (N, Q, 'descriptor blocks' if args.exchange == 'descriptors' else 'score blocks'),
                   'rccl_ranks': dist.get_world_size() if dist is not None else 0,
                   'exchange': args.exchange, 'exchange_algo': ddist.exchange_algo(), 'rows_per_rank': rows, 'mAP_medium': round(float(np.mean([a['medium'] for a in aps])), 6)},
        'roofline': {'bound': 'hbm', 'kernel': ('sim_split_kernel' if os.environ.get('DIRTORCH_AMD_SIM_V1') else
                                                ('sim_split_lc_kernel<pair: two fp16 planes>' if unit else 'sim_split_lc_kernel')) if sim_rows >= 32768 else 'gemm_nt_f32',
                     'achieved': round(sim_bytes / (sim_ms * 1e-3) / 1e9, 1) if sim_ms else None, 'peak': PEAK_HBM_GBS,
                     'unit': 'GB/s', 'frac': round(sim_bytes / (sim_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) if sim_ms else None,
                     'traffic': None, 'avg_launch_ms': round(sim_ms, 4), 'algorithmic_bytes_per_launch': sim_bytes,
                     'launch_ms_min_median_max': [round(min(t_s), 4), round(med(t_s), 4), round(max(t_s), 4)] if t_s else None,
                     'rank_ap_ms': round(r_ms, 4),
                     # a8 at this scale (outside the timed steps): whitening this rank's rows before they are scored
                     'whiten': whiten, 'whiten_ms': (whiten or {}).get('ms'),
                     # both layouts of the one exchange step priced side by side (the timed one is `exchange`): bytes each rank
                     # RECEIVES, and what a ring (one xGMI link, ~153 GB/s) / a direct full-mesh all-gather (W - 1 links) needs;
                     # on one GPU they are priced for the 8-GPU node BASELINE configs[3] names
                     'exchange_costs': exchange_costs(N, Q, D, world if world > 1 else 8),
                     'exchange': {'bound': 'xgmi', 'bytes_received_per_rank': x_bytes, 'ms': round(x_ms, 4),
                                  'achieved': round(x_bytes / (x_ms * 1e-3) / 1e9, 1) if (x_ms and world > 1) else None,
                                  'peak': 7 * 153.0, 'unit': 'GB/s',
                                  'note': '7 xGMI links x ~153 GB/s per GPU; a ring all-gather is bound by ONE link'}},
        'cpu_baseline': None,
    }
    if world == 1 and args.cpu_seconds > 0:
        # the reference's CPU form of the same step (common.matmul -> np.dot, generic.py:207 argsort per query and
        # mode) on a bounded sample: 50k database rows, all queries
        n = min(N, 50000)
        a, b = qs.cpu().numpy(), local[:n].cpu().numpy()
        t0 = time.perf_counter()
        sc = np.dot(a, b.T)
        for q in range(Q):
            for _ in range(3):
                np.argsort(sc[q])[::-1]
        dt = time.perf_counter() - t0
        out['cpu_baseline'] = {'value': round(n / dt, 1), 'unit': 'db_rows/sec', 'cores': torch.get_num_threads(),
                               'kind': 'port', 'sample': 'np.dot(Q x D, D x %d) + 3 argsorts per query (%.2f s)' % (n, dt)}
    return out


def bench_multiscale(args, world, rank, dist):
    """BASELINE configs[4]: ResNet-101 GeM, THREE-scale descriptors (x 0.7071 / 1 / 1.4142, dirtorch/test_dir.py:111-122
    + utils/common.py:41-55) of 1200 x 1200 images, image-parallel over the ranks (replicated weights, each rank its
    own contiguous image range, no data-path collective), ONE all-gather of the descriptor blocks at the end.  A step =
    one batch of uint8 images resident in HBM -> Pillow-identical resize per scale (resize.hip) -> dir_forward per scale
    -> GeM pooling over the scales + L2.  value = 3-scale images/s, whole job; scaling 'weak'.
    Round 6: computed in --dtype (fp16p by default: configs[4] says "fp16 MFMA", and plain fp16 does not meet the 1e-4 cosine
    on a conditioned network at every size - 1.24e-4 at config A; fp16p runs the same fp16 MFMAs with a paired head); the
    plain-fp16 rate rides along, and with --cpu-seconds > 0 one image's 3-scale descriptor is checked against the CPU oracle
    (resize -> ToTensor / Normalize -> forward per scale -> pool -> L2) on the BatchNorm-calibrated checkpoint."""
    import synth
    from dirtorch_amd import nets, ops
    from dirtorch_amd.utils import common, transforms
    B, S, K, Wm = args.ms_batch, args.ms_size, args.steps, args.warmup
    parity = world == 1 and getattr(args, 'cpu_seconds', 0) > 0
    sd = (synth.calibrated_state_dict(args.arch, synth.synth_images(99, 1, S, S), seed=7) if parity
          else synth.synth_state_dict(args.arch, seed=7))

    def engine(dtype):
        n_ = nets.create_model(args.arch + '_rmac', pretrained='')
        n_.load_state_dict(sd)
        n_.compute_dtype = dtype
        return n_.cuda().eval()
    net = engine(args.dtype)
    g = torch.Generator(device='cuda').manual_seed(99 + rank)
    img = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, device='cuda', generator=g)
    pic = None
    if parity:      # one structured picture travels inside the batch: the row the oracle checks
        pic = to_uint8_nhwc(synth.synth_images(4, 1, S, S))
        img[0] = pic[0].cuda()
    scales = [transforms.Scale(0.7071), None, transforms.Scale(1.4142)]
    sizes = [(S, S) if sc is None else sc.target_size((S, S)) for sc in scales]
    shard = torch.empty(K * B, net.out_dim, device='cuda')

    last = {}
    # Round 6: the three forwards go through the host mirror's stream pool (dirtorch_amd/test_dir.py StreamPool - what
    # extract_multiscale_features does with the scales of its images): the small scale's under-filled launches and the large one's
    # tail rounds fill each other's idle CUs.  Same kernels on the same data: bit-identical to one stream (scripts/exp_multiscale_streams.py).
    from dirtorch_amd.test_dir import StreamPool
    spool = StreamPool(max(1, getattr(args, 'ms_streams', 3)))

    def step(n_=None):
        n_ = n_ or net
        per_scale = []
        for size in sizes:
            x = img if size == (S, S) else ops.resize_bilinear_u8(img, size)
            per_scale.append(spool.run(lambda x=x: n_(x), x))
        spool.join()
        last['per_scale'] = per_scale
        return common.l2_normalize(common.pool(per_scale, 'gem', 3))

    for _ in range(max(Wm, 2)):
        step()
    torch.cuda.synchronize()
    gc.collect()
    gc.disable()          # (no cycle collection inside the timed region: see the extract workload)
    allb = None
    if dist is not None:      # buffer + first-call costs of this collective before the clock starts (see the extract workload)
        allb = torch.empty(world * K * B, net.out_dim, device='cuda')
        dist.all_gather_into_tensor(allb, shard)
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        shard[k * B:(k + 1) * B] = step()
    if dist is not None:
        dist.all_gather_into_tensor(allb, shard)                  # the one exchange step (RCCL)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    gc.enable()
    spread = step_spread(step) if world == 1 else None
    if rank != 0:
        return None
    extra = {}
    if world == 1 and args.dtype != 'fp16':      # the plain-fp16 rate of the same step (what rounds 3-5 reported for configs[4])
        n16 = engine('fp16')
        med, lo_, hi_ = grouped_rate(lambda: step(n16), B, steps=10, group=5)
        extra['fp16_images_per_sec_3scale'] = med
        del n16
    if parity:
        sys.path.insert(0, os.path.join(ROOT, 'oracle'))
        import dir_oracle as O
        torch.set_num_threads(cpu_allotted())
        step()                                           # (the headline dtype's per-scale descriptors of this very batch)
        got = common.l2_normalize(common.pool(last['per_scale'], 'gem', 3))[:1].cpu().numpy()
        got_scale = [d[:1].cpu().numpy() for d in last['per_scale']]
        per, per_q = [], []
        t0 = time.perf_counter()
        for (w_, h_) in sizes:
            u = pic[0].numpy() if (w_, h_) == (S, S) else O.resize_bilinear_u8(pic[0].numpy(), w_, h_)
            xo = normalise_uint8(torch.from_numpy(u)[None])
            per.append(O.rmac_forward(sd, args.arch, xo).reshape(1, -1))
            if args.dtype in ('fp16', 'fp16p', 'bf16'):      # the oracle with this format's storage points rounded: an IDEAL implementation
                per_q.append(O.rmac_forward(sd, args.arch, xo, quant=args.dtype).reshape(1, -1))
        ref = torch.nn.functional.normalize(O.pool(per, 'gem', 3), dim=1).numpy()
        e_scale = [float((1 - O.cosine(g_, r_.numpy())).max()) for g_, r_ in zip(got_scale, per)]
        extra['one_minus_cos_per_scale'] = [float('%.3g' % e) for e in e_scale]
        extra['one_minus_cos'] = float('%.3g' % max(e_scale))
        extra['tolerance'] = 1e-4
        extra['meets_tolerance'] = bool(max(e_scale) < 1e-4)
        # ... and AFTER the reference's multi-scale pooling (common.pool 'gem', p = 3: a signed cube root of the mean of cubes, whose
        # derivative is unbounded at 0 - where the scales' cubes cancel, an entry's error is multiplied by (x / z)^2): reported apart,
        # next to what an ideal implementation of the format's storage points gets through the same pooling
        extra['one_minus_cos_pooled'] = float('%.3g' % (1 - O.cosine(got, ref)).max())
        if per_q:
            ideal = torch.nn.functional.normalize(O.pool(per_q, 'gem', 3), dim=1).numpy()
            extra['one_minus_cos_pooled_ideal_' + args.dtype] = float('%.3g' % (1 - O.cosine(ideal, ref)).max())
        extra['parity_sample'] = ('one %dx%d picture inside the timed batch, 3 scales, BatchNorm-calibrated checkpoint, CPU oracle '
                                  '%.1f s on %d threads; one_minus_cos = the worst SCALE\'s descriptor (what the tolerance is stated on), '
                                  'one_minus_cos_pooled = after common.pool(gem 3) + L2' % (S, S, time.perf_counter() - t0, torch.get_num_threads()))
    # ResNet-101 trunk: 448.76 GFLOP at 1200^2, quadratic in the side
    gflop = sum(448.76 * (s_[0] * s_[1]) / (1200.0 * 1200.0) for s_ in sizes)
    ips = world * K * B / el
    return ({
        'metric': 'images/sec 3-scale descriptor extraction (%s-GeM, %dx%d, scales 0.7071/1/1.4142)' % (args.arch, S, S),
        'value': round(ips, 2), 'unit': 'images/sec', 'n_gpus': world, 'steps': K, 'warmup': Wm,
        'ms_per_step': round(el / K * 1e3, 3), 'ms_per_step_spread': spread, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
        'config': {'workload': 'configs[4]: %s-GeM multi-scale (3 scales %s) extraction of %dx%d uint8 images, %s, '
                               'image-parallel, one all-gather of descriptor blocks' % (args.arch, [s_[0] for s_ in sizes], S, S, args.dtype),
                   'batch_per_gpu': B, 'descriptor_dim': int(net.out_dim), 'scale_streams': len(spool.streams) or 1, **extra,
                   'rccl_ranks': dist.get_world_size() if dist is not None else 0},
        'roofline': {'bound': 'mfma', 'kernel': 'whole step (three dir_forward passes + resize + pooling)',
                     'achieved': round(ips / world * gflop / 1e3, 1), 'peak': PEAK_TFLOPS['fp16'], 'unit': 'TFLOP/s',
                     'frac': round(ips / world * gflop / 1e3 / PEAK_TFLOPS['fp16'], 4), 'traffic': None,
                     'note': 'step-level figure: conv FLOPs of the three scales / step time; the per-kernel table is the '
                             'extract workload\'s (same kernels, same layer shapes at 1024^2)'},
        'cpu_baseline': None})


def self_spawn(n, argv, dry=False):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: start N ranks of this very command (fresh
    interpreters: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / a free MASTER_PORT), stream rank 0's stdout
    (the one JSON line) through, return the first non-zero exit code.  Refuses up front when the box has fewer GPUs."""
    import socket
    import subprocess
    if not dry:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            print('bench.py --gpus %d needs %d GPUs, found %d' % (n, n, have), file=sys.stderr)
            return 2
    def launch():
        with socket.socket() as so:
            so.bind(('127.0.0.1', 0))
            port = so.getsockname()[1]
        ps = []
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
            env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # (the pool's driver needs dmabuf IPC; a user's own setting wins)
            ps.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                       stdout=None if r == 0 else subprocess.DEVNULL))
        return ps
    # wait for all ranks; when one fails, the others would sit in a collective forever: stop exactly the processes started here.
    # The port is found by bind(0) + close, so another process can take it before rank 0 binds it: ranks that all die within the
    # first seconds are relaunched on a new port (twice at most); an interrupted launcher takes its ranks with it.
    rc = 0
    for attempt in range(3):
        procs = launch()
        t_start = time.perf_counter()
        rc, live = 0, list(procs)
        try:
            while live:
                time.sleep(0.2)
                for pr in list(live):
                    code = pr.poll()
                    if code is None:
                        continue
                    live.remove(pr)
                    if code != 0 and rc == 0:
                        rc = code
                        for other in live:
                            other.terminate()
        finally:
            for pr in procs:
                if pr.poll() is None:
                    pr.terminate()
            for pr in procs:
                try:
                    pr.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    pr.kill()
        if rc == 0 or time.perf_counter() - t_start > 20:      # (a rendezvous that lost its port fails at once; anything later is real)
            break
    return rc


def dry_launch(world, rank):
    """--dry-launch: the N > 1 plumbing without a GPU - process group over gloo, the one exchange step of the path
    (dirtorch_amd.distributed.allgather_rows over contiguous shards of unequal length), rank 0 prints the line."""
    import torch.distributed as dist
    from dirtorch_amd import distributed as ddist
    if world > 1 or 'RANK' in os.environ:
        dist.init_process_group('gloo')
    n, d = 10 * world + 3, 8                                  # N % W != 0: the padded last shard
    lo, hi = ddist.shard_range(n, rank, world)
    full = torch.arange(n * d, dtype=torch.float32).reshape(n, d)
    got = ddist.allgather_rows(full[lo:hi].clone(), n) if dist.is_initialized() else full
    ok = bool(torch.equal(got, full))
    if rank == 0:
        print(json.dumps({'dry_launch': True, 'n_gpus': world, 'ranks': dist.get_world_size() if dist.is_initialized() else 1,
                          'backend': 'gloo', 'exchange_ok': ok, 'rows': n}))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=32, help='images per GPU per step')
    ap.add_argument('--arch', default='resnet101')
    ap.add_argument('--size', type=int, default=1024)
    ap.add_argument('--dtype', default='fp16p', choices=['bf16', 'fp16', 'fp16p'],
                    help='storage / MFMA format of the headline step.  Default fp16p: fp16 with the paired head, the fast mode that '
                         'meets the north-star 1e-4 cosine on a conditioned network (bf16, which BASELINE configs[1] names, cannot: '
                         'ideal bf16 storage is 7e-4 there - BASELINE.md section 0); the bf16 rate rides along as config.bf16_images_per_sec')
    ap.add_argument('--input', default='u8', choices=['u8', 'f32'],
                    help="what the timed step is fed, resident in HBM: 'u8' (default since round 6) = raw uint8 NHWC images, ToTensor + "
                         "Normalize on the device inside the step - the feed the drop-in CLIs use (utils/pytorch_loader.py) and the "
                         "reference's real input (transforms.py:617-623); 'f32' = the normalised fp32 NCHW tensor the reference's "
                         "net(x) takes (rounds 1-5).  The other feed's rate rides along as config.<feed>_feed_images_per_sec")
    ap.add_argument('--checkpoint', default='he', choices=['calibrated', 'he'],
                    help="weights of the timed step: 'he' (default) = the random-init (He-normal) checkpoint of tests/synth.py, as in rounds "
                         "1-5 and as the bench contract words it; 'calibrated' = the same weights with BatchNorm statistics calibrated on "
                         "synthetic images - the conditioned network the parity numbers (config.one_minus_cos) are quoted on (one fp32 CPU "
                         "forward of two images before the clock starts).  Since round 6 the OTHER checkpoint's rate of the same step rides "
                         "along as config.calibrated_images_per_sec / config.he_init_images_per_sec, with the dominant kernel's launch "
                         "average on its activations (the 3x3 kernels are power-bound and data-dependent: ~4 % between the two)")
    ap.add_argument('--autotune', action='store_true',
                    help='time every admissible tile variant per layer first (default: the built-in tile heuristic, '
                         'which the tuner no longer beats at this shape)')
    ap.add_argument('--no-autotune', action='store_true', help='(default; kept for old command lines)')
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='0 disables the CPU legs (baseline + precision)')
    ap.add_argument('--no-precision', action='store_true',
                    help='skip the fp16 / strict-fp32 throughput and the parity-vs-oracle fields (config.precision)')
    ap.add_argument('--no-workloads', action='store_true',
                    help='skip the bounded configs[3] / configs[4] runs that fill config.workloads')
    ap.add_argument('--workload', default='extract', choices=['extract', 'distractors', 'multiscale'],
                    help="extract = BASELINE configs[1] (default); multiscale = configs[4] (3 scales of 1200^2, fp16); distractors = configs[3]: a database of --db-rows "
                         "2048-d descriptors sharded over the ranks, ONE all-gather, Q x N similarity, device rank + AP")
    ap.add_argument('--ms-batch', type=int, default=16, help='multiscale: images per GPU per step (8 / 16 / 24: 381 / 416 / 400 three-scale img/s)')
    ap.add_argument('--ms-streams', type=int, default=3,
                    help='multiscale: HIP streams the three scales\' forwards are issued on (the host mirror\'s StreamPool, what '
                         'extract_multiscale_features does; 1 = one after the other: 427 -> 445 three-scale img/s on one box, bit-identical)')
    ap.add_argument('--ms-size', type=int, default=1200, help='multiscale: side of the (square) source images')
    ap.add_argument('--db-rows', type=int, default=1006322, help='distractors: database size (RParis6K + 1M)')
    ap.add_argument('--queries', type=int, default=70)
    ap.add_argument('--sim-general', action='store_true',
                    help='distractors: the six-product bf16 similarity kernel (any fp32 operands) instead of the fp16-pair one')
    ap.add_argument('--exchange', default='descriptors', choices=['descriptors', 'scores'],
                    help="distractors: what crosses xGMI - the [N/W, 2048] descriptor blocks (north_star) or, the "
                         "cheaper layout, each rank's [Q, N/W] score block")
    ap.add_argument('--profile-every', type=int, default=10,
                    help='record per-launch HIP events on every n-th timed step (1 = all steps)')
    ap.add_argument('--layers', action='store_true', help='also print the per-layer profile to stderr')
    ap.add_argument('--dry-launch', action='store_true',
                    help='exercise only the multi-process plumbing of --gpus N (self-spawn, rendezvous on 127.0.0.1, the one '
                         'all-gather of descriptor blocks through dirtorch_amd.distributed) on the CPU over gloo: no GPU, no timing')
    ap.add_argument('--dump-launches', default='',
                    help='write the launch sequence of one forward (name, kernel, flops, bytes, avg ms) as JSON: '
                         'scripts/summarize_prof.py aligns rocprofv3 dispatches with it')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without torch.distributed.run: become the launcher (one rank per GPU, the reference's
        # nn.DataParallel fan-out, dirtorch/utils/common.py:150-175, as N processes)
        sys.exit(self_spawn(args.gpus, sys.argv[1:], dry=args.dry_launch))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        args.gpus = world
    if args.dry_launch:
        return dry_launch(world, rank)
    assert torch.cuda.is_available(), 'bench.py needs an MI355X'
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or 'RANK' in os.environ:      # under torch.distributed.run the RCCL path runs even at N = 1
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))   # RCCL

    if args.workload in ('distractors', 'multiscale'):
        out = (bench_distractors if args.workload == 'distractors' else bench_multiscale)(args, world, rank, dist)
        if out is not None:
            print(json.dumps(out))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    import synth
    from dirtorch_amd import nets
    B, S, K, Wm = args.batch, args.size, args.steps, args.warmup
    sd_timed = bench_state_dict(args.arch, S, args.checkpoint, dist, rank)
    net = nets.create_model(args.arch + '_rmac', pretrained='')
    net.load_state_dict(sd_timed)
    net.compute_dtype = args.dtype
    net.cuda().eval()

    g = torch.Generator(device='cuda').manual_seed(1234 + rank)
    x_f32 = torch.randn(B, 3, S, S, generator=g, device='cuda')   # normalised-image statistics (the feed of rounds 1-5)
    # the uint8 feed: pictures with structure (tests/synth.py: low-frequency pattern + noise), eight distinct ones tiled over
    # the batch and rolled per row so that no two batch rows are equal - raw pixels as PIL would hand them over
    pics = to_uint8_nhwc(synth.synth_images(1234 + rank, min(B, 8), S, S)).cuda()
    x_u8 = torch.stack([torch.roll(pics[i % pics.shape[0]], shifts=(7 * (i // pics.shape[0]), 13 * (i // pics.shape[0])), dims=(0, 1))
                        for i in range(B)])
    del pics
    x = x_u8 if args.input == 'u8' else x_f32
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

    net.set_profiling(256 * (K // max(args.profile_every, 1) + 2))    # event pairs pre-created: nothing is allocated while timing
    # The host is ~40x ahead of the GPU here (0.3 ms to enqueue a 13 ms step) - unless the interpreter's cycle collector
    # picks the first timed step for a full collection of the process's heap (torch + the checkpoint: 35-55 ms measured,
    # gpurun_out/r4f-r4g: the GPU then idles with an empty queue and every step of a 12-step run reads 3 ms slower).
    # Collect now, keep the collector out of the timed region.
    gc.collect()
    gc.disable()
    allb = None
    if dist is not None:
        # the exchange buffer exists and RCCL has run this very collective once (communicator channels, kernel load, buffer
        # registration: first-call costs of tens of ms that belong to start-up, not to the K steps) before the clock starts
        allb = torch.empty(world * K * B, D, device='cuda')
        dist.all_gather_into_tensor(allb, shard)
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _trace = [] if os.environ.get('DIRTORCH_AMD_BENCH_TRACE') else None   # (debug: host enqueue time of every step)
    for k in range(K):
        net.pause_profiling(k % args.profile_every != 0)   # event records only on sampled steps
        _a = time.perf_counter()
        shard[k * B:(k + 1) * B] = net(x)
        if _trace is not None:
            _trace.append(round((time.perf_counter() - _a) * 1e3, 2))
    if _trace is not None:
        print('enqueue ms per step:', _trace, file=sys.stderr)
    if dist is not None:
        dist.all_gather_into_tensor(allb, shard)                  # one exchange step (RCCL), inside the timed region
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    gc.enable()
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
        traffic_note = 'no profiles/traffic.json'
        if os.path.isfile(tfile):
            try:
                traffic_all = json.load(open(tfile))
                sys.path.insert(0, os.path.join(ROOT, 'scripts'))
                import summarize_prof
                built, now = traffic_all.pop('_build', None), summarize_prof.csrc_hash()
                traffic_note = 'PMC pass (2 x FETCH_SIZE + WRITE_SIZE) of build %s' % built
                if built != now:        # PMC bytes of other kernel sources are not a measurement of this library
                    traffic_all, traffic_note = None, 'stale: profiles/traffic.json is from build %s, this is %s' % (built, now)
            except Exception:
                traffic_all, traffic_note = None, 'profiles/traffic.json unreadable'
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
                'traffic': traffic, 'traffic_source': traffic_note,
                'note': 'frac = the largest-time-share kernel against ITS roof; whole step = step_mfma_frac of the MFMA peak',
                # measured once on a pool box (profiles/r02_mfma_ceiling.txt): bare register-only MFMA loop, 8 waves/CU
                'mfma_ceiling_measured_tflops': {'random_operands': 1582, 'zero_operands': 2285},
                'hbm_ceiling_measured_gbs': {'read': 6305, 'copy_1to1': 5147, 'read_write_4to1': 5100},   # profiles/r02_hbm_ceiling.txt, r03_read_store_mix_probe.txt
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
        cpu, precision, workloads = None, None, None
        side = {}
        sd_cal = sd_timed if args.checkpoint == 'calibrated' else None
        if world == 1:
            # the same step on the OTHER feed and on the OTHER checkpoint (20 steps each, median over groups of 5; the dominant
            # kernel's average launch from the profile records of 4 more steps)
            other_x = x_f32 if args.input == 'u8' else x_u8
            med, lo_, hi_ = grouped_rate(lambda: net(other_x), B)
            side['%s_feed_images_per_sec' % ('fp32' if args.input == 'u8' else 'u8')] = med
            side['%s_feed_min_max' % ('fp32' if args.input == 'u8' else 'u8')] = [lo_, hi_]
            del other_x
            other_kind = 'he' if args.checkpoint == 'calibrated' else 'calibrated'
            if other_kind == 'he' or args.cpu_seconds > 0:       # (calibrating costs one fp32 CPU forward of two images)
                sd_other = bench_state_dict(args.arch, S, other_kind)
                if other_kind == 'calibrated':
                    sd_cal = sd_other
                net2 = nets.create_model(args.arch + '_rmac', pretrained='')
                net2.load_state_dict(sd_other)
                net2.compute_dtype = args.dtype
                net2.cuda().eval()
                med, lo_, hi_ = grouped_rate(lambda: net2(x), B)
                tag = 'he_init' if other_kind == 'he' else 'calibrated'
                side[tag + '_images_per_sec'] = med
                side[tag + '_min_max'] = [lo_, hi_]
                net2.set_profiling(1024)
                for _ in range(4):
                    net2(x)
                torch.cuda.synchronize()
                doms = [r['ms'] for r in net2.get_profile() if r['kernel'] == dom]
                net2.set_profiling(False)
                if doms:       # the dominant kernel is power-bound and data-dependent: its launch average on the other checkpoint's activations
                    side[tag + '_dominant_kernel_avg_launch_ms'] = round(sum(doms) / len(doms), 5)
                    side[tag + '_dominant_kernel_frac'] = round(dfl / dn / (sum(doms) / len(doms) * 1e-3) / 1e12 / peak_tf, 4) if not hbm_bound else None
                del net2
        del x_f32, x_u8
        if world == 1 and args.cpu_seconds > 0:
            del shard
            net._ws = {}
            torch.cuda.empty_cache()
            if args.no_precision:
                cpu, _ = cpu_baseline(args.arch, S, args.cpu_seconds)
            else:
                precision, cpu = precision_leg(args.arch, S, B, x, args.cpu_seconds, args.dtype, feed=args.input,
                                               sd_cal=sd_cal, sd_rate=sd_timed)
            if not args.no_workloads:
                workloads = other_workloads(args, x)
        value = world * B * K / el
        out = {
            'metric': 'images/sec descriptor extraction (%s-GeM, %dx%d)' % (args.arch, S, S),
            'value': round(value, 2), 'unit': 'images/sec', 'n_gpus': world, 'steps': K, 'warmup': Wm,
            'ms_per_step': round(el / K * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': 'configs[1]: %s_rmac (AP-GeM head) single-scale %dx%d descriptor extraction, '
                                   '1 process per GPU, synthetic weights + images' % (args.arch, S, S),
                       'batch_per_gpu': B, 'global_batch': world * B,
                       'input': ('uint8 NHWC images resident in HBM; ToTensor + Normalize on the device inside the step (the drop-in '
                                 'CLIs\' feed, transforms.py:617-623)' if args.input == 'u8' else 'fp32 NCHW (normalised) resident in HBM'),
                       'timed_checkpoint': ('BatchNorm-calibrated synthetic (tests/synth.py; the checkpoint one_minus_cos is quoted on)'
                                            if args.checkpoint == 'calibrated' else
                                            'random-init (He-normal) synthetic (tests/synth.py), as rounds 1-5; the BatchNorm-calibrated one, on which '
                                            'one_minus_cos is quoted, is timed beside it: calibrated_images_per_sec'),
                       **side,
                       'gflop_per_image': GFLOP_PER_IMG.get((args.arch, S)),
                       # ---- parity of THIS line's dtype as flat scalars (measured in this run, outside the timed region):
                       # 1 - cos of the engine's descriptors vs the fp32 CPU oracle on the BatchNorm-calibrated checkpoint,
                       # two images travelling inside the timed batch; the north-star tolerance it has to meet; and the bf16
                       # numbers (the dtype BASELINE configs[1] names, which cannot meet that tolerance: BASELINE.md section 0)
                       'one_minus_cos': (precision or {}).get('one_minus_cos', {}).get(args.dtype),
                       'tolerance': 1e-4,
                       'meets_tolerance': (None if not precision else bool(precision['one_minus_cos'][args.dtype] < 1e-4)),
                       'd_map': (precision or {}).get('map', {}).get('d_map_' + args.dtype),
                       'bf16_images_per_sec': (round(value, 2) if args.dtype == 'bf16' else
                                               (precision or {}).get('images_per_sec', {}).get('bf16')),
                       'bf16_one_minus_cos': (precision or {}).get('one_minus_cos', {}).get('bf16'),
                       'fp16_images_per_sec': (round(value, 2) if args.dtype == 'fp16' else
                                               (precision or {}).get('images_per_sec', {}).get('fp16')),
                       'fp16_one_minus_cos': (precision or {}).get('one_minus_cos', {}).get('fp16'),
                       'tflops_per_gpu': round(value / world * GFLOP_PER_IMG.get((args.arch, S), 0) / 1e3, 1),
                       'parallelism': 'image-parallel shards, 1 all-gather of descriptors' if world > 1 else 'single GPU',
                       # the other storage formats on the same step, and what each loses against the fp32 CPU
                       # oracle (descriptors at the bench size + mAP through whitening / similarity / AP)
                       'precision': precision,
                       # BASELINE configs[3] / configs[4] on this GPU, a few steps each outside the timed region
                       # (the full lines: --workload distractors / multiscale)
                       'workloads': workloads,
                       'rccl_ranks': dist.get_world_size() if dist is not None else 0},
            'roofline': roof, 'cpu_baseline': cpu,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
