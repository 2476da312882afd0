#!/bin/bash
# Round-3 experiment batch 2: request-window probe, reverse-tile-order A/B, batch-1 streams, distractor workload.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3c
mkdir -p $O
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value scripts/probes/window_probe.hip -o /tmp/window_probe && timeout 120 /tmp/window_probe | tee $O/window_probe.txt
B="python bench.py --cpu-seconds 0 --steps 30 --warmup 5"
for rep in 1 2; do
  $B > $O/ab_base_$rep.json 2>/dev/null
  DIRTORCH_AMD_REV_CONV1=1 $B > $O/ab_rev1_$rep.json 2>/dev/null
  DIRTORCH_AMD_REV_CONV3=1 $B > $O/ab_rev3_$rep.json 2>/dev/null
  DIRTORCH_AMD_REV_CONV1=1 DIRTORCH_AMD_REV_CONV3=1 $B > $O/ab_rev13_$rep.json 2>/dev/null
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3c/ab_*.json')):
    try:
        d=json.load(open(f)); rows={ (r[0],r[1]):r[3] for r in d['roofline']['kernels']['rows']}
        print(f.split('/')[-1], d['value'], d['ms_per_step'], 'l3c1', rows.get(('256x256_persist1x1','layer3.conv1')), 'l3c3', rows.get(('64x512_wreg1x1','layer3.conv3')), 'l3c2', rows.get(('512x128_patch3x3w','layer3.conv2')), 'l2c3', rows.get(('64x512_wreg1x1','layer2.conv3')))
    except Exception as e: print(f, 'ERR', e)
P
timeout 300 python scripts/bench_batch1.py > $O/batch1.json 2> $O/batch1.err; echo "batch1 rc=$?"; cat $O/batch1.json; tail -3 $O/batch1.err
timeout 300 python bench.py --workload distractors --steps 10 --warmup 2 > $O/distractors.json 2> $O/distractors.err; echo "distractors rc=$?"; cat $O/distractors.json; tail -3 $O/distractors.err
timeout 300 python bench.py --workload distractors --exchange scores --steps 10 --warmup 2 --cpu-seconds 0 > $O/distractors_scores.json 2>> $O/distractors.err; cat $O/distractors_scores.json
