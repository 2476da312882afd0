#!/bin/bash
# Round-3 experiment 6: loader / consumer similarity kernel (tests + A/B on the distractor workload), two-source GEMM tile A/B.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3i
mkdir -p $O
timeout 600 python -m pytest tests/test_ranking_gpu.py -k "split or million or widths" -m gpu -q -x -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest.log | cut -c1-300
D="python bench.py --workload distractors --steps 10 --warmup 2 --cpu-seconds 0"
for rep in 1 2; do
  DIRTORCH_AMD_SIM_V1=1 $D > $O/dist_v1_$rep.json 2>/dev/null
  $D > $O/dist_lc_$rep.json 2>/dev/null
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3i/dist_*.json')):
    d=json.load(open(f)); print(f.split('/')[-1], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['rank_ap_ms'])
P
B="python bench.py --cpu-seconds 0 --steps 30 --warmup 5"
for rep in 1 2; do
  $B > $O/ab_base_$rep.json 2>/dev/null
  DIRTORCH_AMD_DUAL_VARIANT=128x256_w2x4_s3 $B > $O/ab_dual_$rep.json 2>/dev/null
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3i/ab_*.json')):
    try:
        d=json.load(open(f))
        print(f.split('/')[-1], d['value'], d['ms_per_step'], [(r[0],r[1],r[3]) for r in d['roofline']['kernels']['rows'] if 'dual' in r[0]])
    except Exception as e: print(f, 'ERR', e)
P
