#!/bin/bash
# Round-3 experiment batch 3: ring kernel (conv_ring.hip) tests + A/B, sorted-probe ranking tests + distractor bench,
# row-pattern probe.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3d
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ranking_gpu.py tests/test_ops_gpu.py -k "ring or rank or device_ap or million or many_probes" -m gpu -q -x -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -15 $O/pytest.log | cut -c1-300
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value scripts/probes/window_probe.hip -o /tmp/window_probe && timeout 120 /tmp/window_probe > $O/window_probe.txt; grep "rows" $O/window_probe.txt | head -8
B="python bench.py --cpu-seconds 0 --steps 30 --warmup 5"
for rep in 1 2; do
  DIRTORCH_AMD_NO_RING=1 $B > $O/ab_base_$rep.json 2>/dev/null
  $B > $O/ab_ring_$rep.json 2>/dev/null
  DIRTORCH_AMD_RING_KEEP_X3=1 $B > $O/ab_ringx3_$rep.json 2>/dev/null
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3d/ab_*.json')):
    try:
        d=json.load(open(f)); rows={ (r[0],r[1]):r[3] for r in d['roofline']['kernels']['rows']}
        print(f.split('/')[-1], d['value'], d['ms_per_step'], {k[1]:v for k,v in rows.items() if 'conv1' in k[1] or 'ring' in k[0]})
    except Exception as e: print(f, 'ERR', e)
P
timeout 300 python bench.py --workload distractors --steps 10 --warmup 2 --cpu-seconds 0 > $O/distractors.json 2> $O/distractors.err; echo "distractors rc=$?"; cut -c1-1500 $O/distractors.json; tail -3 $O/distractors.err
