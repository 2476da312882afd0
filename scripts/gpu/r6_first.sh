#!/bin/bash
# Round 6, first GPU call: (1) the DPP whole-wave shift probe the u8 stem's register pooling rests on, (2) the multi-rank code on the
# one GPU of the box - torch.distributed.run --nproc-per-node 1: RCCL init with device_id, the all-gather of the real buffers, both
# workloads, both exchange layouts, both DIRTORCH_AMD_EXCHANGE values (review item 4), (3) the whole -m gpu suite with durations.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r6first}; mkdir -p $O
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/probes/dpp_wave_shift.hip -o /tmp/dpp_probe 2>/dev/null && /tmp/dpp_probe > $O/dpp_wave_shift.txt 2>&1
cat $O/dpp_wave_shift.txt | cut -c1-400
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611"
timeout 600 $TR bench.py --gpus 1 --steps 20 --warmup 3 --cpu-seconds 0 > $O/ws1_extract.json 2> $O/ws1_extract.err; echo "ws1 extract rc=$?"
for ex in descriptors scores; do for algo in rccl mesh; do
  DIRTORCH_AMD_EXCHANGE=$algo timeout 600 $TR bench.py --gpus 1 --workload distractors --exchange $ex --steps 10 --warmup 2 --cpu-seconds 0 \
      > $O/ws1_distractors_${ex}_${algo}.json 2> $O/ws1_distractors_${ex}_${algo}.err; echo "ws1 distractors $ex $algo rc=$?"
done; done
timeout 600 $TR bench.py --gpus 1 --workload multiscale --steps 5 --warmup 2 > $O/ws1_multiscale.json 2> $O/ws1_multiscale.err; echo "ws1 multiscale rc=$?"
python - <<P
import json, glob
for f in sorted(glob.glob('$O/ws1_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['value'], d['unit'], d['ms_per_step'], 'rccl_ranks', d['config'].get('rccl_ranks'), d['config'].get('exchange'), d['config'].get('exchange_algo'))
    except Exception as e:
        print(f, 'ERR', e); print(open(f.replace('.json', '.err')).read()[-1500:])
P
timeout 2400 python -m pytest tests -m gpu -q -s -p no:cacheprovider --durations=60 > $O/pytest.log 2>&1
echo "pytest rc=$?"
grep -a " passed\| failed\|^FAILED\|^ERROR" $O/pytest.log | tail -30
grep -a -A70 "slowest 60 durations" $O/pytest.log | cut -c1-160
