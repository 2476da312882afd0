#!/bin/bash
# Round 6: the uint8-feed stem (stem_u8.hip) - parity tests, then the bench line on both feeds with per-layer times.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r6stem}; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_stem_u8_gpu.py tests/test_comm_gpu.py -m gpu -q -s -p no:cacheprovider -x > $O/pytest_stem.log 2>&1; echo "pytest stem rc=$?"
grep -a "^\[stem\| passed\| failed\|^FAILED\|^ERROR\|Error\|assert" $O/pytest_stem.log | tail -40
if [ -z "$SKIP_SCALE" ]; then
timeout 900 python -m pytest tests/test_scale_gpu.py -m gpu -q -s -p no:cacheprovider -k "timed" --durations=10 > $O/pytest_timed.log 2>&1; echo "pytest timed rc=$?"
grep -a "^\[timed\| passed\| failed\|^FAILED\|^ERROR\|Error\|s call" $O/pytest_timed.log | tail -30
fi
for SEG in ${SEGS:-0}; do
DIRTORCH_AMD_STEM_U8_SEG=$SEG timeout 600 python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --layers > $O/bench_u8_seg$SEG.json 2> $O/bench_u8_seg$SEG.txt; echo "bench u8 seg=$SEG rc=$?"
python - <<P
import json
d = json.loads(open('$O/bench_u8_seg$SEG.json').read().strip().splitlines()[-1])
print('u8 feed seg=$SEG:', d['value'], 'img/s', d['ms_per_step'], 'ms;', {k: v for k, v in d['config'].items() if 'images_per_sec' in k})
P
grep -a "prep_input\|conv1+maxpool\|layer1.0" $O/bench_u8_seg$SEG.txt | head -8
done
if [ -n "$FULL_BENCH" ]; then
timeout 900 python bench.py --layers > $O/bench_full.json 2> $O/bench_full.txt; echo "bench full rc=$?"; tail -c 3000 $O/bench_full.json
fi
