#!/bin/bash
cd "$GRAFT_REPO_ROOT"
TAG=${1:-r2g}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -s -p no:cacheprovider -k "blocked or beyond or register_stationary or seams" > $O/pytest_new.log 2>&1
echo "pytest(new) rc=$?" | tee $O/pytest_new.rc
grep -a " passed\| failed\|^FAILED\|^ERROR\|Error\|^E  " $O/pytest_new.log | tail -20
run() {
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --layers > $O/bench_$name.json 2> $O/layers_$name.txt
  echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1])
    print(d['value'], 'img/s', d['ms_per_step'], 'ms/step')
except Exception as e:
    print('no result', e)
PY
)"
}
run noblk DIRTORCH_AMD_NO_BLK=1
run blk DIRTORCH_AMD_X=1
run noblk2 DIRTORCH_AMD_NO_BLK=1
run blk2 DIRTORCH_AMD_X=1
grep -a "layer3.\(1\|2\|5\|21\|22\).conv[13]" $O/layers_noblk.txt $O/layers_blk.txt | cut -c1-150
