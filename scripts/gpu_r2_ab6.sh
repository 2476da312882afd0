#!/bin/bash
cd "$GRAFT_REPO_ROOT"
TAG=${1:-r2h}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -s -p no:cacheprovider -k "patch3x3s or full_size or (conv_variant and 3x3_wide)" > $O/pytest_new.log 2>&1
echo "pytest(new) rc=$?" | tee $O/pytest_new.rc
grep -a " passed\| failed\|^FAILED\|^ERROR\|Error\|^E  \|out of tolerance" $O/pytest_new.log | tail -20
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
run nops DIRTORCH_AMD_NO_PATCHS=1
run ps DIRTORCH_AMD_X=1
run nops2 DIRTORCH_AMD_NO_PATCHS=1
run ps2 DIRTORCH_AMD_X=1
grep -a "layer3.\(1\|5\).conv2\|layer4.\(1\|2\).conv2" $O/layers_nops.txt $O/layers_ps.txt | cut -c1-150
