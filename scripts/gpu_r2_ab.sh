#!/bin/bash
# Round-2 A/B on the GPU box: targeted tests of the new kernels, then bench.py with the fused seam kernel
# (conv_c3c1) and the deep-X persistent 1x1 (persist1x1_x3) switched on/off one at a time.
cd "$GRAFT_REPO_ROOT"
TAG=${1:-r2b}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_ranking_gpu.py tests/test_pipeline_gpu.py tests/test_scale_gpu.py tests/test_comm_gpu.py \
  -m gpu -q -s -p no:cacheprovider -k "seam or deep_x or many_probes or expand or trunk_tiles or roxford or comm or million or widths or persist" > $O/pytest_new.log 2>&1
echo "pytest(new) rc=$?" | tee $O/pytest_new.rc
grep -a "^\[scale\|^\[pipeline\| passed\| failed\|^FAILED\|^ERROR\|Error\|bad elements" $O/pytest_new.log | tail -40
run() {  # name, env...
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
run base DIRTORCH_AMD_C3C1=0 DIRTORCH_AMD_NO_X3=1
run x3 DIRTORCH_AMD_C3C1=0
run c3c1 DIRTORCH_AMD_NO_X3=1
run both DIRTORCH_AMD_X=1
run base2 DIRTORCH_AMD_C3C1=0 DIRTORCH_AMD_NO_X3=1
run both2 DIRTORCH_AMD_X=1
grep -a "c3c1\|layer3.1.conv1 \|layer3.0.conv1\|layer3.0.down\|layer4.1.conv1\|layer1.1.conv\|layer2.1.conv[13]" $O/layers_base.txt $O/layers_both.txt | cut -c1-150
