set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { tag=$1; shift; timeout 400 python bench.py "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$tag', d['value'],'img/s', d['ms_per_step'],'ms/step', 'conv TF/s', r['all_conv_tflops'], 'cpu', d['cpu_baseline'])" >> gpurun_out/bench8.txt; }
run fp16_1024_b32 --dtype fp16 --cpu-seconds 0
run r50_224_b64 --arch resnet50 --size 224 --batch 64 --cpu-seconds 8
run r50_224_b1 --arch resnet50 --size 224 --batch 1 --cpu-seconds 0 --steps 50
run r101_848_b32 --size 848 --batch 32 --cpu-seconds 0
run r101_1200_b16 --size 1200 --batch 16 --cpu-seconds 0
run r101_1697_b8 --size 1697 --batch 8 --cpu-seconds 0
run r101_1024x768ish_b1 --size 1024 --batch 1 --cpu-seconds 0 --steps 50
cat gpurun_out/bench8.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
