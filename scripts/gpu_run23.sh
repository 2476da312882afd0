#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -q -m gpu -x > gpurun_out/ops23.log 2>&1; echo "exit $?" >> gpurun_out/ops23.log; tail -2 gpurun_out/ops23.log
V="256x256_w4x2 256x256_w4x4 256x256_w4x2_s3_k32 128x256_w2x4_s3_k32 256x128_w4x2_s3_k32 128x128_w2x2"
python scripts/exp_conv_time.py $V 2>&1 | grep -v "amdgpu.ids\|^lib" | tee gpurun_out/exp_korder.txt
pick='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["all_conv_ms_per_step"])'
rm -f /tmp/new_tune.txt
for i in 1 2 3; do
  echo -n "new "; DIRTORCH_AMD_TUNE_CACHE=/tmp/new_tune.txt timeout 600 python bench.py --cpu-seconds 0 2>/dev/null | tail -1 | python -c "$pick"
done
cp /tmp/new_tune.txt gpurun_out/tune_b32_v5.txt
