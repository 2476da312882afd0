#!/bin/bash
mkdir -p gpurun_out
V="256x256_w4x2 256x256_w4x2_f1 256x256_w4x2_f1p 256x256_w4x2_p 256x256_w4x4 256x256_w4x4_f1p"
python scripts/exp_conv_time.py $V 2>&1 | grep -v "amdgpu.ids\|^lib" | tee gpurun_out/exp_full.txt
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -x 2>&1 | tail -2
