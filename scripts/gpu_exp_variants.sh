#!/bin/bash
# per-variant timing of single conv launches on the layer2-4 shapes (scripts/exp_conv_time.py)
mkdir -p gpurun_out
V="${VARIANTS:-64x512_wreg1x1 128x256_w2x4_s3_k32 256x128_w4x2_s3_k32 256x256_w4x2 256x256_persist1x1}"
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "conv" 2>&1 | tail -3
python scripts/exp_conv_time.py $V 2>&1 | grep -v "amdgpu.ids\|^lib" | tee gpurun_out/exp_variants.txt
