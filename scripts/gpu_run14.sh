#!/bin/bash
# device-side Scale (resize kernel) + fused multi-scale extraction
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_resize_gpu.py tests/test_pipeline_gpu.py tests/test_heads_gpu.py -q -m gpu -x > gpurun_out/resize_tests.log 2>&1
echo "exit $?" >> gpurun_out/resize_tests.log
grep -v Warning gpurun_out/resize_tests.log | tail -25
