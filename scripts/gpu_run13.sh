#!/bin/bash
# head variants (FPN, classifier) + regression of the model-level tests after the block-loop change
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_heads_gpu.py tests/test_model_gpu.py tests/test_pipeline_gpu.py -q -m gpu -x > gpurun_out/heads_tests.log 2>&1
echo "exit $?" >> gpurun_out/heads_tests.log
tail -15 gpurun_out/heads_tests.log
timeout 300 python bench.py --steps 8 --warmup 2 --cpu-seconds 0 > gpurun_out/bench_after_heads.log 2>&1
tail -2 gpurun_out/bench_after_heads.log
