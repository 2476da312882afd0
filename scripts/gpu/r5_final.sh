#!/bin/bash
# round 5, last pass on the final kernel sources: the tests of the kernels touched since the full suite (conv_patchw), then bench + the four rocprofv3 passes
TAG=${1:-r5fin}
mkdir -p gpurun_out/$TAG
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_scale_gpu.py -m gpu -q -x -p no:cacheprovider -k "patchw or patch3x3w or full_size or timed or kernel_mix" > gpurun_out/$TAG/pytest_patchw.log 2>&1; tail -2 gpurun_out/$TAG/pytest_patchw.log
SKIP_TESTS=1 bash scripts/gpu/validate.sh $TAG
