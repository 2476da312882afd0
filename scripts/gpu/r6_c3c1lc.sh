#!/bin/bash
# round 6: layer1's DS seam with loader / consumer roles (conv_c3c1lc.hip): parity against the one-role kernel, standalone timing, A/B in the network
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r6c3c1lc}; mkdir -p $O
timeout 900 python -m pytest tests/test_pair_gpu.py tests/test_ops_gpu.py -m gpu -q -x -p no:cacheprovider -k "downsample_seam or c3c1_ds or ds_seam or seam" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 300 python scripts/exp_ds_seam_time.py 2>&1 | grep -v amdgpu.ids | tee $O/ds_seam_time.txt
BENCH_ARGS="--steps 30 --warmup 5" bash scripts/gpu/ab.sh DIRTORCH_AMD_NO_C3C1LC=1 'layer1\.0' ${1:-r6c3c1lc} 2>&1 | tee $O/ab.txt
