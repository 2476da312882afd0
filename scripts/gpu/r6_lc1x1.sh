#!/bin/bash
# round 6: the persistent 256 x 256 1x1 ring with loader / consumer roles (conv_persistlc.hip): parity, standalone timing against
# conv_persist.hip on the wide 1x1 shapes and on the two-source GEMMs, A/B inside the network (DIRTORCH_AMD_LC1X1=1 puts
# the new kernel wherever conv_persist.hip's residual-free and two-source forms run)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r6lc1x1}; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q -x -p no:cacheprovider -k "lc1x1 or two_source or fused_seams" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
EXP_SHAPES=l3.conv1,l4.conv1,l3.0.conv1 timeout 300 python scripts/exp_conv_time.py 256x256_persist1x1 256x256_persist1x1_x3 256x256_lc1x1 2>&1 | grep -v amdgpu.ids | tee $O/conv_time.txt
timeout 300 python scripts/exp_dual_time.py "" "DIRTORCH_AMD_LC1X1=1 DIRTORCH_AMD_NO_WREGD=1" 2>&1 | grep -v amdgpu.ids | tee $O/dual_time.txt
BENCH_ARGS="--steps 30 --warmup 5" bash scripts/gpu/ab.sh DIRTORCH_AMD_LC1X1=1 'layer3\.0|layer4\.|layer3\.1\.conv1' ${1:-r6lc1x1} 2>&1 | tee $O/ab.txt
