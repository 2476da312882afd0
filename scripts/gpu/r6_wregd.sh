#!/bin/bash
# Round 6: the two-source register-stationary GEMM of layer2's first block (conv_wregd.hip): op-level parity, standalone timing
# against the DUAL ring it replaces (+ the phase builds when present), then the A/B inside the network.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r6wregd}; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -p no:cacheprovider -k "two_source" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 300 python scripts/exp_dual_time.py "" DIRTORCH_AMD_NO_WREGD=1 2>&1 | grep -v amdgpu.ids | tee $O/dual_time.txt
for b in 1 2 3 8 9 10; do
  L=$PWD/scripts/_exp/lib_conv_wregd_$b.so; [ -f "$L" ] || continue
  EXP_SHAPES=l2.0 DIRTORCH_AMD_LIB=$L timeout 200 python scripts/exp_dual_time.py 2>&1 | grep -v "amdgpu.ids\|identical" | sed "s/^/DIR_WREGD_ABL=$b  /" | tee -a $O/wregd_phases.txt
done
BENCH_ARGS="--steps 30 --warmup 5" bash scripts/gpu/ab.sh DIRTORCH_AMD_NO_WREGD=1 'layer2\.0' ${1:-r6wregd} 2>&1 | tee $O/ab.txt
