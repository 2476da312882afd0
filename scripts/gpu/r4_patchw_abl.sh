#!/bin/bash
# what a persistent conv_patch3x3w could hide: the kernel with its epilogue / prologue wait compiled out (timing only)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export EXP_SHAPES=l3.conv2,l2.conv2,l4.conv2
python scripts/exp_conv_time.py 512x128_patch3x3w 2>/dev/null | grep -v amdgpu
for b in 1 2 4 6; do DIRTORCH_AMD_LIB=scripts/_exp/lib_conv_patchw_$b.so python scripts/exp_conv_time.py 512x128_patch3x3w 2>/dev/null | grep -v amdgpu; done
