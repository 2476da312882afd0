#!/bin/bash
# layer3 seam kernel: op-level parity, then timing of the full build and of the experiment builds named on the command line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "layer3_seam" 2>&1 | tail -n 3
python scripts/exp_seam3_time.py
for b in "$@"; do DIRTORCH_AMD_LIB=scripts/_exp/lib_conv_seam3_$b.so python scripts/exp_seam3_time.py; done
