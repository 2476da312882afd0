#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python scripts/exp_seam3_time.py
for b in "$@"; do DIRTORCH_AMD_LIB=scripts/_exp/lib_conv_seam3_$b.so python scripts/exp_seam3_time.py; done
