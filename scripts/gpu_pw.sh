#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "test_conv_variant_vs_oracle and 3x3" 2>&1 | tail -4
bash scripts/gpu_r2_env_ab.sh DIRTORCH_AMD_PATCHW=1 "layer2.(1|2).conv2|layer3.(1|2|3).conv2|layer4.*conv2" pw
