#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "stem" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_scale_gpu.py -x -q -k "golden or tiles or r50_224" 2>&1 | tail -4
bash scripts/gpu_r2_env_ab.sh DIRTORCH_AMD_STEM_V1=1 "maxpool" stem
