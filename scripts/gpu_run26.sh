#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -x -k "descriptor_limit or batch_composition" 2>&1 | grep -v "Warning\|pin_memory" | tail -5
