#!/bin/bash
# round 5, phase 1: the -m gpu suite + the default bench line (fp16p) + the in-place / Infinity-Cache probe of layers 3-4
TAG=${1:-r5p1}
mkdir -p gpurun_out/$TAG
timeout 300 python scripts/exp_inplace.py > gpurun_out/$TAG/inplace.txt 2>&1
tail -4 gpurun_out/$TAG/inplace.txt | cut -c1-400
SKIP_PROF=1 bash scripts/gpu/validate.sh $TAG
