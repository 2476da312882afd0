#!/bin/bash
# round 5, the very last pass (in-place identity blocks became the default): the whole -m gpu suite, then bench + the four rocprofv3 passes
TAG=${1:-r5fin2}
bash scripts/gpu/validate.sh $TAG
