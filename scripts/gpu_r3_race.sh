#!/bin/bash
# Shape of the multi-stream mismatch (scripts/exp_stream_race.py prints where a differing map differs).
cd "$GRAFT_REPO_ROOT"
export RACE_SIZES=1024x1024 RACE_STREAMS=6 RACE_REPS=20
for rep in 1 2 3; do GPU_MAX_HW_QUEUES=8 python scripts/exp_stream_race.py 2>&1 | grep -v amdgpu | cut -c1-400; done
