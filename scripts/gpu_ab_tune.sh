#!/bin/bash
# A/B on one box.  OLD = a tuning table replayed (DIRTORCH_AMD_TUNE_CACHE), NEW = the built-in heuristic.
mkdir -p gpurun_out
pick='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"])'
cp ${OLD_TABLE:-profiles/r01_tuned_variants_b32_1024.txt} /tmp/old_tune.txt
for i in 1 2 3; do
  echo -n "old "; DIRTORCH_AMD_TUNE_CACHE=/tmp/old_tune.txt timeout 600 python bench.py --cpu-seconds 0 2>/dev/null | tail -1 | python -c "$pick"
  echo -n "new "; timeout 600 python bench.py --cpu-seconds 0 2>/dev/null | tail -1 | python -c "$pick"
done
