set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for cfg in "1 " "1 --no-autotune" "4 --no-autotune" "32 --no-autotune"; do
  set -- $cfg
  timeout 300 python bench.py --batch $1 $2 --cpu-seconds 0 --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('B=$1 $2', d['value'],'img/s', d['ms_per_step'],'ms/step kernels', d['roofline']['all_kernels_ms_per_step'])" >> gpurun_out/bench6.txt
done
cat gpurun_out/bench6.txt
timeout 600 python scripts/bench_rank.py > gpurun_out/bench_rank.json 2> gpurun_out/bench_rank.err; cat gpurun_out/bench_rank.json; tail -3 gpurun_out/bench_rank.err
