set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
cat /sys/fs/cgroup/cpu.max > gpurun_out/cpu.txt 2>&1; python -c "import os;print(len(os.sched_getaffinity(0)))" >> gpurun_out/cpu.txt; grep -m1 "model name" /proc/cpuinfo >> gpurun_out/cpu.txt
timeout 900 python -m pytest tests -m gpu -q --tb=short --maxfail=30 -p no:cacheprovider > gpurun_out/pytest2.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest2.log
tail -3 gpurun_out/pytest2.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke2.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke2.log; tail -2 gpurun_out/smoke2.log
for b in 4 8 32; do
  timeout 300 python bench.py --batch $b --cpu-seconds 0 > gpurun_out/bench2_b$b.json 2> gpurun_out/bench2_b$b.err
done
timeout 600 python bench.py --layers > gpurun_out/bench2.json 2> gpurun_out/bench2.err; echo "bench exit $?" >> gpurun_out/bench2.err
cat gpurun_out/bench2*.json
