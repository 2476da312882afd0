set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>&1 | head -8 > gpurun_out/gpu.txt
nproc >> gpurun_out/gpu.txt
timeout 900 python -m pytest tests -m gpu -q --tb=short --maxfail=60 -p no:cacheprovider > gpurun_out/pytest1.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest1.log
tail -5 gpurun_out/pytest1.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke1.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke1.log
timeout 600 python bench.py --layers > gpurun_out/bench1.json 2> gpurun_out/bench1.err; echo "bench exit $?" >> gpurun_out/bench1.err
cat gpurun_out/bench1.json
