set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --tb=short --maxfail=30 -p no:cacheprovider > gpurun_out/pytest3.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest3.log
tail -4 gpurun_out/pytest3.log
timeout 600 python bench.py --layers --cpu-seconds 0 > gpurun_out/bench3.json 2> gpurun_out/bench3.err; echo "bench exit $?" >> gpurun_out/bench3.err
timeout 300 python bench.py --batch 32 --cpu-seconds 0 > gpurun_out/bench3_b32.json 2> gpurun_out/bench3_b32.err
cat gpurun_out/bench3.json gpurun_out/bench3_b32.json
