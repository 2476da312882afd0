set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --tb=short --maxfail=30 -p no:cacheprovider > gpurun_out/pytest5.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest5.log
tail -4 gpurun_out/pytest5.log
timeout 600 python bench.py --layers --cpu-seconds 0 > gpurun_out/bench5.json 2> gpurun_out/bench5.err; echo "bench exit $?" >> gpurun_out/bench5.err
cat gpurun_out/bench5.json
