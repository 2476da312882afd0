set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
true
echo "pytest exit $?" >> gpurun_out/pytest4.log
tail -4 gpurun_out/pytest4.log
true
true
timeout 900 python -m pytest tests/test_ranking_gpu.py tests/test_pipeline_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest4b.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest4b.log; tail -5 gpurun_out/pytest4b.log
