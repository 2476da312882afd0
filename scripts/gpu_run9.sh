set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_pipeline_gpu.py tests/test_ranking_gpu.py tests/test_model_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest9.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest9.log; tail -4 gpurun_out/pytest9.log
bash scripts/gpu_prof.sh > gpurun_out/prof9.log 2>&1; tail -3 gpurun_out/prof9.log
