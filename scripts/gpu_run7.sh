set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python scripts/bench_rank.py > gpurun_out/bench_rank.json 2> gpurun_out/bench_rank.err; cat gpurun_out/bench_rank.json
timeout 600 python -m pytest tests -m gpu -q -k "gemm or postproc or similarity or ranking or eval_model" -p no:cacheprovider 2>&1 | tail -3
