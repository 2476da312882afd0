cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -p no:cacheprovider -k "small_s4k2" 2>&1 | tail -2
bash scripts/gpu/r6_smallprof.sh r6smallprof2
