# rocprofv3 evidence for bench.py (run from the repo root on the GPU box)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
# default bench command: tiles from the built-in heuristic (no autotune, no tuning cache)
unset DIRTORCH_AMD_TUNE_CACHE
ARGS="--steps 10 --warmup 2 --cpu-seconds 0"
(cd $R && timeout 300 python bench.py $ARGS > gpurun_out/prof_bench_plain.json 2> gpurun_out/prof_bench_plain.err)
(cd $R && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o bench -- python bench.py $ARGS > gpurun_out/prof_bench_traced.json 2> gpurun_out/prof_stats.err)
(cd $R && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch -o bench -- python bench.py --steps 3 --warmup 1 --cpu-seconds 0 > /dev/null 2> gpurun_out/prof_fetch.err)
(cd $R && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/prof_write -o bench -- python bench.py --steps 3 --warmup 1 --cpu-seconds 0 > /dev/null 2> gpurun_out/prof_write.err)
cd $R
ls -la gpurun_out/prof_stats gpurun_out/prof_fetch | head -30
python scripts/summarize_prof.py stats gpurun_out/prof_stats gpurun_out/prof_stats_summary.txt | head -30
python scripts/summarize_prof.py pmc gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_pmc.json gpurun_out/traffic.json
# keep the merge-back small: drop the raw multi-MB traces, keep the summaries
du -sh gpurun_out/prof_*
find gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write -name '*.csv' -size +8M -delete
cat gpurun_out/prof_bench_plain.json gpurun_out/prof_bench_traced.json
