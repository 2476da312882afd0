#!/bin/bash
# Round validation on the GPU box (the -m gpu suite, the default bench line, the rocprofv3 passes behind profiles/rNN_*):
#   gpurun -- 'bash scripts/gpu/validate.sh <tag>'          outputs under gpurun_out/<tag>/
#   env: SKIP_TESTS=1 SKIP_BENCH=1 SKIP_PROF=1, TESTS="tests/test_pair_gpu.py ..." (default: the whole suite),
#        BENCH_ARGS="--dtype fp16p" (extra bench.py flags for the bench + profile legs)
# Profile legs: kernel trace (+ --stats) and three SEPARATE --pmc passes (SQ + GRBM, FETCH_SIZE, WRITE_SIZE), condensed by
# scripts/summarize_prof.py into kernel_stats.txt, kernel_roofline.txt and traffic.json (stamped with the kernel-source hash).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-val}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then
  timeout 2400 python -m pytest ${TESTS:-tests} -m gpu -q -s -p no:cacheprovider > $O/pytest.log 2>&1
  echo "pytest rc=$?" | tee $O/pytest.rc
  grep -a "^\[scale\|^\[timed\|^\[strict\|^\[fp16p\|^\[pipeline\|^\[overflow\| passed\| failed\|^FAILED\|^ERROR\|Error" $O/pytest.log | tail -100
fi
if [ -z "$SKIP_BENCH" ]; then
  timeout 900 python bench.py $BENCH_ARGS --layers --dump-launches $O/launches.json > $O/bench.json 2> $O/bench_layers.txt
  echo "bench rc=$?"; tail -c 4500 $O/bench.json | cut -c1-4500; tail -5 $O/bench_layers.txt
fi
if [ -z "$SKIP_PROF" ]; then
  R=$PWD
  ARGS="$BENCH_ARGS --steps 10 --warmup 2 --cpu-seconds 0"
  SHORT="$BENCH_ARGS --steps 3 --warmup 1 --cpu-seconds 0"
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_stats -o bench -- python $R/bench.py $ARGS > $R/$O/bench_traced.json 2> $R/$O/prof_stats.err)
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $R/$O/prof_sq -o bench -- python $R/bench.py $SHORT > /dev/null 2> $R/$O/prof_sq.err)
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$O/prof_fetch -o bench -- python $R/bench.py $SHORT > /dev/null 2> $R/$O/prof_fetch.err)
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$O/prof_write -o bench -- python $R/bench.py $SHORT > /dev/null 2> $R/$O/prof_write.err)
  python scripts/summarize_prof.py stats $O/prof_stats $O/kernel_stats.txt | head -24
  python scripts/summarize_prof.py table $O/launches.json $O/prof_stats $O/prof_sq $O/prof_fetch $O/prof_write $O/kernel_roofline.txt $O/traffic.json | cut -c1-200 | head -60
  tail -n 2 $O/prof_sq.err $O/prof_fetch.err
  find $O -name '*.csv' -size +4M -delete
  du -sh $O
fi
