#!/bin/bash
# Round-3 validation on the GPU box: the -m gpu suite (with the measured [scale]/[timed]/[strict]/[pipeline] lines),
# the default bench line (now with config.precision), and - unless SKIP_PROF - the rocprofv3 passes behind
# profiles/r03_* (kernel trace + SQ/GRBM + FETCH + WRITE, each --pmc pass on its own) condensed into the per-kernel
# roofline table and profiles/traffic.json (stamped with the kernel-source hash).
#   usage: scripts/gpu_r3_validate.sh [tag]        (outputs under gpurun_out/<tag>/)
#   env:   SKIP_TESTS=1, SKIP_PROF=1, TESTS="tests/test_strict_gpu.py ..." (default: the whole suite)
cd "$GRAFT_REPO_ROOT"
TAG=${1:-r3a}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then
  timeout 2400 python -m pytest ${TESTS:-tests} -m gpu -q -s -p no:cacheprovider > $O/pytest.log 2>&1
  echo "pytest rc=$?" | tee $O/pytest.rc
  grep -a "^\[scale\|^\[timed\|^\[strict\|^\[pipeline\|^\[overflow\| passed\| failed\|^FAILED\|^ERROR\|Error" $O/pytest.log | tail -90
fi
if [ -z "$SKIP_BENCH" ]; then
  timeout 900 python bench.py --layers --dump-launches $O/launches.json > $O/bench.json 2> $O/bench_layers.txt
  echo "bench rc=$?"; tail -c 3500 $O/bench.json | cut -c1-3500; tail -5 $O/bench_layers.txt
fi
if [ -z "$SKIP_PROF" ]; then
  R=$GRAFT_REPO_ROOT
  ARGS="--steps 10 --warmup 2 --cpu-seconds 0"
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_stats -o bench -- python $R/bench.py $ARGS > $R/$O/bench_traced.json 2> $R/$O/prof_stats.err)
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $R/$O/prof_sq -o bench -- python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 > /dev/null 2> $R/$O/prof_sq.err)
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$O/prof_fetch -o bench -- python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 > /dev/null 2> $R/$O/prof_fetch.err)
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$O/prof_write -o bench -- python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 > /dev/null 2> $R/$O/prof_write.err)
  python scripts/summarize_prof.py stats $O/prof_stats $O/kernel_stats.txt | head -24
  python scripts/summarize_prof.py table $O/launches.json $O/prof_stats $O/prof_sq $O/prof_fetch $O/prof_write $O/kernel_roofline.txt $O/traffic.json | cut -c1-200 | head -60
  tail -2 $O/prof_sq.err $O/prof_fetch.err
  find $O -name '*.csv' -size +4M -delete
  du -sh $O
fi
