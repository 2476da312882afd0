#!/bin/bash
# round 4, call 1: the paired-fp16 head (DIR_FP16P) - op tests, the literal north-star gate, rates per kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4a; mkdir -p $O
timeout 1200 python -m pytest tests/test_pair_gpu.py -q -s --maxfail=8 > $O/pair_tests.log 2>&1; echo "pair tests rc=$?"
tail -n 40 $O/pair_tests.log
timeout 400 python -m pytest tests/test_pipeline_gpu.py -q -s -k "extract_whiten_rank_map_parity and fp16p" > $O/pipe.log 2>&1; echo "pipeline rc=$?"
grep -E "pipeline\]|passed|failed|Error" $O/pipe.log | tail -n 8
timeout 300 python bench.py --dtype fp16p --layers --cpu-seconds 0 --steps 8 > $O/bench_fp16p.json 2> $O/bench_fp16p_layers.txt; echo "bench fp16p rc=$?"
head -c 600 $O/bench_fp16p.json; echo
grep -E "conv_pair|stem_pool|prep_input" $O/bench_fp16p_layers.txt
DIRTORCH_AMD_PAIR_STAGES=2 timeout 300 python bench.py --dtype fp16p --cpu-seconds 0 --steps 8 > $O/bench_fp16p_s2.json 2>/dev/null; head -c 300 $O/bench_fp16p_s2.json; echo
timeout 300 python bench.py --dtype fp16 --layers --cpu-seconds 0 --steps 8 > $O/bench_fp16.json 2> $O/bench_fp16_layers.txt; head -c 300 $O/bench_fp16.json; echo
