#!/bin/bash
# what the driver runs at round end, plus the torchrun launch form: full GPU suite, smoke(), default bench
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/pytest_final.log 2>&1
echo "exit $?" >> gpurun_out/pytest_final.log
grep -v "Warning\|pin_memory\|^$" gpurun_out/pytest_final.log | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -1 gpurun_out/bench_final.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 8 --warmup 2 --cpu-seconds 0 2>&1 | tail -1 | cut -c1-160
