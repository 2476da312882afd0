cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 8 --warmup 2 --cpu-seconds 0 2>&1 | tail -3
