set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python bench.py > gpurun_out/bench10.json 2> gpurun_out/bench10.err; cat gpurun_out/bench10.json
timeout 600 python bench.py --profile-every 1 --cpu-seconds 0 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('every1', d['value'])"
timeout 600 python bench.py --profile-every 1000 --cpu-seconds 0 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('every1000', d['value'], d['ms_per_step'])"
