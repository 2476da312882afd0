cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for b in 1 2 4; do timeout 300 python bench.py --batch $b --profile-every 1000 --cpu-seconds 0 --steps 50 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B=$b', d['value'], d['ms_per_step'])"; done
