cd "${GRAFT_REPO_ROOT:-/root/repo}"
for env in "" "DIRTORCH_AMD_STEM_V1=1"; do
  echo "== $env"
  env $env timeout 300 python bench.py --dtype fp16p --cpu-seconds 0 --steps 16 --profile-every 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['all_kernels_ms_per_step'], d['value'])"
  env $env timeout 300 python bench.py --dtype fp16p --cpu-seconds 0 --steps 16 --profile-every 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['all_kernels_ms_per_step'], d['value'])"
done
