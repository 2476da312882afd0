cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6nt; mkdir -p $O
for i in 1 2 3; do
  for m in nt plain; do
    if [ $m = nt ]; then export DIRTORCH_AMD_LIB=$PWD/scripts/_exp/lib_conv_patchs2_1.so; else unset DIRTORCH_AMD_LIB; fi
    timeout 600 python bench.py --steps 30 --warmup 3 --cpu-seconds 0 --no-precision --layers > $O/bench_${m}_$i.json 2> $O/layers_${m}_$i.txt
  done
done
unset DIRTORCH_AMD_LIB
grep -h "layer2.0.conv2\|layer2.0.ds" $O/layers_nt_1.txt $O/layers_plain_1.txt $O/layers_nt_2.txt $O/layers_plain_2.txt
python - <<P
import json
for m in ('nt','plain'):
    print(m, [json.loads(open('$O/bench_%s_%d.json'%(m,i)).read().strip().splitlines()[-1])['value'] for i in (1,2,3)])
P
