#!/bin/bash
# round 5: (1) the layer3 seam (conv_seam3.hip, experiments build; with profiles/r05_seam3_storers.patch applied: its storer-wave form): op-level parity, standalone timing against
# the two launches it replaces, the whole step with it; (2) the Winograd pricing builds of conv_patchw.hip (scripts/exp_abl.sh
# conv_patchw DIR_PATCHW_ABL 8 24 56) on the 3x3 shapes.      gpurun -- 'bash scripts/gpu/r5_seam3.sh <tag>'
TAG=${1:-r5seam3}
O=gpurun_out/$TAG
mkdir -p $O
EXP=$PWD/deep-image-retrieval_amd/dirtorch_amd/libdir_engine_exp.so
export DIRTORCH_AMD_EXPERIMENTS=1
DIRTORCH_AMD_LIB=$EXP timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -p no:cacheprovider -k "layer3_seam or ring_kernel" > $O/pytest_seam.log 2>&1
tail -3 $O/pytest_seam.log
DIRTORCH_AMD_LIB=$EXP timeout 200 python scripts/exp_seam3_time.py > $O/seam3_time.txt 2>&1
python scripts/exp_seam3_time.py >> $O/seam3_time.txt 2>&1      # (default library: the two launches, no fused form)
cat $O/seam3_time.txt | grep -v amdgpu.ids
for rep in 1 2; do
  DIRTORCH_AMD_LIB=$EXP timeout 300 python bench.py --steps 30 --warmup 3 --profile-every 3 --cpu-seconds 0 --layers > $O/bench_seam3_$rep.json 2> $O/layers_seam3_$rep.txt
  timeout 300 env -u DIRTORCH_AMD_EXPERIMENTS python bench.py --steps 30 --warmup 3 --profile-every 3 --cpu-seconds 0 --layers > $O/bench_base_$rep.json 2> $O/layers_base_$rep.txt
done
for f in $O/bench_*.json; do echo $f $(python -c "import json;d=json.loads(open('$f').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])"); done
grep -E "layer3\.(5|6)\." $O/layers_seam3_1.txt
unset DIRTORCH_AMD_EXPERIMENTS
# ---- Winograd pricing -------------------------------------------------------------------------------------------------
for b in "" 8 24 56; do
  if [ -z "$b" ]; then L=""; else L=$PWD/scripts/_exp/lib_conv_patchw_$b.so; fi
  echo "== DIR_PATCHW_ABL=${b:-0}" >> $O/winograd_abl.txt
  DIRTORCH_AMD_LIB=$L EXP_SHAPES=l3.conv2,l2.conv2,l4.conv2 timeout 200 python scripts/exp_conv_time.py 512x128_patch3x3w 2>&1 | grep -v "amdgpu.ids" >> $O/winograd_abl.txt
done
cat $O/winograd_abl.txt
