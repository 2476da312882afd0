#!/bin/bash
# Experiment builds of sim_split.hip with phases compiled out (timing only; see DIR_SIM_ABL in the source):
# scripts/_exp/libdir_sim<bits>.so, used through DIRTORCH_AMD_LIB.
set -e
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
C="$HERE/../deep-image-retrieval_amd/csrc"
O="$HERE/_exp"
mkdir -p "$O"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-but-set-variable"
OBJS=$(ls "$C/_build"/*.o | grep -v sim_split.o)
for bits in ${@:-1 2 4}; do
  /opt/rocm/bin/hipcc $F -DDIR_SIM_ABL=$bits -c "$C/sim_split.hip" -o "$O/sim_split_$bits.o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -ldl -o "$O/libdir_sim$bits.so" "$O/sim_split_$bits.o" $OBJS
  rm -f "$O/sim_split_$bits.o"
  echo built "$O/libdir_sim$bits.so"
done
