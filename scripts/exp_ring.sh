#!/bin/bash
# Experiment builds of conv_ring.hip with phases compiled out (timing only; see DIR_RING_ABL in the source):
# scripts/_exp/libdir_ring<bits>.so, used through DIRTORCH_AMD_LIB by scripts/exp_conv_time.py.
set -e
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
C="$HERE/../deep-image-retrieval_amd/csrc"
O="$HERE/_exp"
mkdir -p "$O"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-but-set-variable"
OBJS=$(ls "$C/_build"/*.o | grep -v conv_ring.o)
for bits in ${@:-1 2 3 4 8 7 11}; do
  /opt/rocm/bin/hipcc $F -DDIR_RING_ABL=$bits -c "$C/conv_ring.hip" -o "$O/conv_ring_$bits.o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -ldl -o "$O/libdir_ring$bits.so" "$O/conv_ring_$bits.o" $OBJS
  echo built "$O/libdir_ring$bits.so"
done
