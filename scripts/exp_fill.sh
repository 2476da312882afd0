#!/bin/bash
# Experiment build: the same library with the MFMA phase of conv_igemm compiled out, to measure what
# the LDS-DMA fill ring alone sustains per layer shape (results are NOT valid convolutions).
set -e
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
C="$HERE/../deep-image-retrieval_amd/csrc"
O="$HERE/_exp"
mkdir -p "$O"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-but-set-variable"
for mode in FILL_ONLY NO_FILL; do
  /opt/rocm/bin/hipcc $F -DDIR_EXP_$mode -c "$C/conv_igemm.hip" -o "$O/conv_igemm_$mode.o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$O/libdir_$mode.so" "$O/conv_igemm_$mode.o" \
    "$C/_build"/conv_patch.o "$C/_build"/conv_persist.o "$C/_build"/stem_pool.o "$C/_build"/pointwise.o \
    "$C/_build"/resize.o "$C/_build"/gemm_f32.o "$C/_build"/ranking.o "$C/_build"/engine.o "$C/_build"/c_api.o
  echo built "$O/libdir_$mode.so"
done
