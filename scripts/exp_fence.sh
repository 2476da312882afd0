#!/bin/bash
# Experiment build: the library with ring_barrier() reduced to a raw s_barrier (-DDIR_EXP_NO_RING_FENCE), to price the
# scheduler fences (scripts/_exp/libdir_nofence.so, used through DIRTORCH_AMD_LIB).  NOT a correct build under overlap.
set -e
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
C="$HERE/../deep-image-retrieval_amd/csrc"
O="$HERE/_exp/nofence"
mkdir -p "$O"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-but-set-variable -DDIR_EXP_NO_RING_FENCE"
pids=()
for src in conv_f32 conv_igemm conv_patch conv_patchw conv_persist conv_ring conv_wreg conv_c3c1 stem_pool pointwise resize gemm_f32 sim_split ranking comm engine c_api; do
  /opt/rocm/bin/hipcc $F -c "$C/$src.hip" -o "$O/$src.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -ldl -o "$HERE/_exp/libdir_nofence.so" "$O"/*.o
rm -rf "$O"
echo built "$HERE/_exp/libdir_nofence.so"
