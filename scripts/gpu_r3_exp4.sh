#!/bin/bash
# Round-3 experiment 4: where does the ring kernel's time go?  Phases compiled out (scripts/exp_ring.sh builds).
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3e
mkdir -p $O
export EXP_SHAPES=l3.conv1,l4.conv1
V="128x256_ring1x1 256x256_persist1x1 256x256_persist1x1_x3"
python scripts/exp_conv_time.py $V 2>&1 | grep -v "^lib" | sed 's/^/full        /' | tee $O/abl.txt
for bits in 1 2 3 4 8 7 11; do
  DIRTORCH_AMD_LIB=scripts/_exp/libdir_ring$bits.so python scripts/exp_conv_time.py 128x256_ring1x1 2>&1 | grep -v "^lib" | sed "s/^/abl $bits       /" | tee -a $O/abl.txt
done
