#!/bin/bash
# round 6: kernel-trace durations of the small-map variants on one shape (the python-side timing of exp_small_time.py is host-bound)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r6smallprof}; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
for shp in b1.l3.conv2 b1.l3.conv1 b1.l3.conv3; do
(cd /tmp && EXP_SHAPES=$shp timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$shp -o t -- python $R/scripts/exp_small_time.py 64x64_small_s4k2 64x64_small_s8 64x64_small_s4 64x64_w2x2_s4 64x64_w2x2_s8 > $R/$O/$shp.txt 2> $R/$O/$shp.err)
python scripts/summarize_prof.py stats $O/prof_$shp $O/stats_$shp.txt | grep -v "torch:\|rocclr\|^#" | head -14
done
