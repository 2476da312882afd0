cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; O=gpurun_out/r4simpmc; mkdir -p $O; export TMPDIR=/tmp
A="--workload distractors --steps 3 --warmup 1 --cpu-seconds 0"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$O/fetch -o b -- python $R/bench.py $A > /dev/null 2> $R/$O/fetch.err)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$O/write -o b -- python $R/bench.py $A > /dev/null 2> $R/$O/write.err)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/$O/sq -o b -- python $R/bench.py $A > /dev/null 2> $R/$O/sq.err)
python scripts/pmc_table.py $O/fetch $O/write $O/sq 2>&1 | grep -E "kernel|sim_split|rank_|revisitop|split_queries" | cut -c1-250
