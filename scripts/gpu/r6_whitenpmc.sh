#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r6whitenpmc}; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
python scripts/exp_whiten_time.py 2>&1 | grep -v amdgpu.ids | tee $O/whiten_time.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$O/fetch -o w -- python $R/scripts/exp_whiten_time.py > $R/$O/fetch.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$O/write -o w -- python $R/scripts/exp_whiten_time.py > $R/$O/write.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/$O/sq -o w -- python $R/scripts/exp_whiten_time.py > $R/$O/sq.log 2>&1)
python - <<P
import sqlite3, glob
for d in ('fetch', 'write', 'sq'):
    for f in glob.glob('$O/%s/**/*.db' % d, recursive=True):
        con = sqlite3.connect(f)
        rows = con.execute("select k.name, p.counter_name, sum(p.counter_value), count(distinct p.dispatch_id) from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id where k.name like '%whiten_split%' group by k.name, p.counter_name").fetchall()
        for name, cn, v, n in rows:
            print(d, cn, '%.5g per launch over %d launches' % (v / n, n), '(FETCH/WRITE_SIZE in KiB; FETCH x2 per the gfx950 note)' if 'SIZE' in cn else '')
P
