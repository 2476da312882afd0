#!/bin/bash
# Round 6: instruction mix / stall counters of stem_pool_u8_kernel (what bounds it once the matrix work is 2 MFMAs per term)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r6stempmc}; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE -d $R/$O/pmc1 -o stem -- python $R/scripts/exp_stem_u8_time.py > $R/$O/pmc1.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU -d $R/$O/pmc2 -o stem -- python $R/scripts/exp_stem_u8_time.py > $R/$O/pmc2.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS -d $R/$O/pmc3 -o stem -- python $R/scripts/exp_stem_u8_time.py > $R/$O/pmc3.log 2>&1)
python - <<P
import sqlite3, glob
for d in ('pmc1', 'pmc2', 'pmc3'):
    for f in glob.glob('$O/%s/**/*.db' % d, recursive=True):
        con = sqlite3.connect(f)
        try:
            rows = con.execute("select k.name, p.counter_name, sum(p.counter_value), count(distinct p.dispatch_id) from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id where k.name like '%stem_pool_u8%' group by k.name, p.counter_name").fetchall()
        except Exception as e:
            print(d, 'ERR', e); continue
        for name, cn, v, n in rows:
            print(d, cn, '%.4g per launch over %d launches' % (v / n, n))
P
tail -2 $O/pmc1.log
