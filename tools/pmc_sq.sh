#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/a -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/b -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/c -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/c.log 2>&1
cd $REPO
python3 - <<'PY'
import csv,glob,collections
for d in 'abc':
    for f in glob.glob('gpurun_out/pmc2/%s/**/*counter_collection.csv'%d, recursive=True):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'fx_search_fast' in r['Kernel_Name'] and 'true, false' not in r['Kernel_Name'][:60]:
                pass
            if r['Kernel_Name'].startswith('void fx_search_fast<16, true, false'):
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in acc.items(): print(d,k,sum(v)/len(v),len(v))
PY
