#!/bin/bash
# rocprofv3 PMC passes over a short bench run (counters in their own runs, kernel-trace only)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc
mkdir -p $OUT
run() { # name counters...
  n=$1; shift
  (cd /tmp && rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- python3 $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/$n.log 2>&1)
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq3 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python3 - <<'PY'
import csv,glob,collections,os
out=os.environ.get('OUT','gpurun_out/pmc')
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if not k.startswith('r2l_'): continue
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg):
    print(k)
    for c,v in sorted(agg[k].items()): print('   %-28s %14.0f  (n=%d)' % (c, sum(v)/len(v), len(v)))
PY
