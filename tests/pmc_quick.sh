#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmcq; rm -rf $OUT; mkdir -p $OUT
run() { n=$1; shift; (cd /tmp && rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- python3 $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/$n.log 2>&1); }
run a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_BUSY_CYCLES SQ_WAVE_CYCLES
run b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS
run c FETCH_SIZE
run d WRITE_SIZE
python3 - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmcq/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if k.startswith('r2l_launch_fwd') or k.startswith('r2l_launch_bwd') or 'bn_reduce' in k: agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg):
    print(k)
    for c,v in sorted(agg[k].items()): print('   %-24s %14.0f' % (c, sum(v)/len(v)))
PY
