#!/bin/bash
# SQ counters of the parametrized step's kernels (counters in their own runs, kernel-trace only): tests/pmc_param.sh <tag>
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/${1:-pmc_param}
rm -rf $OUT; mkdir -p $OUT
run() { n=$1; shift; (cd /tmp && R2L_BENCH_PREROLL_S=0 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- python3 $ROOT/bench.py --steps 3 --warmup 1 --quick --no-roofline > $OUT/$n.log 2>&1); }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq3 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM
run sq4 SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if k.startswith('r2l_'):
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
with open(out + '/pmc_summary.txt', 'w') as fh:
    for k in sorted(agg):
        fh.write(k + '\n')
        for c, v in sorted(agg[k].items()):
            fh.write('   %-28s %14.0f  (n=%d)\n' % (c, sum(v) / len(v), len(v)))
print(open(out + '/pmc_summary.txt').read())
PY
tail -3 $OUT/sq3.log $OUT/sq4.log
rm -rf $OUT/sq1 $OUT/sq2 $OUT/sq3 $OUT/sq4
