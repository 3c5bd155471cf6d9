#!/bin/bash
# dynamic instruction mix of the static short chain (counters in their own run, kernel-trace only)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/pmc_static
rm -rf $OUT; mkdir -p $OUT
run() { n=$1; shift; (cd /tmp && rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- python3 $ROOT/bench.py --workload static --steps 3 --warmup 1 --no-cpu-baseline --no-roofline ${DEB:+--debayer $DEB} > $OUT/$n.log 2>&1); }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if k.startswith('r2l_'):
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg):
    print(k)
    for c, v in sorted(agg[k].items()):
        print('   %-28s %14.0f  (n=%d)' % (c, sum(v) / len(v), len(v)))
PY
