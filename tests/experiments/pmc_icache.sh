#!/bin/bash
# instruction-cache counters of the step's kernels (and the static chains with --static): their own PMC pass, kernel-trace only
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/${1:-pmc_icache}
rm -rf $OUT; mkdir -p $OUT
run() { n=$1; shift; (cd /tmp && R2L_BENCH_PREROLL_S=0 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- python3 $ROOT/bench.py --steps 3 --warmup 1 --quick --no-roofline > $OUT/$n.log 2>&1); }
run ic1 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run ic2 SQC_TC_INST_REQ SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAIT_ANY
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if k.startswith('r2l_'):
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
with open(out + '/pmc_icache.txt', 'w') as fh:
    for k in sorted(agg):
        fh.write(k + '\n')
        for c, v in sorted(agg[k].items()):
            fh.write('   %-28s %14.0f  (n=%d)\n' % (c, sum(v) / len(v), len(v)))
print(open(out + '/pmc_icache.txt').read())
PY
tail -2 $OUT/ic1.log $OUT/ic2.log
rm -rf $OUT/ic1 $OUT/ic2
