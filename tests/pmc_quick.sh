#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmcq; rm -rf $OUT; mkdir -p $OUT
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $OUT/sq -- python3 $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/sq.log 2>&1)
python3 - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmcq/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if k.startswith('r2l_launch_fwd') or k.startswith('r2l_launch_bwd'): agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg):
    print(k, {c: int(sum(v)/len(v)) for c,v in sorted(agg[k].items())})
PY
