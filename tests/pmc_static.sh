#!/bin/bash
# dynamic instruction mix of the static short chain (counters in their own run, kernel-trace only)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/${OUTNAME:-pmc_static}
rm -rf $OUT; mkdir -p $OUT
run() { n=$1; shift; (cd /tmp && rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- python3 $ROOT/bench.py --workload static --steps 3 --warmup 1 --no-cpu-baseline --no-roofline ${DEB:+--debayer $DEB} $EXTRA > $OUT/$n.log 2>&1); }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if k.startswith('r2l_'):
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
import json, os
for k in sorted(agg):
    print(k)
    for c, v in sorted(agg[k].items()):
        print('   %-28s %14.0f  (n=%d)' % (c, sum(v) / len(v), len(v)))
# HBM traffic per launch (256x1024x1024): FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE x2 on gfx950
# (MI355X_MICROARCH.md, HBM / rocprofv3 section)
traffic = {}
for k, c in agg.items():
    if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
        fb = 2.0 * 1024.0 * sum(c['FETCH_SIZE']) / len(c['FETCH_SIZE'])
        wb = 1024.0 * sum(c['WRITE_SIZE']) / len(c['WRITE_SIZE'])
        traffic[k] = {'fetch_bytes': fb, 'write_bytes': wb, 'total_bytes': fb + wb,
                      'note': 'FETCH_SIZE x2 (gfx950 correction), WRITE_SIZE as read; KiB -> bytes; 256x1024x1024 static short chain'}
json.dump(traffic, open(out + '/pmc_traffic_static.json', 'w'), indent=1)
print(json.dumps(traffic, indent=1))
PY
