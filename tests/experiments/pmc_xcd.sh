#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, their own passes) of the step's kernels under XCD windows of the band passes:
# tests/experiments/pmc_xcd.sh <M> [tag]   (diagnostic build; M = 0: off)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
ROOT=$PWD
M=${1:-0}
OUT=$ROOT/gpurun_out/${2:-pmc_xcd_$M}
rm -rf $OUT; mkdir -p $OUT
export R2L_LIB_PATH=${LIB:-$ROOT/tests/_build/libr2l_isp_hooks.so}
if [ "$M" != "0" ]; then export R2L_XCD_FS=${XFS:-$M} R2L_XCD_FA=${XFA:-$M} R2L_XCD_BP=${XBP:-$M} R2L_XCD_HB=${XHB:-$M} R2L_XCD_B2S=${XB2S:-$M}; fi
run() { n=$1; shift; (cd /tmp && R2L_BENCH_PREROLL_S=0 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- python3 $ROOT/bench.py --steps 3 --warmup 1 --quick --no-roofline > $OUT/$n.log 2>&1); }
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE
python3 - "$OUT" "$M" <<'PY'
import csv, glob, collections, json, sys
out, M = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if k.startswith('r2l_'):
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
tot = 0.0
lines = []
for k, c in sorted(agg.items()):
    if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
        fb = 2.0 * 1024.0 * sum(c['FETCH_SIZE']) / len(c['FETCH_SIZE'])   # (x2: gfx950 correction, MI355X_MICROARCH.md)
        wb = 1024.0 * sum(c['WRITE_SIZE']) / len(c['WRITE_SIZE'])
        tot += fb + wb
        lines.append('%-46s fetch %8.1f MB  write %8.1f MB  total %8.1f MB' % (k, fb / 1e6, wb / 1e6, (fb + wb) / 1e6))
print('XCD windows M = %s' % M)
print('\n'.join(lines))
print('step total %.1f MB' % (tot / 1e6))
PY
rm -rf $OUT/tcc1 $OUT/tcc2
