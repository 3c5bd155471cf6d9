#!/bin/bash
# kernel timeline of the headline bench: where does the GPU idle between the step's launches?
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/gaps
rm -rf $OUT; mkdir -p $OUT
(cd /tmp && R2L_LIB_PATH=$ROOT/tests/_build/libr2l_isp_hooks.so rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-static-c3 > $OUT/trace.log 2>&1)
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + '/trace/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
rows = [r for r in rows]
names = [r['Kernel_Name'].split('(')[0].replace('r2l_launch_', '').replace('_kernel', '') for r in rows]
st = [int(r['Start_Timestamp']) for r in rows]
en = [int(r['End_Timestamp']) for r in rows]
# steady state: the last 30 steps = everything after the 10th-last pack_fold ... use last 30 occurrences
idx = [i for i, n in enumerate(names) if n.startswith('pack_fold')]
lo, hi = idx[-31], idx[-1]
gaps = collections.defaultdict(list)
dur = collections.defaultdict(list)
for i in range(lo, hi):
    dur[names[i]].append(en[i] - st[i])
    gaps[names[i] + ' -> ' + names[i + 1]].append(st[i + 1] - en[i])
with open(out + '/gaps.txt', 'w') as fh:
    tot = (st[hi] - st[lo]) / 30
    fh.write('step period %.1f us\n' % (tot / 1e3))
    for k, v in dur.items():
        fh.write('kernel %-40s n=%d avg %.1f us\n' % (k, len(v), sum(v) / len(v) / 1e3))
    for k, v in gaps.items():
        fh.write('gap    %-60s n=%d avg %.1f us  min %.1f max %.1f\n' % (k, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3))
print(open(out + '/gaps.txt').read())
PY
