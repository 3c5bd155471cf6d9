#!/bin/bash
# rocprofv3 kernel-trace durations (no HIP-event overhead) of the step at a given shape: tests/prof_small.sh 64 256
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_small_$1x$2
rm -rf $OUT; mkdir -p $OUT
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --batch $1 --size $2 --steps 50 --warmup 10 --quick --no-roofline ${EXTRA} > $OUT/stats.log 2>&1)
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if r['Name'].startswith('r2l_'):
        print('%-44s calls %5s  avg %8.2f us  min %8.2f  max %8.2f' % (r['Name'].split('(')[0], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
PY
t=$(find $OUT/stats -name "*kernel_trace.csv" | head -1)
python3 - "$t" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r['Kernel_Name'].startswith('r2l_')]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last 20 steps: per-kernel duration in launch order + gaps
names, seq = [], []
first = [i for i, r in enumerate(rows) if 'pack_fold' in r['Kernel_Name']]   # the step's first launch
for r in rows[first[-21]:first[-1]]:                                          # the last 20 whole steps
    seq.append((r['Kernel_Name'].split('(')[0].replace('r2l_launch_', '').replace('_kernel', ''), int(r['Start_Timestamp']), int(r['End_Timestamp'])))
import collections
dur, gap = collections.defaultdict(list), collections.defaultdict(list)
for i, (n, s, e) in enumerate(seq):
    key = n + ('#2' if (n.startswith('fwd') and i > 0 and seq[i - 1][0] == n) else '')
    dur[key].append((e - s) / 1e3)
    if i:
        gap[seq[i - 1][0] + '->' + n].append((s - seq[i - 1][2]) / 1e3)
for k, v in dur.items():
    print('  dur %-28s %7.2f us' % (k, sum(v) / len(v)))
for k, v in gap.items():
    print('  gap %-40s %7.2f us' % (k, sum(v) / len(v)))
period = (int(rows[first[-1]]['Start_Timestamp']) - int(rows[first[-21]]['Start_Timestamp'])) / 20 / 1e3
print('  step period (GPU timeline) %.1f us' % period)
PY
rm -rf $OUT/stats
