#!/bin/bash
# A/B of device-library builds on the parametrized step: tests/experiments/ab_param.sh lib.so ...   (two interleaved rounds per shape)
cd "$(dirname "$0")/../.."
line() {
  python -c "
import sys, json
o = json.loads(sys.stdin.readline())
k = o['kernels']
print('%-22s %-12s ms/step %.4f  kernels %.1f us  ' % ('$1', '$2', o['ms_per_step'], sum(v['launches'] * v['avg_us'] for v in k.values()) / o['steps']) + ' '.join('%s=%.1f' % (a.replace('r2l_launch_', '').replace('_kernel', ''), b['avg_us']) for a, b in sorted(k.items())))
"
}
LIBS="$@"
for shape in "64 512" "64 256" "128 256"; do
  b=${shape% *}; s=${shape#* }
  for r in 1 2; do
    for lib in $LIBS; do
      R2L_LIB_PATH=$PWD/$lib python bench.py --steps 40 --warmup 5 --quick --batch $b --size $s 2>/dev/null | line $(basename $lib .so) ${b}x${s}
    done
  done
done
