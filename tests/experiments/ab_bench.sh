#!/bin/bash
# A/B of device-library builds on the headline bench: tests/experiments/ab_bench.sh [rounds] name ... (tests/_build/ab/<name>.so)
rounds=$1; shift
for r in $(seq $rounds); do
  for n in "$@"; do
    R2L_LIB_PATH=tests/_build/ab/$n.so python bench.py --steps 30 --warmup 5 --quick 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.readline())
k = o['kernels']
print('%-10s value %8.1f ms/step %.4f ' % ('$n', o['value'], o['ms_per_step']) + ' '.join('%s=%.1f' % (a.replace('r2l_launch_', '').replace('_kernel', ''), b['avg_us']) for a, b in sorted(k.items())))
"
  done
done
