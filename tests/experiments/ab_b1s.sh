#!/bin/bash
# kernel B1 with the kept luma plane: variants (tests/_build/ab/<name>.so, built by tests/build_ab.sh with -DR2L_TEST_HOOKS)
cd "$(dirname "$0")/../.."
run() { python bench.py --steps 30 --warmup 5 --quick 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.readline()); k = o['kernels']
print('%-40s ms/step %.4f ' % ('$1', o['ms_per_step']) + ' '.join('%s=%.1f' % (a.replace('r2l_launch_', '').replace('_kernel', ''), b['avg_us']) for a, b in sorted(k.items())))
"; }
for r in 1 2; do
for n in "$@"; do
  export R2L_LIB_PATH=tests/_build/ab/$n.so
  run "$n"
done
R2L_BWD1_RECOMPUTE=1 run "$1 recomputing"
done
