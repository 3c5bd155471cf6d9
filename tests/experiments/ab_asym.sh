#!/bin/bash
# kernel B2: uneven tile shares of the two workgroups of a CU (R2L_B2_ASYM = period of the half rounds; 0 / 1 = even),
# swept on the diagnostic build (tests/_build/libr2l_isp_hooks.so honours the environment)
cd "$(dirname "$0")/../.."
export R2L_LIB_PATH=$PWD/tests/_build/libr2l_isp_hooks.so
for rep in 1 2; do
for a in 0 2 3 4 5 6 8; do
  R2L_B2_ASYM=$a python bench.py --steps 50 --warmup 10 --quick ${EXTRA} 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.readline()); k = o['kernels']
print('asym=$a', 'ms/step %.4f' % o['ms_per_step'], ' '.join('%s=%.1f' % (n.replace('r2l_launch_', '').replace('_kernel', ''), v['avg_us']) for n, v in sorted(k.items())))
"
done
done
