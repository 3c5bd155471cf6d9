#!/bin/bash
# A/B of kernel B2: streaming (default) vs tile (R2L_BWD2_TILED=1, diagnostic build) on the headline bench
for r in 1 2; do
for v in 0 1; do
  R2L_BWD2_TILED=$v R2L_LIB_PATH=tests/_build/libr2l_isp_hooks.so python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-static-c3 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.readline())
k = o['kernels']
print('tiled=$v value %8.1f ms/step %.4f ' % (o['value'], o['ms_per_step']) + ' '.join('%s=%.1f' % (a.replace('r2l_launch_', '').replace('_kernel', ''), b['avg_us']) for a, b in sorted(k.items())))
"
done
done
