#!/bin/bash
# forward-stream band heights on other frame shapes (diagnostic build): tests/experiments/ab_fs_shapes.sh "B S" band...
shape=$1; shift
set -- $shape "$@"; B=$1; S=$2; shift 2
for band in "$@"; do
  R2L_FS_BAND=$band R2L_LIB_PATH=tests/_build/libr2l_isp_hooks.so python bench.py --batch $B --size $S --steps 40 --warmup 8 --quick 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.readline())
k = o['kernels']
print('%4dx%4d^2 band %3s  ms/step %.4f ' % ($B, $S, '$band', o['ms_per_step']) + ' '.join('%s=%.1f' % (a.replace('r2l_launch_', '').replace('_kernel', ''), b['avg_us']) for a, b in sorted(k.items())))
"
done
