#!/bin/bash
# bn_reduce grid sweep (diagnostic build honours R2L_GRID_BNR)
export R2L_LIB_PATH=tests/_build/libr2l_isp_hooks.so
for g in 256 512 768 1024 1536 2048; do
  R2L_GRID_BNR=$g python bench.py --steps 30 --warmup 5 --quick 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.readline()); k = o['kernels']
print('grid %5d  ms/step %.4f  bn_reduce %.1f us' % ($g, o['ms_per_step'], k['r2l_launch_bn_reduce_kernel']['avg_us']))
"
done
