#!/bin/bash
# streaming forward: band height sweep (diagnostic build honours R2L_FS_BAND; 0 = the default sizing)
export R2L_LIB_PATH=tests/_build/libr2l_isp_hooks.so
for b in 0 12 16 22 32 44 64 128; do
  R2L_FS_BAND=$b python bench.py --steps 30 --warmup 5 --quick 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.readline()); k = o['kernels']
print('band %4d  ms/step %.4f  fwd %.1f us' % ($b, o['ms_per_step'], k['r2l_launch_fwd_stream_w2_kernel']['avg_us']))
"
done
