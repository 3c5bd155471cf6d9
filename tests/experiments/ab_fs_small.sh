#!/bin/bash
# row-streaming forward on the dataset's tile size (256 x 256, dataset.py:92): band height sweep (diagnostic build)
cd "$(dirname "$0")/../.."
export R2L_LIB_PATH=$PWD/tests/_build/libr2l_isp_hooks.so
for shape in 64x256 128x256; do
for band in 16 12 10 8 6 4; do
  R2L_FS_MINBAND=$band python bench.py --batch ${shape%x*} --size ${shape#*x} --steps 50 --warmup 10 --quick 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.readline()); k = o['kernels']
print('$shape band=$band', 'ms/step %.4f' % o['ms_per_step'], ' '.join('%s=%.1f' % (n.replace('r2l_launch_', '').replace('_kernel', ''), v['avg_us']) for n, v in sorted(k.items()) if 'fwd' in n))
"
done
done
