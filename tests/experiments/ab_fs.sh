#!/bin/bash
# forward-stream experiments: tests/experiments/ab_fs.sh name[:band] ...
for spec in "$@"; do
  n=${spec%%:*}; band=${spec#*:}; [ "$band" = "$spec" ] && band=0
  R2L_FS_BAND=$band R2L_LIB_PATH=tests/_build/ab/$n.so python bench.py --steps 30 --warmup 5 --quick 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.readline())
k = o['kernels']
print('%-10s band %3s value %8.1f ms/step %.4f ' % ('$n', '$band', o['value'], o['ms_per_step']) + ' '.join('%s=%.1f' % (a.replace('r2l_launch_', '').replace('_kernel', ''), b['avg_us']) for a, b in sorted(k.items())))
"
done
