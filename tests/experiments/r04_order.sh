#!/bin/bash
# does the memory-side cache carry dL/dout from bn_reduce to B1's plane pass?  bn_reduce walking the tensors from their end
# (R2L_BNR_ORDER=1, diagnostic build) leaves their START most recently used
cd "$(dirname "$0")/../.."
H=$PWD/tests/_build/libr2l_isp_hooks.so
run() {
  env R2L_LIB_PATH=$H "$@" python bench.py --steps 40 --warmup 10 --quick 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('%-28s' % ('$*'), 'ms/step %.4f ' % d['ms_per_step'], ' '.join('%s=%.1f' % (a.replace('r2l_launch_','').replace('_kernel',''), v['avg_us']) for a,v in sorted(k.items())))
"
}
for i in 1 2 3; do
run R2L_NOTHING=1
run R2L_BNR_ORDER=1
done
