#!/bin/bash
# what the in-kernel final reductions (arrival tickets, two-level tree, unfold) cost at the end of their launches:
# the same step with the trees switched off (diagnostic build; results are then not produced -- timing only)
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
for i in 1 2; do
run R2L_NOTHING=1
run R2L_EXP_NO_TREE=1
run R2L_EXP_NO_TREE=2
run R2L_EXP_NO_TREE=4
run R2L_EXP_NO_TREE=7
done
