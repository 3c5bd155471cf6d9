#!/bin/bash
# the launches tails after the round-4 rewrite: tests/tail_timeline.py stations + the trees-off timing (R2L_EXP_NO_TREE)
cd "$(dirname "$0")/../.."
python tests/tail_timeline.py 2>&1 | grep -v amdgpu.ids
H=$PWD/tests/_build/libr2l_isp_hooks.so
run() {
  env R2L_LIB_PATH=$H "$@" python bench.py --steps 40 --warmup 10 --quick 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('%-28s' % ('$*'), 'ms/step %.4f ' % d['ms_per_step'], ' '.join('%s=%.1f' % (a.replace('r2l_launch_','').replace('_kernel',''), v['avg_us']) for a,v in sorted(k.items())))
"
}
run R2L_NOTHING=1
run R2L_EXP_NO_TREE=7
run R2L_NOTHING=1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-small-shapes 2>/dev/null | cut -c1-330
