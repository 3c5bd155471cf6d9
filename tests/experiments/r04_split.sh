#!/bin/bash
# the statistics pass as one streaming kernel vs luma pass + statistics from the plane (R2L_FWD_STATS_SPLIT), diagnostic build
cd "$(dirname "$0")/../.."
H=$PWD/tests/_build/libr2l_isp_hooks.so
run() {
  env R2L_LIB_PATH=$H "$@" python bench.py --steps 40 --warmup 5 --quick 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-58s' % '$*', 'ms/step', d['ms_per_step'], ' '.join('%s=%.1f' % (k.replace('r2l_launch_','').replace('_kernel',''), v['avg_us']) for k,v in sorted(d['kernels'].items())))
"
}
run R2L_NOTHING=1
run R2L_FWD_STATS_SPLIT=1
run R2L_FWD_STATS_SPLIT=1 R2L_FST_BAND=12 R2L_FL_BAND=12
run R2L_FWD_STATS_SPLIT=1 R2L_FST_BAND=36 R2L_FL_BAND=48
run R2L_BWD_SPLIT_BLUR=1
run R2L_NOTHING=1
