#!/bin/bash
# the step at the reference's tile size under the diagnostic build: tile kernels vs plane passes of the backward, band heights
cd "$(dirname "$0")/../.."
H=$PWD/tests/_build/libr2l_isp_hooks.so
run() {
  b=$1; shift
  env R2L_LIB_PATH=$H "$@" python bench.py --steps 60 --warmup 10 --quick --batch $b --size 256 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('%-4s %-52s' % ('$b', '$*'), 'kernels %.1f us ' % (sum(v['launches']*v['avg_us'] for v in k.values())/d['steps']), ' '.join('%s=%.1f' % (a.replace('r2l_launch_','').replace('_kernel',''), v['avg_us']) for a,v in sorted(k.items())))
"
}
for b in 64 128; do
run $b R2L_NOTHING=1
run $b R2L_BWD_PLANES=1
run $b R2L_BWD_PLANES=1 R2L_BP_BAND=6 R2L_HB_BAND=6 R2L_B2S_BAND=6
run $b R2L_BWD_PLANES=1 R2L_BP_BAND=12 R2L_HB_BAND=12 R2L_B2S_BAND=12
run $b R2L_FWD_STATS_STREAM=1
run $b R2L_FA_BAND=12 R2L_FST_BAND=12 R2L_FL_BAND=12
done
