#!/bin/bash
# band-height sweep of the plane passes under the diagnostic build (one pass of each setting; --quick bench line)
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
for b in 6 12 18 30 36; do run R2L_FA_BAND=$b R2L_BP_BAND=$b R2L_HB_BAND=$b R2L_B2S_BAND=$b; done
run R2L_NOTHING=1
for b in 16 22 32 44; do run R2L_FS_BAND=$b; done
