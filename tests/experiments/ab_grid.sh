#!/bin/bash
# workgroups-per-CU experiments on the tile kernels (diagnostic build honours R2L_GRID_*)
export R2L_LIB_PATH=tests/_build/libr2l_isp_hooks.so
run() {
  python bench.py --steps 30 --warmup 5 --quick 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.readline())
k = o['kernels']
print('%-34s ms/step %.4f ' % ('$1', o['ms_per_step']) + ' '.join('%s=%.1f' % (a.replace('r2l_launch_', '').replace('_kernel', ''), b['avg_us']) for a, b in sorted(k.items())))
"
}
run default
R2L_GRID_BWD1=512 run bwd1=512
R2L_GRID_BWD1=384 run bwd1=384
R2L_GRID_BWD1=512 R2L_GRID_BWD2=768 run "bwd1=512 bwd2=768"
R2L_GRID_BWD1=512 R2L_GRID_BWD2=1024 run "bwd1=512 bwd2=1024"
R2L_GRID_FWD=1536 run "fwd=1536"
R2L_GRID_FWD=768 run "fwd=768"
run default
