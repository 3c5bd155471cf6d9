#!/bin/bash
cd /root/repo
run() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-static-c3 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.readline()); k = o['kernels']
print('%-40s ms/step %.4f ' % ('$1', o['ms_per_step']) + ' '.join('%s=%.1f' % (a.replace('r2l_launch_', '').replace('_kernel', ''), b['avg_us']) for a, b in sorted(k.items())))
"; }
export R2L_LIB_PATH=tests/_build/ab/b1s_occ1.so
run "saved occ1 grid256"
R2L_BWD1_RECOMPUTE=1 run "recompute grid256"
export R2L_LIB_PATH=tests/_build/ab/b1s_occ2.so
run "saved occ2 grid256"
R2L_GRID_BWD1=512 run "saved occ2 grid512"
R2L_GRID_BWD1=384 run "saved occ2 grid384"
R2L_GRID_BWD1=512 R2L_BWD1_RECOMPUTE=1 run "recompute grid512"
