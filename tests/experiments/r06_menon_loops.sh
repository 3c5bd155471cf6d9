#!/bin/bash
# round 6: the Menon2007 stage kernel's loop form (tests/_build/ab/mloop{0,1,2}.so = -DR2L_MENON_LOOP=0 flat loop with 64-bit index
# decomposition, 1 flat with 32-bit decomposition, 2 a workgroup per image row), 64x1024x1024 and 256x256x256, short chain
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=gpurun_out/r06_menon_loops.txt
{
for shape in "64 1024" "256 256"; do set -- $shape
for rnd in 1 2; do for v in mloop0 mloop1 mloop2; do
  R2L_LIB_PATH=$PWD/tests/_build/ab/$v.so python3 bench.py --workload static --debayer menon2007 --batch $1 --size $2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', '$1 x $2 x $2', d['ms_per_step'], 'ms', {k.replace('r2l_launch_','').replace('_kernel',''):v['avg_us'] for k,v in d['kernels'].items()})"
done; done; done
} > $OUT 2>&1
cat $OUT
