#!/bin/bash
# round 6: prefetch depth 3 (rows in flight per stream) for the apply pass and kernel B1's plane pass, back to back and COLD
# (tests/_build/ab/fa3.so = -DR2L_FA_PF=3, bp3.so = -DR2L_BP_PF=3; bnr_planes at depth 3 spills at three wavefronts per SIMD: not run)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=gpurun_out/r06_cold_pf.txt
{
for rnd in 1 2 3; do for v in default fa3 bp3; do
  if [ $v = default ]; then unset R2L_LIB_PATH; else export R2L_LIB_PATH=$PWD/tests/_build/ab/$v.so; fi
  python3 bench.py --quick --cold --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k={n.replace('r2l_launch_','').replace('_kernel',''):v['avg_us'] for n,v in d['kernels'].items()}
c=d['cold']
print('$v', 'ms/step %.4f' % d['ms_per_step'], ' '.join('%s=%.1f'%kv for kv in sorted(k.items())), '| COLD %.4f ms' % c['ms_per_step_kernels'], ' '.join('%s=%.1f'%kv for kv in sorted(c['kernels'].items())))"
done; done
} > $OUT 2>&1
cat $OUT
