#!/bin/bash
# A/B of kernel builds on the GPU box WITHOUT the test run: bench.py per variant, twice each, interleaved
cd "$(dirname "$0")/../.."
for rep in 1 2; do
for v in default "$@"; do
  if [ "$v" = default ]; then unset R2L_LIB_PATH; else export R2L_LIB_PATH=$PWD/tests/_build/lib_$v.so; fi
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-10s value %9.1f ms/step %.4f ' % ('$v', d['value'], d['ms_per_step']) + ' '.join('%s=%.1f' % (k.replace('r2l_launch_','').replace('_kernel',''), v['avg_us']) for k,v in sorted(d['kernels'].items())))
"
done
done
