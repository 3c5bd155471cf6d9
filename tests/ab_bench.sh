#!/bin/bash
# A/B of kernel builds on the GPU box: runs pytest -m gpu on the default build, then bench.py for each variant
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -4 gpurun_out/pytest_gpu.log
for v in default "$@"; do
  if [ "$v" = default ]; then unset R2L_LIB_PATH; else export R2L_LIB_PATH=$PWD/tests/_build/lib_$v.so; fi
  echo "== $v"
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['avg_us']*kv[1]['launches']): print('  %-34s %3d x %8.2f us' % (k, v['launches'], v['avg_us']))
"
done
