#!/bin/bash
# bench.py under several environment settings: each argument is a space-separated list of VAR=value
cd "$(dirname "$0")/.."
for envs in "$@"; do
  echo "== $envs"
  env $envs python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], ' '.join('%s=%.1f' % (k.replace('r2l_launch_','').replace('_kernel',''), v['avg_us']) for k,v in d['kernels'].items()))
"
done
