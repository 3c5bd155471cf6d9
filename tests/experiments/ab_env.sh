#!/bin/bash
# bench.py (the driver's arguments) under several environment settings, each argument a space-separated list of VAR=value
# ("-" = none), the whole list twice:  tests/experiments/ab_env.sh "R2L_LIB_PATH=... R2L_X=1" "-"
cd "$(dirname "$0")/../.."
for r in 1 2; do
for envs in "$@"; do
  [ "$envs" = "-" ] && envs="R2L_NOTHING=1"
  env $envs python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-small-shapes 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-60s' % '$envs'.split('/')[-1], 'value', d['value'], 'ms/step', d['ms_per_step'], ' '.join('%s=%.1f' % (k.replace('r2l_launch_','').replace('_kernel',''), v['avg_us']) for k,v in sorted(d['kernels'].items())))
"
done
done
