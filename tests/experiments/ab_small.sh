#!/bin/bash
# the small shapes (64x256x256, 128x256x256: the reference's tile size, dataset.py:92) under several environment
# settings: tests/experiments/ab_small.sh "VAR=value ..." ...   ("-" = none)
cd "$(dirname "$0")/../.."
for r in 1 2; do
for envs in "$@"; do
  [ "$envs" = "-" ] && envs="R2L_NOTHING=1"
  env $envs python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-static-c3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for r in d['small_shapes']:
    print('%-50s' % '$envs'.split('/')[-1], r['shape'], 'ms/step', r['ms_per_step'], 'kernels', r['kernels_us_per_step'], ' '.join('%s=%.1f' % kv for kv in sorted(r['kernels'].items())))
"
done
done
