#!/bin/bash
# fwd+bwd step time at the shapes of BASELINE configs 4 / 5 (per-GPU shard) next to the headline shape
for bs in "64 512" "128 256" "64 256" "16 1024" "256 512"; do
  set -- $bs
  python bench.py --batch $1 --size $2 --steps 50 --warmup 10 --no-cpu-baseline --no-static-c3 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.readline())
k = o['kernels']
tot = sum(v['launches'] * v['avg_us'] for v in k.values()) / o['steps']
print('%4dx%4d^2  %.4f ms/step  %8.1f Mpix/s  kernels %.1f us/step  host+gaps %.1f us' % ($1, $2, o['ms_per_step'], o['value'], tot, 1e3 * o['ms_per_step'] - tot))
"
done
