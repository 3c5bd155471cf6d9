#!/bin/bash
# fwd+bwd step time at the shapes of BASELINE configs 4 / 5 (per-GPU shard) next to the headline shape
for bs in ${SHAPES:-64x512 128x256 64x256 16x1024 256x512}; do
  set -- ${bs%x*} ${bs#*x}
  python bench.py --batch $1 --size $2 --steps 50 --warmup 10 --quick ${EXTRA} 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.readline())
k = o['kernels']
tot = sum(v['launches'] * v['avg_us'] for v in k.values()) / o['steps']
print('%4dx%4d^2  %.4f ms/step  %8.1f Mpix/s  kernels %.1f us/step  host+gaps %.1f us   ' % ($1, $2, o['ms_per_step'], o['value'], tot, 1e3 * o['ms_per_step'] - tot) + ' '.join('%s=%.1f' % (n.replace('r2l_launch_', '').replace('_kernel', ''), v['avg_us']) for n, v in sorted(k.items())))
"
done
