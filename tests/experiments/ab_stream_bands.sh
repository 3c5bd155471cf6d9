#!/bin/bash
# band-count sweep of the static short chain (diagnostic build): tests/experiments/ab_stream_bands.sh B S nband...
B=$1; S=$2; shift 2
for nb in "$@"; do
  for rep in 1 2; do
  R2L_STREAM_BANDS=$nb R2L_LIB_PATH=tests/_build/libr2l_isp_hooks.so python bench.py --workload static --batch $B --size $S --steps 30 --warmup 12 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.readline())
print('%dx%d^2 bands %s: %.1f us frac %.4f' % ($B, $S, '$nb', o['roofline']['avg_us'], o['roofline']['frac']))
"
  done
done
