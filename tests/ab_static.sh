#!/bin/bash
# A/B of the static short chain (BASELINE config 3) on the GPU box: default build, then each tests/_build/lib_<v>.so,
# optionally under R2L_STREAM_BANDS settings (BANDS="8 32 64")
cd "$(dirname "$0")/.."
for v in default "$@"; do
  if [ "$v" = default ]; then unset R2L_LIB_PATH; else export R2L_LIB_PATH=$PWD/tests/_build/lib_$v.so; fi
  for nb in ${BANDS:-0}; do
    if [ "$nb" = 0 ]; then unset R2L_STREAM_BANDS; else export R2L_STREAM_BANDS=$nb; fi
    for deb in bilinear malvar2004; do
      python bench.py --workload static --steps 30 --warmup 5 --no-cpu-baseline --debayer $deb 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('%-10s bands=%-3s %-10s %9.0f Mpix/s  %7.1f us  %6.1f GB/s  frac %.3f' % ('$v', '$nb', '$deb', d['value'], r['avg_us'], r['achieved'], r['frac']))
"
    done
  done
done
