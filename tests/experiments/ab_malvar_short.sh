#!/bin/bash
# Malvar2004 short chain (BASELINE config 3): A/B of device-library builds on 256x1024x1024 and 1024x512x512
names="$*"
for r in 1 2; do
for n in $names; do
  for shape in "256 1024" "1024 512"; do
    set -- $shape
    R2L_LIB_PATH=tests/_build/ab/$n.so python bench.py --workload static --debayer malvar2004 --batch $1 --size $2 --steps 20 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.readline())
print('%-10s %4dx%4d  %.1f us  frac %.4f  wall %.4f ms' % ('$n', $1, $2, o['roofline']['avg_us'], o['roofline']['frac'], o['ms_per_step']))
"
  done
done
done
