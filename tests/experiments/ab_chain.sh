#!/bin/bash
# band-height sweep of the row-streaming luma-chain kernel (diagnostic build: R2L_CHAIN_BAND is honoured)
for lib in "$@"; do
for band in 32 64 128 256; do
  for shape in "256 1024" "1024 512"; do
    set -- $shape
    R2L_CHAIN_BAND=$band R2L_LIB_PATH=$lib python bench.py --workload static --sharpening sharpening_filter --denoising gaussian_denoising --batch $1 --size $2 --steps 20 --warmup 12 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.readline())
print('$lib band $band  %4dx%4d^2  %.1f us  frac %.4f' % ($1, $2, o['roofline']['avg_us'], o['roofline']['frac']))
"
  done
done
done
