#!/bin/bash
# Malvar2004 kernels: A/B of device-library builds (tests/_build/ab/<name>.so) on 256x1024x1024
for r in 1 2; do
for n in "$@"; do
  for cfg in "--debayer malvar2004" "--debayer malvar2004 --sharpening sharpening_filter --denoising gaussian_denoising" "--debayer malvar2004 --sharpening sharpening_filter --denoising median_denoising"; do
    R2L_LIB_PATH=tests/_build/ab/$n.so python bench.py --workload static $cfg --steps 20 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.readline())
print('%-10s %-100s %.1f us  frac %.4f' % ('$n', '$cfg', o['roofline']['avg_us'], o['roofline']['frac']))
"
  done
done
done
