#!/bin/bash
# A/B of device-library builds on the static chains (256x1024x1024 and 1024x512x512): tests/experiments/ab_static.sh lib.so ...
cd "$(dirname "$0")/../.."
for r in 1 2; do
for lib in "$@"; do
  for cfg in "--debayer bilinear" "--debayer bilinear --batch 1024 --size 512" "--debayer malvar2004" "--debayer bilinear --sharpening sharpening_filter --denoising gaussian_denoising"; do
    R2L_LIB_PATH=$PWD/$lib python bench.py --workload static $cfg --steps 20 --warmup 12 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.readline())
print('%-10s %-88s %.1f us  frac %.4f  wall ms %.4f' % ('$(basename $lib .so)', '$cfg', o['roofline']['avg_us'], o['roofline']['frac'], o['ms_per_step']))
"
  done
done
done
