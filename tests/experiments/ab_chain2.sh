#!/bin/bash
# A/B of device-library builds on the static luma chains (256x1024x1024): tests/experiments/ab_chain2.sh lib.so ...
cd "$(dirname "$0")/../.."
for r in 1 2; do
for lib in "$@"; do
  for cfg in "--debayer bilinear --sharpening sharpening_filter --denoising gaussian_denoising" "--debayer malvar2004 --sharpening sharpening_filter --denoising gaussian_denoising" "--debayer bilinear --sharpening sharpening_filter --denoising median_denoising" "--debayer bilinear --sharpening unsharp_masking --denoising gaussian_denoising" "--debayer malvar2004"; do
    R2L_LIB_PATH=$PWD/$lib python bench.py --workload static $cfg --steps 20 --warmup 12 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.readline())
print('%-10s %-96s %.1f us  frac %.4f  wall ms %.4f' % ('$(basename $lib .so)', '$cfg', o['roofline']['avg_us'], o['roofline']['frac'], o['ms_per_step']))
"
  done
done
done
