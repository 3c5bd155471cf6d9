#!/bin/bash
# unsharp_masking chains: A/B of device-library builds (tests/_build/ab/<name>.so)
for n in "$@"; do
  for cfg in "--size 1024 --batch 256 --debayer bilinear --sharpening unsharp_masking --denoising gaussian_denoising" "--size 1024 --batch 256 --debayer malvar2004 --sharpening unsharp_masking --denoising median_denoising" "--size 512 --batch 1024 --debayer malvar2004 --sharpening unsharp_masking --denoising gaussian_denoising" "--size 512 --batch 1024 --debayer bilinear --sharpening unsharp_masking --denoising gaussian_denoising"; do
    R2L_LIB_PATH=tests/_build/ab/$n.so python bench.py --workload static $cfg --steps 10 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.readline())
print('%-10s %-110s %.1f us  frac %.4f' % ('$n', '$cfg', o['roofline']['avg_us'], o['roofline']['frac']))
"
  done
done
