#!/bin/bash
# cost of the fused T.Normalize epilogue on the static workloads
for cfg in "--debayer bilinear" "--debayer bilinear --sharpening sharpening_filter --denoising gaussian_denoising" "--debayer malvar2004 --sharpening sharpening_filter --denoising median_denoising"; do
for n in "" "--normalize"; do
  python bench.py --workload static $cfg $n --steps 20 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.readline())
print('%-100s %-12s %.1f us/step  kernel %.1f us' % ('$cfg', '$n', 1e3 * o['ms_per_step'], o['roofline']['avg_us']))
"
done; done
