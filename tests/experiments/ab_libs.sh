#!/bin/bash
# A/B of device-library builds over the headline bench and the static workloads: tests/experiments/ab_libs.sh lib.so ...
for r in 1 2; do
for lib in "$@"; do
  R2L_LIB_PATH=$lib python bench.py --steps 30 --warmup 5 --quick 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.readline())
k = o['kernels']
print('%-28s ms/step %.4f ' % ('$lib'.split('/')[-1], o['ms_per_step']) + ' '.join('%s=%.1f' % (a.replace('r2l_launch_', '').replace('_kernel', ''), b['avg_us']) for a, b in sorted(k.items())))
"
done
done
for lib in "$@"; do
  for cfg in "--debayer bilinear" "--debayer bilinear --sharpening sharpening_filter --denoising gaussian_denoising" "--debayer malvar2004" "--debayer bilinear --sharpening unsharp_masking --denoising median_denoising"; do
    R2L_LIB_PATH=$lib python bench.py --workload static $cfg --steps 20 --warmup 12 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.readline())
print('%-28s %-90s %.1f us  frac %.4f' % ('$lib'.split('/')[-1], '$cfg', o['roofline']['avg_us'], o['roofline']['frac']))
"
  done
done
