#!/bin/bash
# every static pipeline of the reference's sweeps (figures/train.sh: debayer x sharpening x denoising) on 256x1024x1024
for deb in bilinear malvar2004; do
for sh in none sharpening_filter unsharp_masking; do
for dn in none gaussian_denoising median_denoising; do
  python bench.py --workload static --debayer $deb --sharpening $sh --denoising $dn --steps 10 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.readline())
k = o.get('kernels', {})
print('%-11s %-18s %-19s %8.1f Gpix/s  %8.1f us/step  step frac %.3f  ' % ('$deb', '$sh', '$dn', o['value'] / 1e3, 1e3 * o['ms_per_step'], o['value'] * 1e6 * 16 / 8e12) + ' '.join('%s=%.0f' % (a.replace('r2l_launch_', '').replace('_kernel', ''), b['avg_us']) for a, b in sorted(k.items())))
"
done; done; done
