#!/bin/bash
# A/B of the apply pass on the kept luma plane (r2l_fwd_apply_block) against the streaming apply pass: band height
# (BANDS="8 16"), workgroup -> item map (MAPS="2 1": identity / one contiguous range per XCD)
#   tests/experiments/ab_apply.sh lib.so [lib.so ...]      (diagnostic builds: the R2L_* environment switches only exist there)
cd "$(dirname "$0")/../.."
run() {
  lib=$1; shift
  env "$@" R2L_LIB_PATH=$lib python bench.py --steps 30 --warmup 5 --quick 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.readline())
k = o['kernels']
print('%-14s %-28s ms/step %.4f ' % ('$lib'.split('/')[-1], '$*', o['ms_per_step']) + ' '.join('%s=%.1f' % (a.replace('r2l_launch_', '').replace('_kernel', ''), b['avg_us']) for a, b in sorted(k.items())))
"
}
for r in 1 2; do
  for lib in "$@"; do
    [ -n "$NOREF" ] || run $lib R2L_FWD_APPLY_RECOMPUTE=1
    for m in ${MAPS:-2}; do for b in ${BANDS:-8 16 32}; do run $lib R2L_FA_BAND=$b R2L_FA_MAP=$m; done; done
  done
done
