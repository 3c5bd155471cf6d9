#!/bin/bash
# whole-step A/B of library builds at the headline shape, alternating processes: tests/experiments/ab_step_libs.sh libA.so libB.so ... (paths from the repo root)
cd "$(dirname "$0")/../.."
for rep in 1 2 3; do
  for L in "$@"; do
    R2L_LIB_PATH=$PWD/$L python bench.py --steps 60 --warmup 10 --no-roofline --quick 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s %.4f ms/step' % ('$L', d['ms_per_step']))"
  done
done
