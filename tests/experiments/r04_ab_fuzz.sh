#!/bin/bash
# A/B of the gradient precision: the build before the tails were shortened (tests/_build/lib_old*.so, from commit 6d1fd1c) against
# the current one (build it first: git archive 6d1fd1c raw2logit_amd/csrc include | tar -x -C <dir>; hipcc ... -DR2L_TEST_HOOKS <dir>/raw2logit_amd/csrc/r2l_api.hip -o tests/_build/lib_old_hooks.so), the same seeded cases (additive layer + BatchNorm train + midtone frames: the class of the sweep's one miss)
cd "$(dirname "$0")/../.."
for lib in lib_old_hooks.so libr2l_isp_hooks.so; do
  for seed in 5 6; do
    echo "== $lib seed $seed additive,train,midtone"
    FUZZ_KEEP_GOING=1 FUZZ_FORCE=additive,train,midtone FUZZ_CASES=${1:-120} SEED=$seed R2L_LIB_PATH=$PWD/tests/_build/$lib python tests/fuzz_gpu.py 2>&1 | grep -v amdgpu.ids | grep "worst gradient\|over its limit\|random cases" | cut -c1-260
  done
  echo "== $lib seed 7 train,midtone (any shape)"
  FUZZ_KEEP_GOING=1 FUZZ_FORCE=train,midtone FUZZ_CASES=${2:-400} SEED=7 R2L_LIB_PATH=$PWD/tests/_build/$lib python tests/fuzz_gpu.py 2>&1 | grep -v amdgpu.ids | grep "worst gradient\|over its limit\|random cases" | cut -c1-260
done
