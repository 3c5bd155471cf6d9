#!/bin/bash
# randomised sweeps on the round-5 build (as r04_fuzz.sh: default dispatch; plane-pass backward forced; luma + statistics passes and the
# split middle pass forced) + fuzz_more (static chains incl. the 10-row Malvar2004 bands, raw2rgb, staged path, SSIM / L2): SECONDS per mode from $1
cd "$(dirname "$0")/../.."
S=${1:-300}
OUT=gpurun_out/r05_fuzz; mkdir -p $OUT
HOOKS=$PWD/tests/_build/libr2l_isp_hooks.so
FUZZ_KEEP_GOING=1 SEED=${SEED0:-201} SECONDS=$S python tests/fuzz_gpu.py > $OUT/fuzz_default.txt 2>&1
FUZZ_KEEP_GOING=1 SEED=$(( ${SEED0:-201} + 1 )) SECONDS=$S R2L_LIB_PATH=$HOOKS R2L_BWD_PLANES=1 python tests/fuzz_gpu.py > $OUT/fuzz_planes.txt 2>&1
FUZZ_KEEP_GOING=1 SEED=$(( ${SEED0:-201} + 2 )) SECONDS=$S R2L_LIB_PATH=$HOOKS R2L_FWD_STATS_SPLIT=1 R2L_BWD_PLANES=1 R2L_BWD_SPLIT_BLUR=1 python tests/fuzz_gpu.py > $OUT/fuzz_split.txt 2>&1
SEED=$(( ${SEED0:-201} + 3 )) SECONDS=$S python tests/fuzz_more.py > $OUT/fuzz_more.txt 2>&1
for f in $OUT/fuzz_*.txt; do echo "== $f"; grep -c FAIL $f; grep -v amdgpu $f | tail -n 4 | cut -c1-400; done
