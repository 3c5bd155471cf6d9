#!/bin/bash
# randomised sweeps of the fused parametrized path in three modes (default dispatch; plane-pass backward forced; luma + statistics
# passes of the forward and the split middle pass of the backward forced): SECONDS per mode from $1 (default 300)
cd "$(dirname "$0")/../.."
S=${1:-300}
OUT=gpurun_out/r04_fuzz; mkdir -p $OUT
HOOKS=$PWD/tests/_build/libr2l_isp_hooks.so
FUZZ_KEEP_GOING=1 SEED=${SEED0:-61} SECONDS=$S python tests/fuzz_gpu.py > $OUT/fuzz_default.txt 2>&1
FUZZ_KEEP_GOING=1 SEED=$(( ${SEED0:-61} + 1 )) SECONDS=$S R2L_LIB_PATH=$HOOKS R2L_BWD_PLANES=1 python tests/fuzz_gpu.py > $OUT/fuzz_planes.txt 2>&1
FUZZ_KEEP_GOING=1 SEED=$(( ${SEED0:-61} + 2 )) SECONDS=$S R2L_LIB_PATH=$HOOKS R2L_FWD_STATS_SPLIT=1 R2L_BWD_PLANES=1 R2L_BWD_SPLIT_BLUR=1 python tests/fuzz_gpu.py > $OUT/fuzz_split.txt 2>&1
for f in $OUT/fuzz_*.txt; do echo "== $f"; grep -c FAIL $f; tail -n 3 $f | cut -c1-400; done
