#!/bin/bash
# randomised sweeps on the round-6 build (as r05_fuzz.sh; fuzz_more now draws menon2007 and processing()'s numeric arguments): SECONDS per mode from $1
cd "$(dirname "$0")/../.."
S=${1:-300}
OUT=gpurun_out/r06_fuzz; mkdir -p $OUT
HOOKS=$PWD/tests/_build/libr2l_isp_hooks.so
FUZZ_KEEP_GOING=1 SEED=${SEED0:-401} SECONDS=$S python tests/fuzz_gpu.py > $OUT/fuzz_default.txt 2>&1
FUZZ_KEEP_GOING=1 SEED=$(( ${SEED0:-401} + 1 )) SECONDS=$S R2L_LIB_PATH=$HOOKS R2L_BWD_PLANES=1 python tests/fuzz_gpu.py > $OUT/fuzz_planes.txt 2>&1
FUZZ_KEEP_GOING=1 SEED=$(( ${SEED0:-401} + 2 )) SECONDS=$S FUZZ_FORCE=additive python tests/fuzz_gpu.py > $OUT/fuzz_additive.txt 2>&1
SEED=$(( ${SEED0:-401} + 3 )) SECONDS=$S python tests/fuzz_more.py > $OUT/fuzz_more.txt 2>&1
SEED=$(( ${SEED0:-401} + 4 )) SECONDS=$S WHICH=static python tests/fuzz_more.py > $OUT/fuzz_static.txt 2>&1
for f in $OUT/fuzz_*.txt; do echo "== $f"; grep -c FAIL $f; grep -v amdgpu $f | tail -n 4 | cut -c1-400; done
