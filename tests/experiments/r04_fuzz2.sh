#!/bin/bash
# round-4 sweep of the build before the last: the bench line, then tests/fuzz_gpu.py in three dispatch modes (gpurun_out/r04_fuzz2/)
cd "$(dirname "$0")/../.."
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err
tail -c 400 gpurun_out/r04_bench_final.json
mkdir -p gpurun_out/r04_fuzz2
S=${1:-300}
HOOKS=$PWD/tests/_build/libr2l_isp_hooks.so
FUZZ_KEEP_GOING=1 SEED=71 SECONDS=$S python tests/fuzz_gpu.py > gpurun_out/r04_fuzz2/fuzz_default.txt 2>&1
FUZZ_KEEP_GOING=1 SEED=72 SECONDS=$S R2L_LIB_PATH=$HOOKS R2L_BWD_PLANES=1 python tests/fuzz_gpu.py > gpurun_out/r04_fuzz2/fuzz_planes.txt 2>&1
FUZZ_KEEP_GOING=1 SEED=73 SECONDS=$S R2L_LIB_PATH=$HOOKS R2L_FWD_STATS_SPLIT=1 R2L_BWD_PLANES=1 R2L_BWD_SPLIT_BLUR=1 python tests/fuzz_gpu.py > gpurun_out/r04_fuzz2/fuzz_split.txt 2>&1
for f in gpurun_out/r04_fuzz2/fuzz_*.txt; do echo "== $f"; grep -c FAIL $f; tail -n 2 $f | cut -c1-300; done
