#!/bin/bash
# second sweep on the final build + the bench's graph-trial branch on the one-GPU box
cd "$(dirname "$0")/../.."
python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "graph_trial or step_graph" 2>&1 | tail -4
mkdir -p gpurun_out/r04_fuzz3
S=${1:-250}
HOOKS=$PWD/tests/_build/libr2l_isp_hooks.so
FUZZ_KEEP_GOING=1 SEED=${SEEDA:-91} SECONDS=$S python tests/fuzz_gpu.py > gpurun_out/r04_fuzz3/fuzz_default.txt 2>&1
FUZZ_KEEP_GOING=1 SEED=$(( ${SEEDA:-91} + 1 )) SECONDS=$S R2L_LIB_PATH=$HOOKS R2L_BWD_PLANES=1 python tests/fuzz_gpu.py > gpurun_out/r04_fuzz3/fuzz_planes.txt 2>&1
FUZZ_KEEP_GOING=1 SEED=$(( ${SEEDA:-91} + 2 )) SECONDS=$S R2L_LIB_PATH=$HOOKS R2L_FWD_STATS_SPLIT=1 R2L_BWD_PLANES=1 R2L_BWD_SPLIT_BLUR=1 python tests/fuzz_gpu.py > gpurun_out/r04_fuzz3/fuzz_split.txt 2>&1
for f in gpurun_out/r04_fuzz3/fuzz_*.txt; do echo "== $f"; grep -c FAIL $f; tail -n 2 $f | cut -c1-300; done
