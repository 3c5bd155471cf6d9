#!/bin/bash
# round-4 batch B: calibration + evidence (results under gpurun_out/r04_b/)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=gpurun_out/r04_b; mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "plane_backward_frame_shapes or dispatch_threshold or fused_forward_streaming" 2>&1 | grep -v "^\[parity\]\|^param/\|^plane\|^fwd-\|^threshold" | tail -60 > $OUT/tests.log
HOOKS=$PWD/tests/_build/libr2l_isp_hooks.so
FUZZ_KEEP_GOING=1 SEED=51 SECONDS=150 python tests/fuzz_gpu.py > $OUT/fuzz_default.txt 2>&1
FUZZ_KEEP_GOING=1 SEED=52 SECONDS=150 R2L_LIB_PATH=$HOOKS R2L_BWD_PLANES=1 python tests/fuzz_gpu.py > $OUT/fuzz_planes.txt 2>&1
FUZZ_KEEP_GOING=1 SEED=53 SECONDS=150 R2L_LIB_PATH=$HOOKS R2L_FWD_STATS_SPLIT=1 R2L_BWD_PLANES=1 R2L_BWD_SPLIT_BLUR=1 python tests/fuzz_gpu.py > $OUT/fuzz_split.txt 2>&1
tail -3 $OUT/fuzz_*.txt
python tests/bench_static.py > $OUT/static_cold.txt 2>&1
bash tests/experiments/gap_trace.sh > $OUT/gaps.log 2>&1; cp gpurun_out/gaps/gaps.txt $OUT/gaps.txt
OUTNAME=r04_b/pmc_static_malvar DEB=malvar2004 bash tests/pmc_static.sh > $OUT/pmc_static_malvar.log 2>&1
OUTNAME=r04_b/pmc_static_chain EXTRA="--sharpening sharpening_filter --denoising gaussian_denoising" bash tests/pmc_static.sh > $OUT/pmc_static_chain.log 2>&1
OUTNAME=r04_b/pmc_static_short bash tests/pmc_static.sh > $OUT/pmc_static_short.log 2>&1
rm -rf $OUT/pmc_static_*/sq1 $OUT/pmc_static_*/sq2 $OUT/pmc_static_*/tcc1 $OUT/pmc_static_*/tcc2 gpurun_out/gaps/trace
ls $OUT
