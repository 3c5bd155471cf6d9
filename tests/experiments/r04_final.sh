#!/bin/bash
# round-4 artefacts (gpurun_out/r04/ ...): tests, bench + rocprofv3 + PMC (parametrized and static kernels), fuzz sweeps, timelines
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
R2L_PARITY_LOG=$PWD/gpurun_out/r04_parity_gpu.tsv python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r04_gputests.log
bash tests/profile_round.sh r04 > gpurun_out/r04_profile_round.log 2>&1
OUTNAME=r04/pmc_static_short bash tests/pmc_static.sh > gpurun_out/r04/pmc_static_short.log 2>&1
OUTNAME=r04/pmc_static_malvar DEB=malvar2004 bash tests/pmc_static.sh > gpurun_out/r04/pmc_static_malvar.log 2>&1
OUTNAME=r04/pmc_static_chain EXTRA="--sharpening sharpening_filter --denoising gaussian_denoising" bash tests/pmc_static.sh > gpurun_out/r04/pmc_static_chain.log 2>&1
rm -rf gpurun_out/r04/pmc_static_*/sq1 gpurun_out/r04/pmc_static_*/sq2 gpurun_out/r04/pmc_static_*/tcc1 gpurun_out/r04/pmc_static_*/tcc2
R2L_STAMPS_LIB=lib_tl_noprio.so python3 tests/timeline_fwd.py > gpurun_out/r04/timeline_fwd_noprio.txt 2>&1
R2L_TL_BWD=1 R2L_STAMPS_LIB=lib_tl.so python3 tests/timeline_fwd.py > gpurun_out/r04/timeline_fwd_bwd.txt 2>&1
R2L_TL_BWD=1 R2L_STAMPS_LIB=lib_tl_noprio.so python3 tests/timeline_fwd.py > gpurun_out/r04/timeline_fwd_bwd_noprio.txt 2>&1
bash tests/experiments/r04_fuzz.sh ${FUZZ_S:-300} > gpurun_out/r04_fuzz.log 2>&1
SEED=63 SECONDS=${FUZZ_S:-300} python tests/fuzz_more.py > gpurun_out/r04_fuzz/fuzz_more.txt 2>&1
tail -3 gpurun_out/r04_gputests.log; tail -c 700 gpurun_out/r04/bench.json; tail -12 gpurun_out/r04_fuzz.log
