#!/bin/bash
# round-2 artefacts: tests/profile_round.sh r02_r + the PMC passes of three static chains (results under gpurun_out/)
cd "$(dirname "$0")/../.."
bash tests/profile_round.sh r02_r > gpurun_out/profile_round.log 2>&1
OUTNAME=pmc_static_short bash tests/pmc_static.sh > gpurun_out/pmc_static_short.log 2>&1
OUTNAME=pmc_static_chain EXTRA="--sharpening sharpening_filter --denoising gaussian_denoising" bash tests/pmc_static.sh > gpurun_out/pmc_static_chain.log 2>&1
OUTNAME=pmc_static_malvar_median DEB=malvar2004 EXTRA="--sharpening sharpening_filter --denoising median_denoising" bash tests/pmc_static.sh > gpurun_out/pmc_static_malvar_median.log 2>&1
tests/static_matrix.sh > gpurun_out/static_matrix_final.txt 2>&1
tail -3 gpurun_out/profile_round.log
