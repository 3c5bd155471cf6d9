#!/bin/bash
# round 5, GPU batch 3: the re-pinned plane-backward tests; static A/B on the same buffers (round-3 library, Malvar2004 with lane shifts)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "dispatch_threshold or headline_shape" 2>&1 | tail -45 > gpurun_out/r05_run3_tests.txt
DEB=0 python tests/static_ab_inproc.py r03=lib_r03.so ml=ab/ml.so > gpurun_out/r05_static_ab_bilinear.txt 2>&1
DEB=1 python tests/static_ab_inproc.py r03=lib_r03.so ml=ab/ml.so > gpurun_out/r05_static_ab_malvar.txt 2>&1
DEB=0 python tests/static_ab_inproc.py r03=lib_r03.so ml=ab/ml.so >> gpurun_out/r05_static_ab_bilinear.txt 2>&1
DEB=1 python tests/static_ab_inproc.py r03=lib_r03.so ml=ab/ml.so >> gpurun_out/r05_static_ab_malvar.txt 2>&1
python tests/lib_ab.py ../../raw2logit_amd/libr2l_isp.so lib_r04.so > gpurun_out/r05_lib_ab_scratch.txt 2>&1
tail -30 gpurun_out/r05_run3_tests.txt; grep -v amdgpu gpurun_out/r05_static_ab_bilinear.txt; grep -v amdgpu gpurun_out/r05_static_ab_malvar.txt; grep -v amdgpu gpurun_out/r05_lib_ab_scratch.txt
