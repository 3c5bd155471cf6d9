#!/bin/bash
# the measured artefacts once more, all from the round's last build (no sweeps: tests/experiments/r04_fuzz3.sh did those)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tests/profile_round.sh r04 > gpurun_out/r04_profile_round.log 2>&1
R2L_STAMPS_LIB=lib_tl_noprio.so python3 tests/timeline_fwd.py > gpurun_out/r04/timeline_fwd_noprio.txt 2>&1
R2L_TL_BWD=1 R2L_STAMPS_LIB=lib_tl.so python3 tests/timeline_fwd.py > gpurun_out/r04/timeline_fwd_bwd.txt 2>&1
R2L_TL_BWD=1 R2L_STAMPS_LIB=lib_tl_noprio.so python3 tests/timeline_fwd.py > gpurun_out/r04/timeline_fwd_bwd_noprio.txt 2>&1
bash tests/experiments/r04_small.sh > gpurun_out/r04/small.txt 2>&1
tail -c 600 gpurun_out/r04/bench.json; ls gpurun_out/r04 | head -40
