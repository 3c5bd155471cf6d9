#!/bin/bash
# round 6: GPU suite (parity log) + the bench line with its new sub-records (cold, fallback_paths, pass_rooflines)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r06_validate}
mkdir -p $OUT
R2L_PARITY_LOG=$PWD/$OUT/parity_gpu.tsv python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > $OUT/gputests.log
python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
tail -5 $OUT/gputests.log; tail -c 2500 $OUT/bench.json; tail -5 $OUT/bench.err
