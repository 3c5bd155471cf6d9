#!/bin/bash
# round 6: the full GPU suite N times (one failure of the one-rank RCCL graph-capture test in four runs, none in 14 partial repeats):
# keep every run's complete report and parity log
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
N=${1:-5}
OUT=gpurun_out/r06_suite; rm -rf $OUT; mkdir -p $OUT
for i in $(seq 1 $N); do
  R2L_PARITY_LOG=$PWD/$OUT/parity_gpu_$i.tsv python -m pytest tests -x -q -m gpu > $OUT/run$i.log 2>&1
  echo "run $i rc $?  $(grep -E ' passed| failed' $OUT/run$i.log | tail -1)" >> $OUT/summary.txt
done
cat $OUT/summary.txt
for i in $(seq 1 $N); do if grep -q " failed" $OUT/run$i.log; then echo "=== run $i"; grep -v "^multirank\|^param\|^static\|^canary\|^fwd-stats\|^stream\|^planes\|^shapes" $OUT/run$i.log | head -200 | cut -c1-600; fi; done
