#!/bin/bash
# stochastic PC sampling of one bench run (where do the waves of the fused kernels wait?)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/pcs
rm -rf $OUT; mkdir -p $OUT
(cd /tmp && timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method ${METHOD:-stochastic} --pc-sampling-unit ${UNIT:-cycles} --pc-sampling-interval ${INTERVAL:-1048576} --kernel-trace --output-format csv -d $OUT/run -- python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline "$@" > $OUT/run.log 2>&1)
echo "rc=$?"; tail -5 $OUT/run.log
find $OUT -type f | head -20
for f in $(find $OUT -name "*pc_sampling*csv" | head -3); do echo "== $f"; head -5 $f; wc -l $f; done
