#!/bin/bash
# after the BatchNorm-backward change: A/B of the gradient precision against the build before it, the GPU suite, the sweeps, the bench line
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r04b
bash tests/experiments/r04_ab_fuzz.sh 120 400 > gpurun_out/r04b/ab_fuzz.txt 2>&1
R2L_PARITY_LOG=$PWD/gpurun_out/r04_parity_gpu.tsv python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r04b/gputests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04b/smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/r04b/smoke.log
SEED0=${SEED0:-101} bash tests/experiments/r04_fuzz.sh ${FUZZ_S:-300} > gpurun_out/r04b/fuzz.log 2>&1
SEED=103 SECONDS=${FUZZ_S:-300} python tests/fuzz_more.py > gpurun_out/r04_fuzz/fuzz_more.txt 2>&1
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r04b/bench.json 2> gpurun_out/r04b/bench.err
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/r04b/stats -- python3 $OLDPWD/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-small-shapes > $OLDPWD/gpurun_out/r04b/stats.log 2>&1)
cp $(find gpurun_out/r04b/stats -name "*kernel_stats.csv" | head -1) gpurun_out/r04b/kernel_stats.csv; rm -rf gpurun_out/r04b/stats
cat gpurun_out/r04b/ab_fuzz.txt | cut -c1-200; cat gpurun_out/r04b/gputests.log gpurun_out/r04b/smoke.log | tail -6; tail -12 gpurun_out/r04b/fuzz.log | cut -c1-300; tail -c 500 gpurun_out/r04b/bench.json
