#!/bin/bash
# round 6: GPU suite with Menon2007 / static options / spill-free fallbacks, timing of the Menon2007 chains
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=gpurun_out/r06_validate3
mkdir -p $OUT
R2L_PARITY_LOG=$PWD/$OUT/parity_gpu.tsv python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > $OUT/gputests.log
tail -3 $OUT/gputests.log
for args in "--debayer menon2007" "--debayer menon2007 --sharpening sharpening_filter --denoising gaussian_denoising"; do
  python3 bench.py --workload static --batch 64 --size 1024 --steps 10 --warmup 3 --no-cpu-baseline $args 2>> $OUT/bench.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['config']['workload'][:80], d['ms_per_step'], 'ms', d['value'], 'Mpix/s', {k.replace('r2l_launch_','').replace('_kernel',''):(v['launches'],v['avg_us']) for k,v in d['kernels'].items()})" | tee -a $OUT/menon.txt
done
