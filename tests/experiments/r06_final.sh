#!/bin/bash
# round-6 artefacts (gpurun_out/r06_z/ ...): PMC passes first (their JSONs, tagged with the digest of the sources that ran, are what
# bench.py's roofline.traffic reads), the bench line + rocprofv3 kernel stats + SQ counters (tests/profile_round.sh), the GPU suite
# with its parity log, the Menon2007 timings
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
OUTNAME=r06_pmc_static_short bash tests/pmc_static.sh > gpurun_out/r06_pmc_static_short.log 2>&1
OUTNAME=r06_pmc_static_malvar DEB=malvar2004 bash tests/pmc_static.sh > gpurun_out/r06_pmc_static_malvar.log 2>&1
OUTNAME=r06_pmc_static_chain EXTRA="--sharpening sharpening_filter --denoising gaussian_denoising" bash tests/pmc_static.sh > gpurun_out/r06_pmc_static_chain.log 2>&1
rm -rf gpurun_out/r06_pmc_static_*/sq1 gpurun_out/r06_pmc_static_*/sq2 gpurun_out/r06_pmc_static_*/tcc1 gpurun_out/r06_pmc_static_*/tcc2
python3 tests/tools/merge_pmc.py profiles/r06_pmc_traffic_static.json r06 gpurun_out/r06_pmc_static_short/pmc_traffic_static.json \
    gpurun_out/r06_pmc_static_malvar/pmc_traffic_static.json gpurun_out/r06_pmc_static_chain/pmc_traffic_static.json
bash tests/profile_round.sh r06_z > gpurun_out/r06_profile_round.log 2>&1
python3 tests/tools/merge_pmc.py profiles/r06_pmc_traffic.json r06 gpurun_out/r06_z/pmc_traffic.json
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r06_z/bench_final.json 2>> gpurun_out/r06_z/bench.err     # (with this round's PMC files in place)
R2L_PARITY_LOG=$PWD/gpurun_out/r06_parity_gpu.tsv python -m pytest tests -x -q -m gpu > gpurun_out/r06_gputests.log 2>&1     # (the WHOLE report: a tail lost the one failure this round saw)
cp profiles/r06_pmc_traffic.json profiles/r06_pmc_traffic_static.json gpurun_out/r06_z/ 2>/dev/null
for args in "--debayer menon2007" "--debayer menon2007 --sharpening sharpening_filter --denoising gaussian_denoising" "--debayer menon2007 --sharpening unsharp_masking --denoising fft_denoising"; do
  python3 bench.py --workload static --batch 64 --size 1024 --steps 10 --warmup 3 --no-cpu-baseline $args 2>> gpurun_out/r06_z/bench.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['config']['workload'][:90], '|', d['ms_per_step'], 'ms |', d['value'], 'Mpix/s |', {k.replace('r2l_launch_','').replace('_kernel',''):(v['launches'],v['avg_us']) for k,v in d['kernels'].items()})" >> gpurun_out/r06_z/menon.txt
done
python3 tests/kernel_resources.py > gpurun_out/r06_z/kernel_resources.txt 2>&1
tail -3 gpurun_out/r06_gputests.log; tail -c 1200 gpurun_out/r06_z/bench_final.json; cat gpurun_out/r06_z/menon.txt
