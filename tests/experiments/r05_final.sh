#!/bin/bash
# round-5 artefacts (gpurun_out/r05_z/ ...): PMC passes first (their JSONs are what bench.py's roofline.traffic reads), then the GPU
# suite with its parity log, the bench line + rocprofv3 kernel stats + SQ counters (tests/profile_round.sh), static PMC passes
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
OUTNAME=r05_pmc_static_short bash tests/pmc_static.sh > gpurun_out/r05_pmc_static_short.log 2>&1
OUTNAME=r05_pmc_static_malvar DEB=malvar2004 bash tests/pmc_static.sh > gpurun_out/r05_pmc_static_malvar.log 2>&1
OUTNAME=r05_pmc_static_chain EXTRA="--sharpening sharpening_filter --denoising gaussian_denoising" bash tests/pmc_static.sh > gpurun_out/r05_pmc_static_chain.log 2>&1
rm -rf gpurun_out/r05_pmc_static_*/sq1 gpurun_out/r05_pmc_static_*/sq2 gpurun_out/r05_pmc_static_*/tcc1 gpurun_out/r05_pmc_static_*/tcc2
python3 - <<'PY'
import json, glob
m = {}
for f in sorted(glob.glob('gpurun_out/r05_pmc_static_*/pmc_traffic_static.json')):
    m.update(json.load(open(f)))
json.dump(m, open('profiles/r05_pmc_traffic_static.json', 'w'), indent=1)
print('static PMC kernels:', sorted(m))
PY
bash tests/profile_round.sh r05_z > gpurun_out/r05_profile_round.log 2>&1
cp gpurun_out/r05_z/pmc_traffic.json profiles/r05_pmc_traffic.json
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r05_z/bench_final.json 2>> gpurun_out/r05_z/bench.err     # (with this round's PMC files in place)
R2L_PARITY_LOG=$PWD/gpurun_out/r05_parity_gpu.tsv python -m pytest tests -x -q -m gpu 2>&1 | tail -80 > gpurun_out/r05_gputests.log
cp profiles/r05_pmc_traffic.json profiles/r05_pmc_traffic_static.json gpurun_out/r05_z/ 2>/dev/null
tail -3 gpurun_out/r05_gputests.log; tail -c 900 gpurun_out/r05_z/bench_final.json
