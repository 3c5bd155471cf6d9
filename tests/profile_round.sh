#!/bin/bash
# Round artefacts on the GPU box: bench line, rocprofv3 kernel stats of the same command, PMC passes
# (counters in their own runs, kernel-trace only).  Usage: bash tests/profile_round.sh <tag>
# Results land in gpurun_out/<tag>/ ; copy what is to be judged into profiles/.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-r04}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-small-shapes > $OUT/stats.log 2>&1)
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
run() { n=$1; shift; (cd /tmp && R2L_BENCH_PREROLL_S=0 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- python3 $ROOT/bench.py --steps 3 --warmup 1 --quick --no-roofline > $OUT/$n.log 2>&1); }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE
python3 - "$OUT" <<'PY'
import csv, glob, collections, json, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if k.startswith('r2l_'):
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
with open(out + '/pmc_summary.txt', 'w') as fh:
    for k in sorted(agg):
        fh.write(k + '\n')
        for c, v in sorted(agg[k].items()):
            fh.write('   %-28s %14.0f  (n=%d)\n' % (c, sum(v) / len(v), len(v)))
# HBM traffic per launch: FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE under-counts by 2x on gfx950
# (MI355X_MICROARCH.md, HBM / rocprofv3 section), WRITE_SIZE is taken as read
traffic = {}
for k, c in agg.items():
    if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
        fb = 2.0 * 1024.0 * sum(c['FETCH_SIZE']) / len(c['FETCH_SIZE'])
        wb = 1024.0 * sum(c['WRITE_SIZE']) / len(c['WRITE_SIZE'])
        traffic[k] = {'fetch_bytes': fb, 'write_bytes': wb, 'total_bytes': fb + wb,
                      'note': 'FETCH_SIZE x2 (gfx950 correction), WRITE_SIZE as read; KiB -> bytes; 64x512x512 '
                              'workload; averaged over the launches of a step (fwd: stats-only and apply)'}
json.dump(traffic, open(out + '/pmc_traffic.json', 'w'), indent=1)
print(open(out + '/pmc_summary.txt').read()[-1500:])
PY
rm -rf $OUT/stats $OUT/sq1 $OUT/sq2 $OUT/tcc1 $OUT/tcc2
# the other measured paths: 16-bit ingest, static chains (BASELINE config 3), auxiliary losses, phase stamps
python3 bench.py --steps 30 --warmup 5 --raw-u16 --no-cpu-baseline > $OUT/bench_u16.json 2>> $OUT/bench.err
python3 bench.py --workload static --steps 30 --warmup 10 > $OUT/bench_static.json 2>> $OUT/bench.err
python3 bench.py --workload static --sharpening sharpening_filter --denoising gaussian_denoising --steps 30 --warmup 12 --no-cpu-baseline > $OUT/bench_static_default_chain.json 2>> $OUT/bench.err
python3 bench.py --workload e2e-microscopy --steps 20 > $OUT/bench_e2e_microscopy.json 2>> $OUT/bench.err
python3 bench.py --workload e2e-drone --steps 20 > $OUT/bench_e2e_drone.json 2>> $OUT/bench.err
bash tests/sizes.sh > $OUT/sizes.txt 2>&1
python3 bench.py --workload static --steps 30 --warmup 10 --debayer malvar2004 --no-cpu-baseline > $OUT/bench_static_malvar.json 2>> $OUT/bench.err
python3 tests/bench_static.py > $OUT/static.txt 2>&1
python3 tests/bench_aux.py > $OUT/aux.txt 2>&1
if [ -f tests/_build/lib_stamps.so ]; then R2L_STAMPS_KEEP_LUMA=1 R2L_STAMPS_DETAIL=1 python3 tests/stamps.py > $OUT/stamps.txt 2>&1; fi
ls -la $OUT
if [ -f tests/_build/lib_tl.so ]; then R2L_STAMPS_LIB=lib_tl.so python3 tests/timeline_fwd.py > $OUT/timeline_fwd.txt 2>&1; SHAPE=64x256x256 R2L_STAMPS_LIB=lib_tl.so python3 tests/timeline_fwd.py >> $OUT/timeline_fwd.txt 2>&1; fi
python3 tests/bench_epilogue.py > $OUT/epilogue.txt 2>&1
