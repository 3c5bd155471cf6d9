#!/bin/bash
# round 6: (a) the COLD regime (scrub between forward and backward): which BatchNorm-backward-sums kernel, which band heights;
# (b) 64 / 128 x 256^2: band heights of every pass (hooks library, overrides read per launch)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=gpurun_out/r06_cold_small
mkdir -p $OUT
export R2L_LIB_PATH=$PWD/tests/_build/libr2l_isp_hooks.so
kern() { python3 - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print('no line', e); sys.exit(0)
k = {n.replace('r2l_launch_', '').replace('_kernel', ''): v['avg_us'] for n, v in d.get('kernels', {}).items()}
c = d.get('cold') or {}
print('ms/step %.4f  kernels %s' % (d['ms_per_step'], ' '.join('%s=%.1f' % kv for kv in sorted(k.items()))))
if c:
    print('   COLD kernels %.4f ms: %s' % (c['ms_per_step_kernels'], ' '.join('%s=%.1f' % kv for kv in sorted(c['kernels'].items()))))
PY
}
run() { tag=$1; shift; env "$@" python3 bench.py --quick --steps 20 --warmup 5 $ARGS > $OUT/$tag.json 2> $OUT/$tag.err; echo "== $tag ($*)"; kern $OUT/$tag.json; }
{
ARGS="--cold"
run cold_default X=1
run cold_bn_reduce R2L_BNR_READ_OUT=1
for b in 12 18 24 30 48; do run cold_bnr_band$b R2L_BNR_BAND=$b; done
for b in 12 24 48; do run cold_bp_band$b R2L_BP_BAND=$b; done
for b in 12 36 48; do run cold_fa_band$b R2L_FA_BAND=$b; done
ARGS="--batch 64 --size 256"
run s64_default X=1
for b in 12 18; do run s64_all$b R2L_FL_BAND=$b R2L_FST_BAND=$b R2L_FA_BAND=$b R2L_BP_BAND=$b R2L_HB_BAND=$b R2L_B2S_BAND=$b; done
for v in FL FST FA BP HB B2S; do run s64_${v}12 R2L_${v}_BAND=12; done
run s64_planes_bnr R2L_BWD_PLANES=1
run s64_tiled R2L_BWD1_TILED=1 R2L_BWD2_TILED=1
ARGS="--batch 128 --size 256"
run s128_default X=1
for v in FL FST FA BP HB B2S BNR; do run s128_${v}12 R2L_${v}_BAND=12; done
} > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
