#!/bin/bash
# tile forward under several grid sizes (GRIDS="256 512") and named builds (tests/_build/lib_<name>.so): round-1 A/B
cd "$(dirname "$0")/../.."
for v in default "$@"; do
  if [ "$v" = default ]; then unset R2L_LIB_PATH; else export R2L_LIB_PATH=$PWD/tests/_build/lib_$v.so; fi
  for g in $GRIDS; do
  export R2L_GRID_FWD=$g
  python - <<'PY'
import os, torch, ctypes
from oracle import isp_oracle as orc
from raw2logit_amd import _lib
from raw2logit_amd.processing.pipeline_torch import ParametrizedProcessing
lib=_lib.device_library()
raw=torch.from_numpy(orc.synth_raw(64,512,512,seed=0)).cuda()
m=ParametrizedProcessing(orc.DRONE_CAMERA_PARAMS, batch_norm_output=False).cuda()
with torch.no_grad():
    for _ in range(3): m(raw)
    torch.cuda.synchronize(); lib.r2l_timing_enable(1)
    for _ in range(20): m(raw)
    torch.cuda.synchronize()
buf=ctypes.create_string_buffer(1<<16); lib.r2l_timing_report(buf,len(buf)); lib.r2l_timing_enable(0)
for l in buf.value.decode().splitlines():
    n,c,ms=l.split()
    if 'fwd' in n: print(os.environ.get('R2L_LIB_PATH','default').split('/')[-1], 'grid', os.environ.get('R2L_GRID_FWD'), n, '%.1f us'%(1e3*float(ms)/int(c)))
PY
  done
done
