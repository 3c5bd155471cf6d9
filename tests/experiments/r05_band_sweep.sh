#!/bin/bash
# band heights of the step's passes re-swept under the round's final cache policy (diagnostic build; tests/lib_ab.py: HIP events per kernel, us)
cd "$(dirname "$0")/../.."
run() { echo "$1"; env $1 REPS=1 python tests/lib_ab.py libr2l_isp_hooks.so 2>&1 | grep -v amdgpu | sed 's/libr2l_isp_hooks.so *//'; }
run "R2L_NOOP=1"
for b in 12 18 30 36 48; do run "R2L_FA_BAND=$b"; done
for b in 24 30 42 48; do run "R2L_BP_BAND=$b"; done
for b in 12 18 24 30 48; do run "R2L_HB_BAND=$b"; done
for b in 18 24 30 42 48; do run "R2L_B2S_BAND=$b"; done
for b in 16 20 26 32; do run "R2L_FS_BAND=$b"; done
