#!/bin/bash
# build named variants of the device library for A/B runs: tests/build_ab.sh name "flags" [name "flags" ...]
set -e
cd "$(dirname "$0")/.."
mkdir -p tests/_build/ab
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  hipcc -O3 -std=c++17 -fno-slp-vectorize --offload-arch=gfx950 -shared -fPIC $flags raw2logit_amd/csrc/r2l_api.hip -o tests/_build/ab/$name.so -ldl &
done
wait
ls -la tests/_build/ab/
