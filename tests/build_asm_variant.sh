#!/bin/bash
# experiment: build the device library through its assembly, optionally rewriting it (sed script in $2), to measure
# ISA-level changes the compiler gives no switch for.  tests/build_asm_variant.sh <name> [sed-script]  ->  tests/_build/ab/<name>.so
set -e
cd "$(dirname "$0")/.."
LLVM=/opt/rocm/lib/llvm/bin
W=/tmp/r2l_asm_$1; rm -rf $W; mkdir -p $W tests/_build/ab
FLAGS="-O3 -std=c++17 -fno-slp-vectorize --offload-arch=gfx950 -DR2L_TEST_HOOKS"
if [ ! -f /tmp/r2l_asm_dev.s ] || [ raw2logit_amd/csrc/r2l_api_impl.h -nt /tmp/r2l_asm_dev.s ]; then
  hipcc $FLAGS --cuda-device-only -S raw2logit_amd/csrc/r2l_api.hip -o /tmp/r2l_asm_dev.s 2>/dev/null
fi
if [ -n "$2" ]; then sed -E "$2" /tmp/r2l_asm_dev.s > $W/dev.s; else cp /tmp/r2l_asm_dev.s $W/dev.s; fi
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $W/dev.s -o $W/dev.o
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $W/dev.out $W/dev.o
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$W/dev.out -output=$W/dev.hipfb
hipcc $FLAGS --cuda-host-only -fPIC -Xclang -fcuda-include-gpubinary -Xclang $W/dev.hipfb -c raw2logit_amd/csrc/r2l_api.hip -o $W/host.o 2>/dev/null
hipcc -shared $W/host.o -o tests/_build/ab/$1.so -lrocfft
ls -la tests/_build/ab/$1.so
