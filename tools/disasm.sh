#!/bin/bash
# Disassemble the gfx950 code object embedded in a built csrc/*.o:  tools/disasm.sh gemm [out.dis]   (no GPU needed)
LLVM=${ROCM_LLVM_BIN:-/opt/rocm/lib/llvm/bin}
R=$(cd "$(dirname "$0")/.." && pwd)
obj=$R/fantasy_world_amd/csrc/$1.o; out=${2:-/tmp/$1.dis}; td=$(mktemp -d)
$LLVM/llvm-objcopy --dump-section=.hip_fatbin=$td/f.fatbin $obj && \
$LLVM/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$td/f.fatbin --output=$td/k.co && \
$LLVM/llvm-objdump -d --no-show-raw-insn $td/k.co | c++filt > $out; rm -rf $td; echo $out
