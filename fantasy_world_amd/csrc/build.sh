#!/bin/bash
# Builds libfw_mi355x.so (gfx950 only) next to the package.  hipcc cross-compiles without a GPU.
# Every compile runs in the background; each one's exit status is checked (a failed compile removes its stale object and fails the
# build -- round 4: `wait` without pids had let a source that no longer compiled link silently against its previous object), and the
# linked library must define the host stub of every kernel it references (an undefined __device_stub__ = the library cannot load).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="${here}/../libfw_mi355x.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
flags=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -Wno-unused-result)
objs=()
pids=()
names=()
for f in gemm.hip gemm_pp.hip attention.hip elementwise.hip heads.hip fp8.hip gemm_fp8.hip attention_fp8.hip; do
  o="${here}/${f%.hip}.o"
  if [ ! -f "$o" ] || [ "${here}/$f" -nt "$o" ] || [ "${here}/fw_common.h" -nt "$o" ] || [ "${here}/gemm_common.h" -nt "$o" ] || [ "${here}/../../include/fw_mi355x.h" -nt "$o" ]; then
    rm -f "$o"
    if [ "$f" = "fp8.hip" ]; then   # IEEE division for the fp8 quantiser: no -ffast-math
      "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -c "${here}/$f" -o "$o" &
    elif [ "$f" = "attention.hip" ]; then
      # no SLP vectorisation: hipcc otherwise packs the softmax row sums into v_pk_add_f32 (an anti-lever beside MFMAs,
      # MI355X_MICROARCH.md) and, in the ring-unrolled kernels, spills (81 scratch accesses per four tiles at hd 128)
      "$HIPCC" "${flags[@]}" ${FW_ATTN_EXTRA_FLAGS--fno-slp-vectorize} -c "${here}/$f" -o "$o" &
    else
      "$HIPCC" "${flags[@]}" -c "${here}/$f" -o "$o" &
    fi
    pids+=($!); names+=("$f")
  fi
  objs+=("$o")
done
o="${here}/api.o"
"$HIPCC" "${flags[@]}" -x hip -c "${here}/api.cpp" -o "$o" &
pids+=($!); names+=("api.cpp")
objs+=("$o")
fail=0
for i in "${!pids[@]}"; do
  if ! wait "${pids[$i]}"; then echo "build.sh: compiling ${names[$i]} FAILED" >&2; fail=1; fi
done
[ "$fail" = 0 ] || exit 1
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$out"
if nm -D --undefined-only "$out" | grep -q "__device_stub__"; then
  echo "build.sh: $out references kernel host stubs it does not define:" >&2
  nm -DC --undefined-only "$out" | grep "__device_stub__" | head -5 >&2
  exit 1
fi
echo "built $out"
