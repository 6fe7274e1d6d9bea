#!/bin/bash
# Builds libfw_mi355x.so (gfx950 only) next to the package.  hipcc cross-compiles without a GPU.
# Every compile runs in the background; each one's exit status is checked (a failed compile removes its stale object and fails the
# build -- round 4: `wait` without pids had let a source that no longer compiled link silently against its previous object), and the
# linked library must define the host stub of every kernel it references (an undefined __device_stub__ = the library cannot load).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
# FW_BUILD_TAG=<tag> (A/B builds of the whole library with other flags, tools/lib_ab.py): objects csrc/<name>.<tag>.o, output
# libfw_mi355x.<tag>.so next to the default one (never loaded unless FW_LIB_PATH points at it).  FW_MFMA_EXTRA_FLAGS: appended to the
# flags of the MFMA files (attention*.hip, gemm*.hip), e.g. "-fno-associative-math".
tag="${FW_BUILD_TAG:+.${FW_BUILD_TAG}}"
out="${here}/../libfw_mi355x${tag}.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
# -fno-associative-math (round 6): -ffast-math let hipcc re-associate the softmax row sums and the epilogue sums PER
# INSTANTIATION, so "bit-identical" claims held per compiler version only.  Measured in one process against the build without it
# (tools/lib_ab.py, profiles/r06/lib_ab_no_associative_math_call1.txt): -0.6 .. +0.5 % on the four attention kernels and the five GEMM
# shapes (noise), bits moved in the three bf16 attention kernels and the GELU epilogue -- adopted; the other fast-math sub-flags stay
# (approximate exp / division in the epilogues sit beside MFMAs).  tests/golden/kernel_digests_gfx950.json pins the outputs.
flags=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -fno-associative-math -Wno-unused-result ${FW_MFMA_EXTRA_FLAGS-})
objs=()
pids=()
names=()
for f in gemm.hip gemm_pp.hip attention.hip elementwise.hip heads.hip fp8.hip gemm_fp8.hip attention_fp8.hip; do
  o="${here}/${f%.hip}${tag}.o"
  # rebuild decision by CONTENT, not by mtime (round 5: this container's clock stepped backwards by minutes mid-session, so edited
  # sources were "older" than their objects and a renamed kernel silently kept its old object): hash of source + headers + this script
  want="$(cat "${here}/$f" "${here}/fw_common.h" "${here}/gemm_common.h" "${here}/../../include/fw_mi355x.h" "${BASH_SOURCE[0]}" | sha256sum | cut -d' ' -f1) ${FW_ATTN_EXTRA_FLAGS-default} ${FW_MFMA_EXTRA_FLAGS-}"
  if [ ! -f "$o" ] || [ ! -f "$o.sha" ] || [ "$(cat "$o.sha")" != "$want" ]; then
    rm -f "$o" "$o.sha"
    echo "$want" > "$o.sha.new"
    if [ "$f" = "fp8.hip" ] || [ "$f" = "elementwise.hip" ] || [ "$f" = "heads.hip" ]; then
      # no -ffast-math: IEEE division for the fp8 quantiser; and (round 5) the row passes -- LayerNorm / RMSNorm statistics, RoPE,
      # GroupNorm, the head activations -- are HBM-bound, so the flag bought nothing there while it let the compiler re-associate
      # sums and replace divisions / rsqrt by approximations from one hipcc to the next (round 4 had to pin the LayerNorm statistics by
      # hand after a re-association moved bits).  Only the MFMA files keep it (their epilogue transcendentals sit beside MFMAs).
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
o="${here}/api${tag}.o"
"$HIPCC" "${flags[@]}" -x hip -c "${here}/api.cpp" -o "$o" &
pids+=($!); names+=("api.cpp")
objs+=("$o")
fail=0
for i in "${!pids[@]}"; do
  if ! wait "${pids[$i]}"; then echo "build.sh: compiling ${names[$i]} FAILED" >&2; fail=1; rm -f "${here}/${names[$i]%.*}${tag}.o.sha.new"
  elif [ -f "${here}/${names[$i]%.*}${tag}.o.sha.new" ]; then mv "${here}/${names[$i]%.*}${tag}.o.sha.new" "${here}/${names[$i]%.*}${tag}.o.sha"; fi
done
[ "$fail" = 0 ] || exit 1
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$out"
if nm -D --undefined-only "$out" | grep -q "__device_stub__"; then
  echo "build.sh: $out references kernel host stubs it does not define:" >&2
  nm -DC --undefined-only "$out" | grep "__device_stub__" | head -5 >&2
  exit 1
fi
echo "built $out"
