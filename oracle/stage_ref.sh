#!/bin/bash
# TEST INFRASTRUCTURE ONLY.  Stage the UNMODIFIED reference Python tree for the GPU box (VERDICT r02 "next round" item 2).
#
# /root/reference exists only in the build container; the GPU box receives a snapshot of /root/repo.  This script bundles the
# reference's first-party Python files (FantasyWorld/, the two inference scripts and their utils.py; no third-party, no media) into
# oracle/_ref/reference_py.tgz.  oracle/_ref/ is git-ignored (never committed, never part of the product) but not
# gpurun-ignored, so the bundle travels like the built .so files.  On the box oracle/ref_locate.py extracts it into a scratch
# directory and the `-m gpu` tests in tests/test_reference_on_gpu.py import the REAL reference modules from there:
#   * install() + HipOps under the reference's own generate_video loop (model_wan21.py:226-324) on MI355X,
#   * fw_gemm_fp8 against the real torch._scaled_mm call of AutoWrappedLinear.fp8_linear (vram_management/layers.py:115-151),
#   * the Wan2.2 sampler's own loop (inference_wan22.py:164-283, two experts) on top of install(),
#   * bench.py's cpu_baseline leg with kind = "reference" (the reference's own blocks on the box's host cores).
# The product (fantasy_world_amd/) never imports anything under oracle/.
set -e
SRC=${FW_REFERENCE_ROOT:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
if [ ! -d "$SRC/FantasyWorld" ]; then echo "stage_ref: $SRC/FantasyWorld not found (not the build container): nothing staged"; exit 0; fi
mkdir -p "$HERE/_ref"
tar -C "$SRC" --exclude='__pycache__' --exclude='*.pyc' -czf "$HERE/_ref/reference_py.tgz" FantasyWorld inference_wan21.py inference_wan22.py utils.py
echo "stage_ref: $(du -h "$HERE/_ref/reference_py.tgz" | cut -f1) -> $HERE/_ref/reference_py.tgz"
