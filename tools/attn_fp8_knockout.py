#!/usr/bin/env python
"""TIMING-ONLY knock-out arms of the fp8 attention kernel (results are wrong by construction): what each part of the tile costs where
it sits.  Needs a library built with the arms:
    FW_BUILD_TAG=knock FW_MFMA_EXTRA_FLAGS=-DFW8_KNOCKOUTS bash fantasy_world_amd/csrc/build.sh          # here, no GPU needed
    FW_LIB_PATH=fantasy_world_amd/libfw_mi355x.knock.so python tools/attn_fp8_knockout.py                # on the box
Arms (FW_ATTN_VAR = 1000 + bits, on the default kernel): 8 no s_barrier, 16 no row maxima, 32 no conversions (P = raw score bits), 64 no
ones-MFMA, 256 no tile requests after the prologue.  (A "no fragment reads" arm is not offered: without them the score MFMAs are loop
invariant and the compiler hoists them.  Without the row maxima the probabilities are NaN bytes, which the matrix pipe runs SLOWER.)"""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fantasy_world_amd.hip_ops import HipOps
ops = HipOps("cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)
mk = lambda *s: torch.randn(*s, device="cuda", generator=g).to(torch.bfloat16)
H, hd, L = 8, 128, 32760
ROUNDS, ITERS = int(os.environ.get("ROUNDS", 5)), int(os.environ.get("ITERS", 3))
NAMES = {8: "barrier", 16: "row maxima", 32: "conversions", 64: "ones-MFMA", 256: "tile requests"}
ARMS = [0, 8, 16, 32, 64, 256, 8 + 256, 16 + 32, 16 + 32 + 64, 8 + 16 + 32 + 64]
q, k, v = mk(L, H * hd), mk(L, H * hd), mk(L, H * hd)
q8 = ops.cast_fp8((q.float() * ops.q_scale_fp8(hd)).to(torch.bfloat16)); k8 = ops.cast_fp8(k)
vt8, lk = ops.prepare_v_fp8(v, H, hd)


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(ITERS): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / ITERS


times = {a: [] for a in ARMS}
for a in ARMS:
    ops.set_option("attn_var", 1000 + a if a else 192); ops.attention_fp8(q8, k8, vt8, H, hd, lk)
for r in range(ROUNDS):
    for a in ARMS:
        ops.set_option("attn_var", 1000 + a if a else 192)
        times[a].append(timed(lambda: ops.attention_fp8(q8, k8, vt8, H, hd, lk)))
ops.set_option("attn_var", 192)
fl = 4.0 * L * L * H * hd
base = statistics.median(times[0])
print(f"# fp8 attention, hd 128, {H} heads, L = {L}; 9 MFMAs of 64 cycles per tile and wave, two waves per SIMD: matrix floor at 2.4 GHz = {fl / 5e15 * 1e3:.3f} ms")
for a in ARMS:
    ms = statistics.median(times[a])
    what = "the kernel" if not a else "without " + ", ".join(n for b, n in NAMES.items() if a & b)
    print(f"  {what:100s} {ms:7.3f} ms  {100 * (ms / base - 1):+6.1f} %   {fl / ms / 1e9:7.0f} 'TF/s'")
