#!/usr/bin/env python
"""Interleaved A/B of the fp8 attention kernels (fw_attention_fp8, hd 128) in ONE process, random data, medians -- the steps from
round 4's kernel to round 6's default, each arm adding one thing (csrc/attention_fp8.hip lists them at fw_attention_fp8):
    FW_ATTN_VAR 9   two-group ping-pong kernel (rounds 2-4 default)
    13 / 12         round 5: single-stream kernel, row sums by a ones-MFMA, P = e4m3(exp2(.)) by v_exp_f32 + v_cvt_pk_fp8_f32; all eight waves
                    in phase / the two waves of a SIMD half a tile apart (round 5's default)
    14              + P's e4m3 byte = round(8 log2 P + 56) by ONE v_cvt_pk_u8_f32 per score (no transcendental in the loop)
    11              + the tile body in two basic blocks (one overflow test, no branch around the barrier, in phase), row sums through a
                    temporary accumulator, the shift updated register by register (no per-tile copies)
    15              + the tile requests between the PV MFMAs
    16              + one barrier per two tiles
    17              + the steady loop unrolled by the ring depth (slot offsets are immediates)
    default         + the row sums by a 16x16x128 MFMA with a per-lane ones operand (half the matrix time of the 32x32x64 one)
and the bf16 kernel on the same shape as the yardstick.  Shapes: the DiT self-attention of BASELINE configs[1] (L = 32 760) and
configs[4] (L = 111 600), 8 of the 40 heads (same work per work-group).  -> stdout (tools/gpu_pass.sh run: stage logs it)."""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fantasy_world_amd.hip_ops import HipOps
ops = HipOps("cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)
mk = lambda *s: torch.randn(*s, device="cuda", generator=g).to(torch.bfloat16)
H, hd = 8, 128
ROUNDS, ITERS = int(os.environ.get("ROUNDS", 5)), int(os.environ.get("ITERS", 3))


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(ITERS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / ITERS


for L in (32760, 111600):
    q, k, v = mk(L, H * hd), mk(L, H * hd), mk(L, H * hd)
    q8 = ops.cast_fp8((q.float() * ops.q_scale_fp8(hd)).to(torch.bfloat16))
    k8 = ops.cast_fp8(k)
    vt8, lk = ops.prepare_v_fp8(v, H, hd)
    qs = (q.float() * ops.q_scale(hd)).to(torch.bfloat16)
    vt = ops.prepare_v(v, H, hd)
    arms = {"9  ping-pong (rounds 2-4)": 9, "13 round 5, in phase": 13, "12 round 5, half-tile skew (its default)": 12,
            "14 + linear-byte probabilities": 14, "11 + two-block tile, in phase": 11, "15 + requests between the PV MFMAs": 15,
            "16 + one barrier per two tiles": 16, "17 + unrolled by the ring depth": 17,
            "default: + row sums by a 16x16x128 MFMA": 192}
    for extra in os.environ.get("EXTRA_VARS", "").split(","):
        if extra:
            arms[f"fp8 experiment var {extra}"] = int(extra)
    outs, times = {}, {n: [] for n in list(arms) + ["bf16 kernel"]}
    for name, var in arms.items():
        ops.set_option("attn_var", var)
        outs[name] = ops.attention_fp8(q8, k8, vt8, H, hd, lk).float()
    ops.set_option("attn_var", 192)
    ref = ops.attention(qs, k, v, H, hd, q_prescaled=True, v_prepared=vt).float()
    for r in range(ROUNDS):
        for name, var in arms.items():
            ops.set_option("attn_var", var)
            times[name].append(timed(lambda: ops.attention_fp8(q8, k8, vt8, H, hd, lk)))
        ops.set_option("attn_var", 192)
        times["bf16 kernel"].append(timed(lambda: ops.attention(qs, k, v, H, hd, q_prescaled=True, v_prepared=vt)))
    fl = 4.0 * L * L * H * hd
    print(f"# hd 128, {H} heads, L = {L}: {fl / 1e12:.2f} TFLOP per launch; medians of {ROUNDS} interleaved rounds x {ITERS} launches")
    for name, ts in times.items():
        ms = statistics.median(ts)
        rel = "" if name == "bf16 kernel" else f"   rel-L2 vs bf16 kernel {((outs[name] - ref).norm() / ref.norm()).item():.3e}"
        print(f"  {name:48s} {ms:9.3f} ms  {fl / ms / 1e9:8.1f} TF/s  ({fl / ms / 1e9 / 5000:.3f} of the 5 PF fp8 peak, {fl / ms / 1e9 / 2500:.3f} of 2.5 PF){rel}")
ops.set_option("attn_var", 192)
