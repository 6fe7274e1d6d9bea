#!/usr/bin/env python
"""Interleaved A/B of the fp8 attention kernels (fw_attention_fp8, hd 128) in ONE process, random data, medians:
    FW_ATTN_VAR 9   two-group ping-pong kernel (rounds 2-4 default)
    FW_ATTN_VAR 11  single-stream kernel (row sums by a ones-MFMA), all eight waves in phase
    default         the same with the two waves of a SIMD half a tile apart (round 5 default)
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
    arms = {"fp8 ping-pong (var 9)": 9, "fp8 single-stream, in phase (var 11)": 11, "fp8 single-stream, half-tile skew (default)": 192}
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
        print(f"  {name:42s} {ms:9.3f} ms  {fl / ms / 1e9:8.1f} TF/s  ({fl / ms / 1e9 / 5000:.3f} of the 5 PF fp8 peak, {fl / ms / 1e9 / 2500:.3f} of 2.5 PF){rel}")
ops.set_option("attn_var", 192)
