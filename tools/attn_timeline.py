#!/usr/bin/env python
"""Segment timeline of the ping-pong attention kernel (TIMING build, FW_ATTN_VAR=66): s_memtime at the segment boundaries of
work-group 0, KV tile 100, one line per wave.  ticks -> shader cycles (s_memtime counts at a fixed 100 MHz on gfx950, so the
numbers are scaled by the measured kernel clock if REFCLK is set)."""
import ctypes
import sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fantasy_world_amd.hip_ops import HipOps

ops = HipOps("cuda:0")
for (H, hd, L) in [(40, 128, 32760), (12, 96, 32760), (16, 64, 32865)]:
    g = torch.Generator(device="cuda").manual_seed(0)
    q = (torch.randn(L, H * hd, device="cuda", generator=g) * ops.q_scale(hd)).to(torch.bfloat16)
    k = torch.randn(L, H * hd, device="cuda", generator=g).to(torch.bfloat16)
    v = torch.randn(L, H * hd, device="cuda", generator=g).to(torch.bfloat16)
    ops.set_option("attn_var", 66)
    for _ in range(2):
        ops.attention(q, k, v, H, hd, q_prescaled=True)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 64)()
    rc = ops.lib.fw_debug_attention_timestamps(ctypes.cast(buf, ctypes.c_void_p), 64)
    assert rc == 0, rc
    t0 = min(buf[w * 8] for w in range(8))
    print(f"hd={hd}: ticks relative to the earliest V-segment start; columns: V start | V end | wait end | barrier passed (MM start) | last MFMA issued | wait end | barrier passed")
    for w in range(8):
        print(f"  wave {w} (group {'A' if w < 4 else 'B'}): " + " ".join(f"{buf[w*8+i]-t0:7d}" for i in range(7)))
ops.set_option("attn_var", 64)
