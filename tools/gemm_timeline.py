#!/usr/bin/env python
"""Phase timeline of the bf16 ping-pong GEMM (TIMING build, FW_GEMM_KERNEL=4 var bit 1): s_memtime at the start (barrier passed)
and at the end of the work of each of the four phases of slabs 16..19, work-group 0, one wave of each group."""
import ctypes
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fantasy_world_amd.hip_ops import HipOps, Linear

ops = HipOps("cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)
M, N, K = 32760, 5120, 5120
x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
lin = Linear(torch.randn(N, K, device="cuda", generator=g).to(torch.bfloat16) * K ** -0.5, torch.zeros(N, device="cuda"))
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
for var, with_res in ((2, False), (4, False), (4, True)):
    ops.set_option("gemm_kernel", 4)
    ops.set_option("gemm_var", var)
    xs = torch.randn(M, N, device="cuda", generator=g) if with_res else None
    gate = torch.randn(N, device="cuda", generator=g) if with_res else None
    run = (lambda: ops.linear(x, lin, g1=gate, res=xs, out_f32=True, out=xs)) if with_res else (lambda: ops.linear(x, lin, out=out))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        run()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    buf = (ctypes.c_ulonglong * 128)()
    assert ops.lib.fw_debug_gemm_timestamps(ctypes.cast(buf, ctypes.c_void_p), 128) == 0
    t0 = min(buf[0], buf[64])
    print(f"== FW_GEMM_KERNEL=4 var {var} ({'phase + tile stamps' if var == 2 else 'tile stamps only'}{', gate + fp32 residual in place' if with_res else ''}): {ms:.3f} ms, {2.0*M*N*K/ms/1e9:.0f} TF/s; "
          f"kernel = 10 rounds x 80 slabs -> {ms*1e6/800:.0f} ns per slab on average")
    names = ["LOAD0", "MFMA0", "LOAD1", "MFMA1"]
    for grp in range(2 if var == 2 else 0):
        print(f"  group {'AB'[grp]} (ticks since first stamp; per phase: start | end-of-work | -> next start = barrier wait)")
        for sl in range(4):
            row = []
            for ph in range(4):
                s_, e_ = buf[grp * 64 + sl * 8 + ph * 2] - t0, buf[grp * 64 + sl * 8 + ph * 2 + 1] - t0
                row.append(f"{names[ph]} {s_:6d}+{e_ - s_:4d}")
            print("    slab", 16 + sl, " | ".join(row))
    if var == 2:
        span = buf[3 * 8 + 6] - buf[0]
        print(f"  4 slabs (group A LOAD0(16) start -> MFMA1(19) start): {span} ticks")
    # ---- per-tile stamps (s_memrealtime, 100 MHz): what a tile costs outside its mainloop, and the gap between tiles on one CU
    import statistics
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    big = (ctypes.c_ulonglong * (128 + 6 * tiles))()
    assert ops.lib.fw_debug_gemm_timestamps(ctypes.cast(big, ctypes.c_void_p), 128 + 6 * tiles) == 0
    rows = [[big[128 + 6 * i + j] for j in range(6)] for i in range(tiles)]
    us = lambda a, b: (b - a) / 100.0
    pro = [us(r[0], r[1]) for r in rows]
    main = [us(r[1], r[2]) for r in rows]
    epi = [us(r[2], r[3]) for r in rows]
    drain = [us(r[3], r[4]) for r in rows]
    med = statistics.median
    print(f"  per tile ({tiles} tiles, median us): entry -> prologue landed {med(pro):.2f} | mainloop {med(main):.2f} | epilogue (stores issued) "
          f"{med(epi):.2f} | store drain {med(drain):.2f} | total {med([us(r[0], r[4]) for r in rows]):.2f}")
    t_start, t_end = min(r[0] for r in rows), max(r[4] for r in rows)
    starts = sorted(us(t_start, r[0]) for r in rows)
    rounds = (tiles + 255) // 256
    print(f"  kernel span {us(t_start, t_end):.1f} us for {rounds} rounds of 256 tiles = {us(t_start, t_end)/rounds:.2f} us per round; "
          f"tile starts at (us, every 256th in start order): {[round(starts[k], 1) for k in range(0, tiles, 256)]}")
ops.set_option("gemm_kernel", 9)
ops.set_option("gemm_var", 0)
