#!/usr/bin/env python
"""PREDICTION (not a measurement) of the 1 -> 8 GPU curve of the headline workload for both partitions, communication included --
VERDICT r04 missing 4: the first hardware run needs something to be compared against.

    compute   per-forward compute of one rank at its exact shard shapes, MEASURED on one MI355X with the collectives stubbed
              (tools/shard_emulation.py, profiles/r04/bench_lines_and_shard_emulation_call13.txt)
    bytes     what each GPU sends per forward, COUNTED from the engine's exchanges (same formulas as CommStats reports at run time)
    links     xGMI is point-to-point: 7 links per GPU, ~153 GB/s per link (task statement / SURVEY.md 8(e)); RCCL payload efficiency
              is unknown here, so two rates are carried: 153 GB/s per link and direction (optimistic) and 76.8 GB/s (one direction of a
              153.6 GB/s bidirectional link, conservative).  Inside a group of n ranks every peer pair has its own link, so an
              all-to-all moves bytes/(n-1) per link in parallel and a direct reduce-scatter + all-gather all-reduce 2*payload/n per link.
    overlap   sequence shard: the q|k|v exchange of a block flies behind the other branch's work (IRG blocks) or behind the other
              head group's attention (16 preconditioning blocks, 2 groups): between 50 % and 100 % of the exchange time is hidden;
              tensor parallel: the reduction of one row block flies behind the GEMM of the next (2 row blocks): 0-50 % hidden.
Prints one row per (N, partition): step time range and speed-up over the measured 1-GPU step.
"""
L, L2, D, Fd, C, Cm, Bd = 32760, 32865, 5120, 13824, 1024, 4096, 1152
N_DIT, N_IRG, N_BI, N_AD = 40, 24, 24, 25
ONE_GPU_STEP_MS = 3505.5            # profiles/r05/bench_call2.log (two forwards, cache off)
COMPUTE_MS = {("sp", 1): 1728.5, ("sp", 2): 918.2, ("sp", 4): 492.0, ("tp", 2): 977.0, ("tp", 4): 588.0, ("tp", 8): 408.8}
LINK = {"optimistic": 153e9, "conservative": 76.8e9}


def sp_bytes_sent(n):
    """Sequence shard with head exchange: all-to-all of q|k|v (bf16) in, attention output back; per GPU per forward."""
    f = (n - 1) / n
    dit = N_DIT * (L / n) * (3 * D + D) * 2 * f
    vggt = N_IRG * (L2 / n) * (3 * C + C) * 2 * f
    bi = N_BI * ((L / n) * (2 * Bd) + (L2 / n) * (2 * Bd) + (L / n) * Bd + (L2 / n) * Bd) * 2 * f if 12 % n == 0 else \
        N_BI * (L2 * 2 * Bd + L * 2 * Bd) * 2 * f                   # K/V row all-gather fallback
    return dit + vggt + bi


def tp_payload(reduce_bytes):
    """Tensor parallel: all-reduced partial sums per forward (o, cross o, ffn2 per DiT block; proj, fc2 per VGGT block x2 kinds; bicross
    out projections; camera adapter 5120->1024)."""
    dit = N_DIT * 3 * L * D
    ad = N_AD * L * 1024
    vggt = 2 * N_IRG * 2 * L2 * C
    bi = N_BI * (L * D + L2 * C)
    return (dit + ad + vggt + bi) * reduce_bytes


def main():
    print(f"{'N':>2s} {'partition':52s} {'compute ms':>10s} {'GB sent/GPU':>11s} {'comm ms (opt..cons)':>20s} {'step ms (best..worst)':>22s} {'speed-up':>12s}")
    rows = [(2, "CFG x2 (one forward per GPU)", ("sp", 1), 0.0, None)]
    for N, n in ((4, 2), (8, 4)):
        rows.append((N, f"CFG x2 x sequence shard {n} (default)", ("sp", n), sp_bytes_sent(n), "sp"))
        for name, rb in (("fp32 sums (default)", 4), ("bf16 sums", 2)):
            rows.append((N, f"CFG x2 x tensor parallel {n}, {name}", ("tp", n), tp_payload(rb), "tp"))
    for N, label, key, b, kind in rows:
        comp = COMPUTE_MS[key]
        n = key[1]
        if kind is None:
            lo = hi = comp + 0.1
            sent, c_opt, c_con = 0.0, 0.0, 0.0
        elif kind == "sp":
            sent = b
            per_link = b / (n - 1)
            c_opt, c_con = 1e3 * per_link / LINK["optimistic"], 1e3 * per_link / LINK["conservative"]
            lo, hi = comp + 0.0 * c_opt, comp + 0.5 * c_con
        else:
            sent = 2 * b * (n - 1) / n
            per_link = 2 * b / n
            c_opt, c_con = 1e3 * per_link / LINK["optimistic"], 1e3 * per_link / LINK["conservative"]
            lo, hi = comp + 0.5 * c_opt, comp + 1.0 * c_con
        print(f"{N:2d} {label:52s} {comp:10.0f} {sent / 1e9:11.2f} {c_opt:9.0f} .. {c_con:6.0f} {lo:12.0f} .. {hi:6.0f} "
              f"{ONE_GPU_STEP_MS / hi:5.2f} .. {ONE_GPU_STEP_MS / lo:4.2f}x")
    print("\n(1 GPU measured: %.0f ms per step.  north_star's target: >= 6x at 8 GPUs.)" % ONE_GPU_STEP_MS)


if __name__ == "__main__":
    main()
