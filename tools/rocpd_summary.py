#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (ROCm 7.2 default output) as text: per-kernel (name, grid) call count,
average / total duration and share -- the `--kernel-trace --stats` view -- plus per-kernel PMC counter averages when
the run collected counters.  Usage: tools/rocpd_summary.py <results.db> [--filter substr] > profiles/rNN/xxx.txt"""
import argparse
import re
import sqlite3
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:<>, ]+?)\(", name)
    s = m.group(1) if m else name
    return s[:70]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--filter", default="")
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--split", action="store_true", help="split a (kernel, grid) group into 4x duration buckets")
    args = ap.parse_args()
    con = sqlite3.connect(args.db)
    cur = con.cursor()
    rows = cur.execute("select name, grid_x, workgroup_x, duration, vgpr_count, accum_vgpr_count, lds_size from kernels").fetchall()
    agg = defaultdict(lambda: [0, 0.0, 0, 0, 0])
    total = 0.0
    for name, gx, wx, dur, vg, ag, lds in rows:
        import math
        k = (short(name), gx // max(wx, 1), wx, int(math.log2(max(dur, 1) / 1e3 + 1e-9) // 2) if args.split else 0)
        a = agg[k]
        a[0] += 1
        a[1] += dur
        a[2], a[3], a[4] = vg, ag, lds
        total += dur
    print(f"# {args.db}: {len(rows)} dispatches, total kernel time {total/1e6:.3f} ms")
    print(f"{'kernel':70s} {'wgs':>8s} {'wgsz':>5s} {'calls':>6s} {'avg_us':>11s} {'total_ms':>10s} {'share%':>7s} {'vgpr':>5s} {'lds':>6s}")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[: args.top]:
        if args.filter and args.filter not in k[0]:
            continue
        print(f"{k[0]:70s} {k[1]:8d} {k[2]:5d} {a[0]:6d} {a[1]/a[0]/1e3:11.2f} {a[1]/1e6:10.3f} {100*a[1]/total:7.2f} {a[2]+a[3]:5d} {a[4]:6d}")
    try:
        crow = cur.execute("select kernel_name, grid_size, workgroup_size, counter_name, value from counters_collection").fetchall()
    except sqlite3.Error:
        crow = []
    if crow:
        cagg = defaultdict(lambda: [0, 0.0])
        for name, gs, ws, cname, val in crow:
            k = (short(name), gs // max(ws, 1), cname)
            cagg[k][0] += 1
            cagg[k][1] += val
        print("\n# PMC counters: average per dispatch (summed over the counter's dimensions)")
        print(f"{'kernel':70s} {'wgs':>8s} {'counter':28s} {'dispatch-rows':>13s} {'avg_value':>18s}")
        for k, a in sorted(cagg.items()):
            if args.filter and args.filter not in k[0]:
                continue
            if k[0].startswith("at::") or k[0].startswith("void at::"):
                continue
            print(f"{k[0]:70s} {k[1]:8d} {k[2]:28s} {a[0]:13d} {a[1]/a[0]:18.1f}")


if __name__ == "__main__":
    main()
