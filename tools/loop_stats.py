#!/usr/bin/env python
"""Instruction mix of the hot loops of one kernel, from the gfx950 ISA of a built csrc/*.o (no GPU needed):
    tools/loop_stats.py attention 'attention_pp3_kernel<96, 0>' [--list LO HI]
For every backward branch whose body holds at least 10 MFMAs: instruction count and the split into mfma / valu / s_ (scalar,
waits, barriers) / branch / ds (LDS) / vmem.  --list prints the instructions LO..HI of the kernel (indices as printed)."""
import collections, os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
obj, want = sys.argv[1], sys.argv[2]
dis = f"/tmp/{obj}.dis"
subprocess.check_call(["bash", os.path.join(root, "tools", "disasm.sh"), obj, dis], stdout=subprocess.DEVNULL)
txt = open(dis).read()
def klass(k):
    if "mfma" in k: return "mfma"
    if k.startswith(("s_cbranch", "s_branch")): return "branch"
    if k.startswith("s_"): return "s_"
    if k.startswith("ds_"): return "ds"
    if k.startswith(("buffer", "global", "flat")): return "vmem"
    return "valu"
for f in re.split(r"\n(?=[0-9a-f]+ <)", txt):
    m = re.match(r"[0-9a-f]+ <(.*)>:", f)
    if not m or want not in m.group(1): continue
    ops = []
    for l in f.split("\n")[1:]:
        t = l.split("//")[0].strip()
        mm = re.search(r"//\s*([0-9A-Fa-f]+):", l)
        if t and mm: ops.append((int(mm.group(1), 16), t))
    print(m.group(1).replace("(anonymous namespace)::", "")[:90], "-", len(ops), "instructions")
    if "--list" in sys.argv:
        lo, hi = int(sys.argv[sys.argv.index("--list") + 1]), int(sys.argv[sys.argv.index("--list") + 2])
        for i in range(lo, hi + 1): print(i, ops[i][1])
        continue
    index = {a: k for k, (a, _) in enumerate(ops)}
    for i, (a, t) in enumerate(ops):
        m2 = re.match(r"(s_cbranch_\w+|s_branch)\s+(\d+)", t)
        if not m2: continue
        off = int(m2.group(2)); off = off - 65536 if off > 32767 else off
        tgt = a + 4 + off * 4
        if tgt > a or tgt not in index: continue
        j = index[tgt]
        c = collections.Counter(x.split()[0] for _, x in ops[j:i + 1])
        if sum(v for k, v in c.items() if "mfma" in k) < 10: continue
        cls = collections.Counter()
        for k, v in c.items(): cls[klass(k)] += v
        print(f"  loop {j}..{i} ({i - j + 1} instructions)", dict(cls))
        print("    scalar:", sorted(((k, v) for k, v in c.items() if klass(k) in ("s_", "branch")), key=lambda kv: -kv[1])[:12])
