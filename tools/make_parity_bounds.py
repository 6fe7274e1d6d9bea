#!/usr/bin/env python
"""tests/golden/parity_bounds_gpu.json = a margin x the errors measured on MI355X (profiles/rNN/parity.json, written by the GPU test
session through conftest.ParityLog).  Usage: tools/make_parity_bounds.py profiles/r04/parity.json

Margins: 1.5 x for comparisons whose BOTH sides are deterministic and ours to pin (HIP kernels against CPU-generated goldens or against
each other: a measured value reproduces bit for bit on any gfx950); 2.5 x where the checker is the real reference running on PyTorch-ROCm
kernels on the box (`ref_on_gpu/`, `full_depth/`, `full_depth_fp8/`, `config5/`): the vendor library's algorithm selection may move their side a little
between boxes and versions -- 2.5 x absorbs that and still bites (round 3 left these at the physical 4e-2 / 6e-2 only, where the
latents at 9.7e-3 could have regressed 4x unnoticed)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
rows = json.load(open(src))
LOOSE = ("ref_on_gpu/", "full_depth/", "full_depth_fp8/", "config5/")
bounds = {}
for name, row in rows.items():
    if isinstance(row, dict) and "measured" in row:
        margin = 2.5 if name.startswith(LOOSE) else 1.5
        # a floor keeps exact (0.0) or near-exact comparisons from becoming `< 0`
        bounds[name] = max(margin * row["measured"], 1e-7)
out = {"source": os.path.relpath(src, ROOT),
       "rule": "bound = max(margin * measured, 1e-7), margin 1.5 (2.5 where the checker runs on PyTorch-ROCm kernels: ref_on_gpu/, "
               "full_depth/, full_depth_fp8/, config5/); enforced = min(test's physical bound, this)",
       "bounds": dict(sorted(bounds.items()))}
path = os.path.join(ROOT, "tests", "golden", "parity_bounds_gpu.json")
with open(path, "w") as f:
    json.dump(out, f, indent=1)
print(f"{len(bounds)} bounds -> {path}")
