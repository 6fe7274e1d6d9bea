#!/usr/bin/env python
"""Per-kernel resources of the built gfx950 code (VGPRs, AGPRs, LDS bytes, scratch bytes), read from the code-object notes of
fantasy_world_amd/csrc/*.o with the ROCm LLVM tools -- no GPU needed.

    python tools/kernel_resources.py [--filter attention_sp] [--scratch-only]

Why it exists: the hot loops live at the 256-VGPR edge (two waves per SIMD); a change that tips hipcc's register allocator over it
shows up as `scratch > 0` and costs 2-3x (round 3: the ring-unrolled attention kernel with SLP vectorisation on, 45.9 ms instead
of 17.1 ms).  tests/test_abi.py::test_hot_kernels_do_not_spill asserts scratch == 0 for the kernels the forward runs at full size."""
import argparse
import glob
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("ROCM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def kernels_of(obj):
    """[(demangled-ish name, vgpr, agpr, lds, scratch)] of one host object with an embedded HIP fat binary."""
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "f.fatbin"), os.path.join(td, "k.co")
        subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj], check=True, capture_output=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--input={fat}", f"--output={co}"], check=True, capture_output=True)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    out, cur = [], {}
    for line in notes.splitlines():
        m = re.match(r"^\s+(?:- )?\.(agpr_count|group_segment_fixed_size|name|private_segment_fixed_size|vgpr_count):\s+(\S+)", line)
        if not m:
            continue
        if m.group(1) == "agpr_count" and cur:
            out.append(cur)
            cur = {}
        cur[m.group(1)] = m.group(2)
    if cur:
        out.append(cur)
    res = []
    for k in out:
        if "name" not in k:
            continue
        try:
            name = subprocess.run(["c++filt", k["name"]], capture_output=True, text=True).stdout.strip() or k["name"]
        except OSError:
            name = k["name"]
        name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
        res.append((name, int(k.get("vgpr_count", 0)), int(k.get("agpr_count", 0)), int(k.get("group_segment_fixed_size", 0)),
                    int(k.get("private_segment_fixed_size", 0))))
    return res


def all_kernels():
    res = []
    for obj in sorted(glob.glob(os.path.join(ROOT, "fantasy_world_amd", "csrc", "*.o"))):
        if os.path.basename(obj).count(".") > 1:
            continue            # csrc/<name>.<tag>.o: an A/B build (FW_BUILD_TAG, csrc/build.sh), not the shipped library
        try:
            res += [(os.path.basename(obj),) + k for k in kernels_of(obj)]
        except subprocess.CalledProcessError:
            continue            # an object without device code (api.o)
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--filter", default="")
    ap.add_argument("--scratch-only", action="store_true")
    a = ap.parse_args()
    print(f"{'object':18s} {'kernel':72s} {'vgpr':>5s} {'agpr':>5s} {'lds':>7s} {'scratch':>8s}")
    for obj, name, vg, ag, lds, scr in all_kernels():
        if a.filter in name and (scr or not a.scratch_only):
            print(f"{obj:18s} {name[:72]:72s} {vg:5d} {ag:5d} {lds:7d} {scr:8d}")
