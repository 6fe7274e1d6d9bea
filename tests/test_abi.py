"""C-ABI checks that need no GPU: the library builds, loads, and exports every symbol include/fw_mi355x.h declares;
the product path refuses to run without a GPU instead of falling back."""
import os
import re

import pytest
import torch

from conftest import ROOT
from fantasy_world_amd import hip_ops


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "fw_mi355x.h")).read()
    return sorted(set(re.findall(r"^\s*(?:int|const char\*)\s+(fw_[a-z0-9_]+)\s*\(", txt, flags=re.M)))


def test_library_exports_every_declared_symbol():
    lib = hip_ops.load_library()
    declared = _declared_symbols()
    assert declared, "header parse failed"
    assert sorted(hip_ops.SYMBOLS) == declared, (sorted(hip_ops.SYMBOLS), declared)
    for sym in declared:
        assert getattr(lib, sym) is not None
    assert lib.fw_abi_version() == 5


def test_no_cpu_fallback():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        hip_ops.HipOps("cuda")


def test_product_does_not_import_oracle():
    """Nothing under fantasy-world_amd/ may import oracle/ (the oracle is test infrastructure)."""
    pkg = os.path.join(ROOT, "fantasy-world_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(".py"):
                src = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), fn
