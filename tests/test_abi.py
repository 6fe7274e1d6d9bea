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
    return sorted(set(re.findall(r"^\s*(?:int|int64_t|const char\*)\s+(fw_[a-z0-9_]+)\s*\(", txt, flags=re.M)))


def test_library_exports_every_declared_symbol():
    lib = hip_ops.load_library()
    declared = _declared_symbols()
    assert declared, "header parse failed"
    assert sorted(hip_ops.SYMBOLS) == declared, (sorted(hip_ops.SYMBOLS), declared)
    for sym in declared:
        assert getattr(lib, sym) is not None
    assert lib.fw_abi_version() == 10


def test_no_cpu_fallback():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        hip_ops.HipOps("cuda")


def test_product_does_not_import_oracle():
    """Nothing under fantasy_world_amd/ may import oracle/ (the oracle is test infrastructure)."""
    pkg = os.path.join(ROOT, "fantasy_world_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(".py"):
                src = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), fn


def test_argument_validation_needs_no_gpu():
    """Every entry point validates its arguments before it touches the device: bad calls come back as FW_E_BADARG with a message
    (error behaviour of the boundary, include/fw_mi355x.h), never as a crash -- checked here without a GPU."""
    import ctypes
    lib = hip_ops.load_library()
    buf = (ctypes.c_uint16 * 4096)()
    base = ctypes.addressof(buf)
    base += (-base) % 16
    p, odd = ctypes.c_void_p(base), ctypes.c_void_p(base + 2)

    def bad(rc, needle):
        assert rc != 0, needle
        msg = lib.fw_last_error().decode()
        assert needle in msg, (needle, msg)

    # fw_im2col: C not a multiple of 8; frame window outside the volume; misaligned base
    bad(lib.fw_im2col(p, 12, p, 108, 12, 1, 4, 4, 1, 3, 3, 1, 1, 1, 1, 1, 0, 1, 0, None), "fw_im2col")
    bad(lib.fw_im2col(p, 64, p, 576, 64, 2, 4, 4, 1, 3, 3, 1, 1, 1, 1, 1, 1, 2, 0, None), "fw_im2col")
    bad(lib.fw_im2col(odd, 64, p, 576, 64, 1, 4, 4, 1, 3, 3, 1, 1, 1, 1, 1, 0, 1, 0, None), "fw_im2col")
    # fw_conv_gemm_bf16: C % 64; frame window outside the volume; a map too tall for the packed row coordinates; ldw < taps * C
    cg = lambda *a: lib.fw_conv_gemm_bf16(*a)
    bad(cg(p, 72, 72, 1, 4, 4, 1, 3, 3, 1, 1, 1, 1, 1, 0, 1, p, 648, p, 64, 1, 64, None, 0, None, None, None, 0, 0, None), "fw_conv_gemm_bf16: C")
    bad(cg(p, 64, 64, 2, 4, 4, 1, 3, 3, 1, 1, 1, 1, 1, 1, 2, p, 576, p, 64, 1, 64, None, 0, None, None, None, 0, 0, None), "bad geometry")
    bad(cg(p, 64, 64, 1, 5000, 4, 1, 3, 3, 1, 1, 1, 1, 1, 0, 1, p, 576, p, 64, 1, 64, None, 0, None, None, None, 0, 0, None), "chunk the volume")
    bad(cg(p, 64, 64, 1, 4, 4, 1, 3, 3, 1, 1, 1, 1, 1, 0, 1, p, 512, p, 64, 1, 64, None, 0, None, None, None, 0, 0, None), "ldw < taps")
    bad(lib.fw_v_transpose_fp8(p, 128, 128 * 4, p, 100, 1, 1, 128, 4, None), "fw_v_transpose_fp8")            # lkp % 64 != 0
    bad(lib.fw_attention_fp8(p, 128, 0, p, 128, 0, p, 64, p, 128, 0, 1, 1, 96, 4, 4, 3, None), "head_dim must be 128")
    bad(lib.fw_attention_fp8(p, 120, 0, p, 128, 0, p, 64, p, 128, 0, 1, 1, 128, 4, 4, 3, None), "alignment contract")
    bad(lib.fw_resize_bilinear(p, 60, p, 60, 1, 2, 2, 4, 4, 60, None), "fw_resize_bilinear")
    bad(lib.fw_chan_rmsnorm_silu(p, 64, p, 64, 4, 64, 100, p, 1, None), "fw_chan_rmsnorm_silu")        # c_true > C
    bad(lib.fw_depth_to_space(p, 64, p, 64, 1, 2, 2, 0, 64, None), "fw_depth_to_space")               # k < 1
    bad(lib.fw_add_table(p, 64, p, 10, 4, 64, None), "fw_add_table")                                    # rows % hw != 0
    bad(lib.fw_unfold_time2(p, 60, p, 64, 1, 4, 60, None), "fw_unfold_time2")
    bad(lib.fw_add_act(p, None, p, 12, 0, None), "fw_add_act")                                          # n % 8 != 0
    bad(lib.fw_adaln_rows(None, p, p, 4, 64, ctypes.c_float(1e-6), None), "fw_adaln_rows")
    bad(lib.fw_head_activation(p, 4, 1, 0, p, p, None), "fw_head_activation")                           # n < 2
    bad(lib.fw_pixel_unshuffle(p, 7, p, 64, 1, 8, 8, 1, 8, None), "fw_pixel_unshuffle")                 # unknown dtype code
    bad(lib.fw_group_norm_rows(p, 64, p, 64, 1, 4, 64, 3, p, p, ctypes.c_float(1e-5), 0, None), "fw_group_norm_rows")   # C % groups
    bad(lib.fw_time_avg_pool(p, 64, p, 64, 0, 4, 64, None), "fw_time_avg_pool")
    bad(lib.fw_activation(p, p, 12, 1, None), "fw_activation")
    bad(lib.fw_softmax_rows(p, 16, p, 16, 2, 16, 8, ctypes.c_float(1.0), None), "fw_softmax_rows")      # cols_pad < cols
    # empty problems are accepted and do nothing
    assert lib.fw_add_act(p, None, p, 0, 0, None) == 0 and lib.fw_head_activation(p, 0, 4, 0, p, p, None) == 0
