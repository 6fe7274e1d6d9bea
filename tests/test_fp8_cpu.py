"""fp8 linear: the torch statement used as the checker (oracle/ref_ops.py: quantize_fp8_rows / linear_fp8) against
the reference's own AutoWrappedLinear.fp8_linear (FantasyWorld/diffsynth_wan22/vram_management/layers.py:115-151).

PARITY PIN, and its limit: the reference function cannot run on a CPU as written -- torch._scaled_mm with a per-row scale exists
only on GPU back-ends -- so the pin executes the reference's code with torch._scaled_mm replaced by its documented definition,
(a @ b) * scale_a * scale_b + bias -> out_dtype.  Everything before that call (row maximum, the clamp at 1, the 1e-8, the raw weight
cast, the bf16 bias) is the reference's own arithmetic; the matrix product itself is pinned to the definition only."""
import os

import pytest
import torch

from conftest import rel_l2
from oracle.ref_ops import TorchRefOps

from oracle import ref_locate

pytestmark = pytest.mark.skipif(not ref_locate.available(), reason="reference not mounted / staged")


def _scaled_mm_definition(a, b, scale_a=None, scale_b=None, bias=None, out_dtype=None, **kw):
    y = (a.float() @ b.float()) * scale_a.float() * scale_b.float()
    if bias is not None:
        y = y + bias.float()
    return y.to(out_dtype or a.dtype)


@pytest.mark.parametrize("M,N,K,amp", [(5, 64, 128, 3.0), (33, 192, 256, 300.0), (7, 128, 64, 0.01)])
def test_fp8_linear_restatement_matches_reference_code(M, N, K, amp, monkeypatch):
    from oracle import ref_harness
    ref_harness.install_stubs()
    from FantasyWorld.diffsynth_wan22.vram_management.layers import AutoWrappedLinear
    g = torch.Generator().manual_seed(M)
    lin = torch.nn.Linear(K, N)
    lin.weight.data = (torch.randn(N, K, generator=g) * K ** -0.5)
    lin.bias.data = torch.randn(N, generator=g) * 0.1
    lin = lin.bfloat16()
    wrapped = AutoWrappedLinear(lin, offload_dtype=torch.bfloat16, offload_device="cpu", onload_dtype=torch.bfloat16,
                                onload_device="cpu", computation_dtype=torch.float8_e4m3fn, computation_device="cpu", vram_limit=None)
    x = (torch.randn(M, K, generator=g) * amp).bfloat16()
    x[0, 0] = 1000.0 * amp                    # a row whose maximum exceeds 448: its scale leaves the clamp
    monkeypatch.setattr(torch, "_scaled_mm", _scaled_mm_definition)
    want = wrapped.fp8_linear(x, lin.weight, lin.bias)
    ops = TorchRefOps()
    got = ops.linear_fp8(x.float(), ops.pack_linear_fp8(lin.weight.float(), lin.bias.float(), bias_through_fp8=False))
    assert want.dtype == torch.bfloat16 and want.shape == (M, N)
    assert torch.equal(got.to(torch.bfloat16), want), rel_l2(got, want.float())
    q, scale = ops.quantize_fp8_rows(x.float())
    assert scale.min() >= 1.0 and (amp < 1.0 or scale.max() > 1.0)
    # through the MODULE (forward casts weight and bias to the computation dtype first, layers.py:158-159): the default packing
    want_mod = wrapped(x)
    got_mod = ops.linear_fp8(x.float(), ops.pack_linear_fp8(lin.weight.float(), lin.bias.float()))
    assert torch.equal(got_mod.to(torch.bfloat16), want_mod), rel_l2(got_mod, want_mod.float())
