"""Host logic of the geometry heads (fantasy_world_amd.heads) on the torch reference op set: weight packing
(tap-major, channel padding), the whole-sequence form of the cached temporal decode, the ReLU placement of the
ResidualConvUnits, frame chunking -- against the oracle and the reference's golden prediction dict."""
import torch

from conftest import PRED_KEYS, rel_l2
from fantasy_world_amd import heads as fw_heads
from oracle import fw_heads_oracle, ref_ops


def _predict(case, emulate_bf16=False, **kw):
    ops = ref_ops.TorchRefOps(emulate_bf16=emulate_bf16)
    gh = fw_heads.GeometryHeads(case.hc, case.weights.__getitem__, ops, **kw)
    ol = {k: v[None] for k, v in case.output_list.items()}
    return gh.predict(ol, case.S, case.ph, case.pw)


def test_heads_on_ref_ops_match_golden(heads_case):
    pred = _predict(heads_case)
    for k in PRED_KEYS:
        assert pred[k].shape == heads_case.golden[k].shape, (k, pred[k].shape)
        err = rel_l2(pred[k], heads_case.golden[k])
        assert err < 2e-5, f"{heads_case.name}:{k} rel-L2 {err:.3e}"


def test_heads_frame_chunking_is_invisible(heads_case):
    """Tiny gather budget (one frame per chunk) and a 2-frame fusion chunk give the same numbers."""
    a = _predict(heads_case)
    b = _predict(heads_case, max_col_bytes=1, frames_chunk=2)
    for k in PRED_KEYS:
        assert rel_l2(b[k], a[k]) < 1e-6, k


def test_heads_bf16_yardstick(heads_case):
    """With bf16 rounding where the HIP path stores bf16 the prediction stays within the tolerance the GPU test uses."""
    pred = _predict(heads_case, emulate_bf16=True)
    for k in PRED_KEYS:
        assert rel_l2(pred[k], heads_case.golden[k]) < 3e-2, k


def test_pos_embed_matches_oracle():
    for (C, ph, pw, asp) in [(64, 4, 6, 1.5), (32, 16, 24, 1.5), (128, 5, 3, 0.6)]:
        a = fw_heads.uv_pos_embed(C, ph, pw, asp)
        b = fw_heads_oracle.uv_pos_embed(C, ph, pw, asp).permute(1, 2, 0).reshape(ph * pw, C)
        assert torch.allclose(a, b, atol=1e-6)
