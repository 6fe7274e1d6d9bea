"""Test infrastructure has tests too: the checker-side helpers of oracle/ref_harness.py that the GPU box relies on (round 5)."""
import torch
import torch.nn.functional as F


def test_sdpa_by_head_chunks_is_the_unchunked_attention_and_restores_the_function():
    """The fp32 reference's attention runs head chunk by head chunk on the box (24 GB of scores at a time instead of 172 GB at the
    headline grid): heads are independent, so the chunked call must return what the unchunked call returns, for the [b, h, L, d] form
    (DiT, VGGT) and the [b * h, L, d] form (bicross, fusion/layer/block.py:598-605), and calls with a mask / positional extras must
    pass through untouched."""
    from oracle.ref_harness import sdpa_by_head_chunks
    torch.manual_seed(0)
    q, k, v = torch.randn(2, 6, 50, 16), torch.randn(2, 6, 70, 16), torch.randn(2, 6, 70, 16)
    orig = F.scaled_dot_product_attention
    want4 = orig(q, k, v)
    want3 = orig(q[0], k[0], v[0], attn_mask=None, dropout_p=0.0)
    mask = torch.zeros(50, 70)
    with sdpa_by_head_chunks(limit_bytes=2 * 50 * 70 * 4 * 2) as c:           # room for two heads of one batch element's scores
        assert F.scaled_dot_product_attention is not orig
        got4 = F.scaled_dot_product_attention(q, k, v)
        got3 = F.scaled_dot_product_attention(q[0], k[0], v[0], attn_mask=None, dropout_p=0.0)
        gotm = F.scaled_dot_product_attention(q, k, v, attn_mask=mask)       # masked: not chunked
    assert F.scaled_dot_product_attention is orig
    assert c.calls == 3 and c.chunked == 2
    assert torch.allclose(got4, want4, atol=1e-6) and torch.allclose(got3, want3, atol=1e-6)
    assert torch.allclose(gotm, orig(q, k, v, attn_mask=mask), atol=1e-6)
    with sdpa_by_head_chunks() as c:                                          # default limit: small problems are never chunked
        assert torch.equal(F.scaled_dot_product_attention(q, k, v), want4)
    assert c.chunked == 0


def test_fp8_linear_swap_counts_every_site_once():
    """swap_fp8_linears replaces exactly the nn.Linear modules enable_vram_management(module_map={nn.Linear: ...}) would wrap inside the
    DiT blocks (10 per block), for preconditioning blocks and the DiT halves of the IRG blocks alike."""
    from oracle import ref_harness

    class Attn(torch.nn.Module):
        def __init__(self):
            super().__init__()
            for n in ("q", "k", "v", "o"):
                setattr(self, n, torch.nn.Linear(8, 8))

    class Blk(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.self_attn, self.cross_attn = Attn(), Attn()
            self.ffn = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.GELU(), torch.nn.Linear(16, 8))

    class Irg(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.x_dit = Blk()

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.pipe = torch.nn.Module()
            self.pipe.dit = torch.nn.Module()
            self.pipe.dit.blocks = torch.nn.ModuleList([Blk(), Blk(), torch.nn.Identity()])
            self.IRGBlock = torch.nn.ModuleList([Irg()])

    m = Model()
    assert ref_harness.swap_fp8_linears(m, start_index=2) == 3 * len(ref_harness.FP8_SITES)
    assert isinstance(m.pipe.dit.blocks[0].ffn[0], ref_harness.Fp8LinearByDefinition)
    assert isinstance(m.IRGBlock[0].x_dit.cross_attn.o, ref_harness.Fp8LinearByDefinition)
    x = torch.randn(4, 8)
    y = m.pipe.dit.blocks[1].self_attn.q(x)
    assert y.shape == (4, 8) and torch.isfinite(y).all()
