"""Boundary B2 (fantasy_world_amd/blocks.py) on the REAL reference modules, CPU op set: DiTBlock.forward in its three modes
(wan_video_dit.py:279-313), VGGT Block.forward (vggt/layers/block.py:82-116), IRGBlock.forward (fusion/layer/block.py:97-143),
and install_blocks() under the reference's OWN joint_forward (model_wan21.py:104-224)."""
import pytest
import torch

from conftest import rel_l2
from oracle import ref_locate

pytestmark = pytest.mark.skipif(not ref_locate.available(), reason="reference tree not mounted / staged")


def _dit_inputs(L, D, hd, g, img=True):
    from oracle import fw_oracle
    x = torch.randn(1, L, D, generator=g)
    ctx = torch.randn(1, (257 if img else 0) + 40, D, generator=g)
    t_mod = torch.randn(1, 6, D, generator=g) * 0.2
    f, h, w = 2, 3, L // 6
    freqs = fw_oracle.expand_freqs(fw_oracle.precompute_freqs_cis_3d(hd), f, h, w)
    return x, ctx, t_mod, freqs


@pytest.mark.parametrize("flavour,img", [("diffsynth_wan21", True), ("diffsynth_wan22", False)])
def test_dit_block_three_modes(flavour, img):
    from oracle import ref_harness
    from oracle.ref_ops import TorchRefOps
    from fantasy_world_amd.blocks import install_dit_block
    ref_harness.install_stubs()
    dit = __import__(f"FantasyWorld.{flavour}.models.wan_video_dit", fromlist=["DiTBlock"])
    torch.manual_seed(0)
    D, H = 256, 2
    blk = dit.DiTBlock(img, D, H, 512).eval()
    g = torch.Generator().manual_seed(1)
    x, ctx, t_mod, freqs = _dit_inputs(24, D, D // H, g, img)
    with torch.no_grad():
        want_full = blk(x, ctx, t_mod, freqs)
        want_part, want_mods = blk(x, ctx, t_mod, freqs, return_partial=True)
        want_rest = blk(want_part, run_remaining=True, modifiers=want_mods)
    assert torch.equal(want_rest, want_full)
    x0 = x.clone()
    undo = install_dit_block(blk, ops=TorchRefOps())
    try:
        got_full = blk(x, ctx, t_mod, freqs)
        got_part, got_mods = blk(x, ctx, t_mod, freqs, return_partial=True)
        got_rest = blk(got_part, run_remaining=True, modifiers=got_mods)
        # modifiers override in the full mode (wan_video_dit.py:308-309)
        other = tuple(m * 0.5 for m in want_mods)
        got_over = blk(x, ctx, t_mod, freqs, modifiers=other)
        with torch.no_grad():
            pass
    finally:
        undo()
    with torch.no_grad():
        want_over = blk(x, ctx, t_mod, freqs, modifiers=other)
    assert torch.equal(x, x0), "the block must not modify its input"
    assert got_full.shape == want_full.shape and got_full.dtype == want_full.dtype
    assert rel_l2(got_full, want_full) < 2e-5 and rel_l2(got_part, want_part) < 2e-5 and rel_l2(got_rest, want_full) < 2e-5
    assert rel_l2(got_over, want_over) < 2e-5
    for a, b in zip(got_mods, want_mods):
        assert a.shape == b.shape and rel_l2(a, b) < 1e-6
    with torch.no_grad():
        assert torch.equal(blk(x, ctx, t_mod, freqs), want_full)        # undo restored the reference forward


def _vggt_block(C=128, heads=2):
    from FantasyWorld.vggt.layers.block import Block
    from FantasyWorld.vggt.layers.rope import RotaryPositionEmbedding2D
    return Block(dim=C, num_heads=heads, mlp_ratio=4.0, qkv_bias=True, proj_bias=True, ffn_bias=True, init_values=0.01,
                 qk_norm=True, rope=RotaryPositionEmbedding2D(frequency=100.0)).eval()


def _pos(S, h, w, ns=5):
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    p = torch.stack([ys.reshape(-1), xs.reshape(-1)], dim=-1) + 1
    return torch.cat([torch.zeros(ns, 2, dtype=p.dtype), p], dim=0)[None].expand(S, -1, -1).contiguous()


@pytest.mark.parametrize("global_mode", [False, True])
def test_vggt_block_modes(global_mode):
    from oracle import ref_harness
    from oracle.ref_ops import TorchRefOps
    from fantasy_world_amd.blocks import install_vggt_block
    ref_harness.install_stubs()
    torch.manual_seed(2)
    blk = _vggt_block()
    with torch.no_grad():
        blk.modulation.normal_(0, 0.2)
        blk.ls1.gamma.normal_(0, 0.3)
        blk.ls2.gamma.normal_(0, 0.3)
    S, h, w, C = 3, 2, 4, 128
    P = 5 + h * w
    g = torch.Generator().manual_seed(3)
    x = torch.randn(S, P, C, generator=g)
    pos = _pos(S, h, w)
    if global_mode:
        x, pos = x.reshape(1, S * P, C), pos.reshape(1, S * P, 2)
    e0 = torch.randn(1, 6, C, generator=g) * 0.3
    with torch.no_grad():
        want = blk(x, pos=pos, e0=e0)
        wp, wm = blk(x, pos=pos, e0=e0, return_partial=True)
    undo = install_vggt_block(blk, ops=TorchRefOps())
    try:
        got = blk(x, pos=pos, e0=e0)
        gp, gm = blk(x, pos=pos, e0=e0, return_partial=True)
        grest = blk(gp, run_remaining=True, modifiers=gm)
        with pytest.raises(ValueError, match="modulates"):
            blk(x, pos=pos)
    finally:
        undo()
    assert got.shape == want.shape and got.dtype == want.dtype
    assert rel_l2(got, want) < 2e-5 and rel_l2(gp, wp) < 2e-5 and rel_l2(grest, want) < 2e-5
    assert len(gm) == 6 and all(a.shape == b.shape and rel_l2(a, b) < 1e-6 for a, b in zip(gm, wm))


@pytest.mark.parametrize("uncond", [False, True])
def test_irg_block(uncond):
    from oracle import ref_harness, fw_oracle
    from oracle.ref_ops import TorchRefOps
    from fantasy_world_amd.blocks import install_irg_block
    ref_harness.install_stubs()
    import FantasyWorld.diffsynth_wan21.models.wan_video_dit as dit
    from FantasyWorld.fusion.layer.block import IRGBlock
    torch.manual_seed(4)
    D, H, C = 256, 2, 128
    irg = IRGBlock(x_dit_block=dit.DiTBlock(True, D, H, 512), x_agg_block=_vggt_block(C, 2), m1_dim=D, m2_dim=C, hidden_size=192,
                   num_heads=2, drop_path=None).eval()
    with torch.no_grad():
        irg.bicross_attention.gamma_m1.normal_()
        irg.bicross_attention.gamma_m2.normal_()
        irg.x_agg.modulation.normal_(0, 0.2)
    f, h, w, ns = 2, 3, 4, 5
    L, P = f * h * w, ns + h * w
    g = torch.Generator().manual_seed(5)
    x, ctx, t_mod, freqs = _dit_inputs(L, D, D // H, g)
    fb = fw_oracle.precompute_freqs_cis_3d(96)
    fd, fa = fw_oracle.expand_freqs(fb, f, h, w), fw_oracle.build_freqs_3d_with_extra_cis(fb, f, h, w, ns)
    tok = torch.randn(f, P, C, generator=g)
    pos = _pos(f, h, w, ns)
    e0 = torch.randn(1, 6, C, generator=g) * 0.3
    kw = dict(context=ctx, t_mod=t_mod, freqs=freqs, freqs_dit=fd, freqs_agg=fa, pos=pos, e0=e0, uncond=uncond)
    with torch.no_grad():
        wx, wt, wi = irg(x_dit=x, x_agg=tok, **kw)
    undo = install_irg_block(irg, ops=TorchRefOps())
    try:
        gx, gt, gi = irg(x_dit=x, x_agg=tok, **kw)
    finally:
        undo()
    assert gx.shape == wx.shape and gt.shape == wt.shape and gi[0].shape == wi[0].shape == (1, f, P, C)
    assert gx.dtype == wx.dtype and gt.dtype == wt.dtype
    assert rel_l2(gx, wx) < 2e-5 and rel_l2(gt, wt) < 2e-5 and torch.equal(gi[0].reshape(gt.shape), gt)


def test_install_blocks_under_reference_joint_forward(case_l3):
    """Every PCB DiT block (with its camera adapter processor), VGGT frame block and IRGBlock of the real fusion model rebound;
    the reference's own joint_forward walks them and must reproduce its golden."""
    from oracle import ref_harness
    from oracle.ref_ops import TorchRefOps
    from fantasy_world_amd.blocks import install_blocks
    case = case_l3
    model = ref_harness.build_reference_wan21(case.cfg, weights=case.weights)
    ins = case.inputs
    kw = dict(timestep=ins["timestep"], context=ins["context"], clip_feature=ins["clip_feature"], y=ins["y"],
              use_gradient_checkpointing=False, camera_token=None, plucker_fea=ins["plucker_fea"],
              plucker_context_lens=ins["plucker_context_lens"], return_prediction=False)
    calls = []
    ops = TorchRefOps()
    orig = ops.attention
    ops.attention = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    undo = install_blocks(model, ops=ops)
    with torch.no_grad():
        got, pred = model.joint_forward(ins["x"], **kw)
    # 1 PCB block (self + text + image) + 2 x [frame + DiT (3) + global + bicross (2)] attention launches
    assert pred is None and len(calls) == 3 + 2 * 7
    assert rel_l2(got, case.golden["noise_pred"]) < 2e-5
    undo()
    with torch.no_grad():
        again, _ = model.joint_forward(ins["x"], **kw)
    assert rel_l2(again, case.golden["noise_pred"]) < 5e-6 and len(calls) == 3 + 2 * 7
