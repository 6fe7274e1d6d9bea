"""Drop-in boundary: install() on the REAL reference model (build container only; skipped where /root/reference is
absent).  The reference object keeps its signature and returns; only the arithmetic provider changes."""
import os

import pytest
import torch

from conftest import rel_l2

from oracle import ref_locate

pytestmark = pytest.mark.skipif(not ref_locate.available(), reason="reference tree not mounted / staged")


def test_install_rebinds_joint_forward_on_reference_model(case_l2):
    from oracle import ref_harness
    from oracle.ref_ops import TorchRefOps
    from fantasy_world_amd import install, uninstall
    case = case_l2
    model = ref_harness.build_reference_wan21(case.cfg, weights=case.weights)
    ins = case.inputs
    kw = dict(timestep=ins["timestep"], context=ins["context"], clip_feature=ins["clip_feature"], y=ins["y"],
              use_gradient_checkpointing=False, camera_token=None, plucker_fea=ins["plucker_fea"],
              plucker_context_lens=ins["plucker_context_lens"], return_prediction=False)
    with torch.no_grad():
        want, p0 = model.joint_forward(ins["x"], **kw)
    eng = install(model, ops=TorchRefOps())
    assert eng.cfg.ffn_dim == case.cfg.ffn_dim and eng.cfg.cross_attention_list == case.cfg.cross_attention_list
    got, p1 = model.joint_forward(ins["x"], **kw)           # same call site, same kwargs as M21:295-305
    assert p0 is None and p1 is None
    assert got.shape == want.shape and got.dtype == want.dtype
    assert rel_l2(got, want) < 2e-5
    # the engine runs on a packed SNAPSHOT of the weights: a later change of the live parameters must not be silently ignored
    sd = {k: v * 1.01 for k, v in model.state_dict().items()}
    model.load_state_dict(sd)
    with pytest.raises(RuntimeError, match="install"):
        model.joint_forward(ins["x"], **kw)
    eng2 = install(model, ops=TorchRefOps(), precision="fp8")       # installing again re-packs (here: into the fp8 mode)
    assert eng2 is not eng and eng2.precision == "fp8" and not eng.invariants.entries
    got2, _ = model.joint_forward(ins["x"], **kw)
    assert torch.isfinite(got2).all() and 1e-3 < rel_l2(got2, want) < 0.5      # other weights, other precision: a new forward
    uninstall(model)
    with torch.no_grad():
        again, _ = model.joint_forward(ins["x"], **kw)
    assert not hasattr(model, "_fw_engine") and again.shape == want.shape


def test_install_can_release_the_reference_copy_of_the_weights(case_pred):
    """install(release_reference_weights=True): one resident copy of the weights (the packed one) instead of two -- the packed
    parameters' storage is released on the reference tree, joint_forward (incl. the geometry heads, packed eagerly) is unchanged,
    and uninstall() refuses, because the reference forward is gone."""
    from conftest import PRED_KEYS
    from oracle import ref_harness
    from oracle.ref_ops import TorchRefOps
    from fantasy_world_amd import install, uninstall
    c, ins = case_pred, case_pred.inputs
    model = ref_harness.build_reference_wan21(c.cfg, weights=c.weights, heads_cfg=c.hc)
    before = sum(p.numel() for p in model.parameters())
    eng = install(model, ops=TorchRefOps(), release_reference_weights=True)
    after = sum(p.numel() for p in model.parameters())
    assert model._fw_released_weights > 100 and after < 0.05 * before, (before, after)
    assert model.pipe.dit.blocks[0].ffn[0].weight.numel() == 0 and model.IRGBlock[0].x_dit.ffn[0].weight.numel() == 0
    kw = dict(timestep=ins["timestep"], context=ins["context"], clip_feature=ins["clip_feature"], y=ins["y"],
              use_gradient_checkpointing=False, camera_token=None, plucker_fea=ins["plucker_fea"],
              plucker_context_lens=ins["plucker_context_lens"], return_prediction=True)
    got, pred = model.joint_forward(ins["x"], **kw)
    assert rel_l2(got, c.golden["noise_pred"]) < 2e-5
    for k in PRED_KEYS:
        assert rel_l2(pred[k], c.golden[k]) < 5e-5, k
    got2, _ = model.joint_forward(ins["x"], **kw)                 # the weight watch accepts the released tree
    assert torch.equal(got2, got)
    with pytest.raises(RuntimeError, match="released"):
        uninstall(model)


def test_install_merges_the_cfg_pair_under_the_reference_loop(case_l2):
    """Single GPU, install() default (merge_cfg=True): the reference loop's two sequential joint_forward calls per step
    (model_wan21.py:295-319) become ONE merged pass from the second step on -- learnt from the identity of the latents / timestep /
    context objects, answered from a stash -- with the latents of the plain loop."""
    from oracle import ref_harness
    from oracle.ref_ops import TorchRefOps
    from fantasy_world_amd import install
    from fantasy_world_amd.sampler import FlowMatchScheduler
    case, ins = case_l2, case_l2.inputs
    model = ref_harness.build_reference_wan21(case.cfg, weights=case.weights)
    sched = FlowMatchScheduler()
    sched.set_timesteps(4)
    cond = dict(clip_feature=ins["clip_feature"], y=ins["y"], use_gradient_checkpointing=False, camera_token=None,
                plucker_fea=ins["plucker_fea"], plucker_context_lens=ins["plucker_context_lens"])

    def loop():
        lat = ins["x"]
        for step in range(3):
            t = sched.timesteps[step].reshape(1)
            with torch.no_grad():
                pos, _ = model.joint_forward(lat, timestep=t, context=ins["context"], **cond)
                neg, _ = model.joint_forward(lat, timestep=t, context=ins["context_neg"], **cond)
            lat = sched.step(neg + 5.0 * (pos - neg), step, lat)
        return lat
    want = loop()
    eng = install(model, ops=TorchRefOps())
    passes = []
    orig = eng._forward
    eng._forward = lambda x, t, contexts, *a, **k: (passes.append(len(contexts)), orig(x, t, contexts, *a, **k))[1]
    got = loop()
    assert passes == [1, 1, 2, 2], passes          # step 0: two plain forwards (learning); steps 1, 2: one merged pass each
    assert eng.cfg_pairing is not None and eng.cfg_pairing.pair[0] is ins["context"] and eng.cfg_pairing.pair[1] is ins["context_neg"]
    assert rel_l2(got, want) < 2e-5
    eng2 = install(model, ops=TorchRefOps(), merge_cfg=False)
    assert eng2.cfg_pairing is None


def test_install_on_reference_wan22_model(case_w22):
    """Same boundary on the Wan2.2 flavour: the M22 signature (control_camera_latents_input, no clip_feature / plucker_fea,
    FantasyWorld/fusion/model_wan22.py:231-242) is kept; the engine reads the control adapter off pipe.dit."""
    from oracle import ref_harness
    from oracle.ref_ops import TorchRefOps
    from fantasy_world_amd import install, uninstall
    case = case_w22
    model = ref_harness.build_reference_wan22(case.cfg, weights=case.weights)
    ins = case.inputs
    kw = dict(timestep=ins["timestep"], context=ins["context"], y=ins["y"], use_gradient_checkpointing=False,
              camera_token=None, control_camera_latents_input=ins["control_camera_latents_input"], uncond=False,
              return_prediction=False)
    with torch.no_grad():
        want, _ = model.joint_forward(ins["x"], **kw)
    eng = install(model, ops=TorchRefOps())
    assert eng.cfg.control_adapter and not eng.cfg.has_image_input and not eng.cfg.camera_adapter
    got, p1 = model.joint_forward(ins["x"], **kw)           # same call site, same kwargs as inference_wan22.py:243-253
    assert p1 is None and got.shape == want.shape and got.dtype == want.dtype
    assert rel_l2(got, want) < 2e-5
    assert rel_l2(got, case.golden["noise_pred"]) < 2e-5
    uninstall(model)


def _wan22_two_experts(build):
    """Two Wan2.2 experts (different weights) with narrow geometry heads, and the inputs of generate_video_with_dual_models.
    Memory is the constraint here (the CPU suite shares one 62 GB container with the session-scoped cases; a 3-layer pair built side
    by side took the process to 65 GB and the OOM killer): 1 PCB + 1 IRG block (0.7 B parameters per expert), head taps that read
    layer 0 only, and the experts are built ONE AT A TIME through `build(cfg, weights, hc)` so that only one fp32 weight dictionary
    exists at any moment."""
    import gc
    from fantasy_world_amd import config as fwc, synth
    cfg = fwc.plumbing22(num_layers=2, start_index=1)
    e2e = fwc.HeadsConfig.e2e_small()
    hc = fwc.HeadsConfig(dim_in=e2e.dim_in, trunk_depth=e2e.trunk_depth, cam_heads=e2e.cam_heads, features=e2e.features,
                         out_channels=list(e2e.out_channels), layer_idx=[0, 0, 0, 0], dpt_patch=e2e.dpt_patch)
    experts = []
    for seed in (0, 1):
        w = synth.make_weights(cfg, seed=seed)
        w.update(synth.make_heads_weights(hc, seed=seed))
        experts.append(build(cfg, w, hc))
        del w
        gc.collect()
    f, h2, w2 = 2, 8, 12
    ins = synth.make_inputs(cfg, f, h2, w2, seed=1, timestep=900.0, text_len=512)
    frames = 4 * (f - 1) + 1
    kw = dict(context_pos=ins["context"], context_neg=ins["context_neg"], y=ins["y"], height=8 * h2, width=8 * w2,
              num_frames=frames, sample_steps=4, plucker_embedding=synth.make_plucker(frames, 8 * h2, 8 * w2))
    return experts[0], experts[1], kw


def test_install_under_the_reference_wan22_dual_expert_loop():
    """The Wan2.2 sampler's OWN loop (inference_wan22.py:164-283 generate_video_with_dual_models, unmodified): it picks the
    high-noise or the low-noise expert per step by the timestep boundary and calls that model's joint_forward twice (CFG), with
    return_prediction on the last step.  install() on BOTH experts: two engines side by side, each learning its own CFG pair on
    the first step it serves and merging from its second; latents and the prediction dict of the plain reference loop."""
    from conftest import PRED_KEYS
    from oracle import ref_harness
    from oracle.ref_ops import TorchRefOps
    from fantasy_world_amd import install, uninstall
    high, low, kw = _wan22_two_experts(lambda cfg, w, hc: ref_harness.build_reference_wan22(cfg, weights=w, heads_cfg=hc))
    # shift-5 schedule over 4 steps: timesteps 1000, 937.5, 833.3, 625 -> boundary 900: two steps on each expert
    sampler = ref_harness.build_reference_wan22_sampler(high, low, seed=3, cfg_scale=5.0, timestep_boundary=900)
    with torch.no_grad():
        want, wpred = sampler.generate_video_with_dual_models(**kw)
    engines = [install(m, ops=TorchRefOps()) for m in (high, low)]
    passes = ([], [])
    for eng, log in zip(engines, passes):
        orig = eng._forward
        eng._forward = (lambda orig, log: lambda x, t, contexts, *a, **k: (log.append(len(contexts)), orig(x, t, contexts, *a, **k))[1])(orig, log)
    with torch.no_grad():
        got, pred = sampler.generate_video_with_dual_models(**kw)            # the same call, both experts on the engine
    assert passes == ([1, 1, 2], [1, 1, 2]), passes                          # per expert: learn the pair, then one merged pass
    assert got.shape == want.shape and got.dtype == want.dtype
    assert rel_l2(got, want) < 5e-5
    for k in PRED_KEYS:
        assert pred[k].shape == wpred[k].shape and rel_l2(pred[k], wpred[k]) < 1e-4, k
    for m in (high, low):
        uninstall(m)
    with torch.no_grad():
        back, _ = sampler.generate_video_with_dual_models(**kw)
    assert torch.equal(back, want)
    del engines, sampler, high, low
    import gc
    gc.collect()


def test_install_returns_prediction_dict_on_last_step(case_pred):
    """return_prediction=True (the last sampling step, M21:303-305): the rebound joint_forward returns the same dict as
    the reference's vggt._head_predction, computed by fantasy_world_amd.heads."""
    from conftest import PRED_KEYS
    from oracle import ref_harness
    from oracle.ref_ops import TorchRefOps
    from fantasy_world_amd import install
    c, ins = case_pred, case_pred.inputs
    model = ref_harness.build_reference_wan21(c.cfg, weights=c.weights, heads_cfg=c.hc)
    kw = dict(timestep=ins["timestep"], context=ins["context"], clip_feature=ins["clip_feature"], y=ins["y"],
              use_gradient_checkpointing=False, camera_token=None, plucker_fea=ins["plucker_fea"],
              plucker_context_lens=ins["plucker_context_lens"], return_prediction=True)
    with torch.no_grad():
        want, wpred = model.joint_forward(ins["x"], **kw)
    eng = install(model, ops=TorchRefOps())
    assert eng.heads_cfg is not None and eng.heads_cfg.layer_idx == c.hc.layer_idx and eng.heads_cfg.features == c.hc.features
    got, pred = model.joint_forward(ins["x"], **kw)
    assert rel_l2(got, want) < 2e-5
    for k in PRED_KEYS:
        assert pred[k].shape == wpred[k].shape and rel_l2(pred[k], wpred[k]) < 5e-5, k


def test_flash_attention_hook_b3():
    """Boundary B3: the reference's module-level flash_attention hook rebound to the engine's attention op; the reference's
    own DiTBlock then runs unchanged on top of it (SelfAttention -> AttentionModule -> flash_attention, DIT21:149-182)."""
    from oracle import ref_harness
    from oracle.ref_ops import TorchRefOps
    from fantasy_world_amd import install_flash_attention
    ref_harness.install_stubs()
    import FantasyWorld.diffsynth_wan21.models.wan_video_dit as dit
    g = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn(2, n, 4 * 64, generator=g) for n in (37, 50, 50))
    want = dit.flash_attention(q, k, v, 4)
    calls = []
    ops = TorchRefOps()
    orig_attention = ops.attention
    ops.attention = lambda *a, **kw: (calls.append(1), orig_attention(*a, **kw))[1]
    undo = install_flash_attention([dit], ops=ops)
    try:
        got = dit.flash_attention(q, k, v, 4)
        assert got.shape == want.shape and got.dtype == want.dtype and rel_l2(got, want) < 1e-5
        blk = dit.SelfAttention(256, 4).eval()
        x = torch.randn(1, 24, 256, generator=g)
        freqs = torch.polar(torch.ones(24, 1, 32, dtype=torch.float64), torch.zeros(24, 1, 32, dtype=torch.float64))
        n0 = len(calls)
        with torch.no_grad():
            y = blk(x, freqs)
        assert len(calls) == n0 + 1 and y.shape == x.shape       # the reference module went through the hook
    finally:
        undo()
    assert dit.flash_attention(q, k, v, 4).equal(want)


def test_install_rebinds_pose_encoder(case_l2):
    """CameraConditionModel.get_pose_fea (camera_control.py:233-234), the call generate_video makes before the loop
    (model_wan21.py:271): same signature, same plucker_fea after install()."""
    from oracle import ref_harness
    from oracle.ref_ops import TorchRefOps
    from fantasy_world_amd import install, uninstall, synth
    W = dict(case_l2.weights)
    W.update(synth.make_pose_encoder_weights())
    model = ref_harness.build_reference_wan21(case_l2.cfg, weights=W)
    pl = synth.make_plucker(9, 32, 48)
    with torch.no_grad():
        want = model.camera_condition.get_pose_fea(pl)
    install(model, ops=TorchRefOps())
    got = model.camera_condition.get_pose_fea(pl)
    assert got.shape == want.shape and got.dtype == want.dtype and rel_l2(got, want) < 5e-6
    assert model.camera_condition.get_pose_fea(None) is None
    uninstall(model)
    with torch.no_grad():
        assert torch.equal(model.camera_condition.get_pose_fea(pl), want)


def test_install_vae_keeps_reference_tiling():
    """WanVideoVAE.decode(tiled=True) (wan_video_vae.py:643-692, 776-782) with `model.decode` rebound: the reference's own tiling,
    mask blending and clamp run unchanged around the replaced per-tile decoder."""
    from oracle import ref_harness
    from oracle.ref_ops import TorchRefOps
    from fantasy_world_amd import install_vae, synth
    ref_harness.install_stubs()
    from FantasyWorld.diffsynth_wan21.models.wan_video_vae import WanVideoVAE
    vae = WanVideoVAE(z_dim=16)
    sd = vae.model.state_dict()
    sd.update(synth.make_vae_decoder_weights())
    vae.model.load_state_dict(sd)
    z = synth.make_latents(2, 6, 7)
    kw = dict(device="cpu", tiled=True, tile_size=(4, 4), tile_stride=(2, 3))
    with torch.no_grad():
        want = vae.decode(z, **kw)
        want1 = vae.decode(z, device="cpu", tiled=False)
    undo = install_vae(vae, ops=TorchRefOps())
    got = vae.decode(z, **kw)
    got1 = vae.decode(z, device="cpu", tiled=False)
    assert got.shape == want.shape == (1, 3, 5, 48, 56) and rel_l2(got, want) < 1e-5 and rel_l2(got1, want1) < 1e-5
    undo()
    with torch.no_grad():
        assert torch.equal(vae.decode(z, device="cpu", tiled=False), want1)


# ---------------------------------------------------------------------------------------------------------------
# the finer-grained B3 hooks (fantasy_world_amd/hooks.py) on the REAL reference modules, CPU op set
# ---------------------------------------------------------------------------------------------------------------
def test_bicross_attention_hook_b3():
    """BiMultiHeadAttention.attn_implementation / forward_sdpa (fusion/layer/block.py:323-325,393-410,532-625): the reference
    module keeps its projections and RoPE, the two SDPA calls go through the engine's attention op."""
    from oracle import ref_harness, fw_oracle
    from oracle.ref_ops import TorchRefOps
    from fantasy_world_amd import install_bicross_attention
    ref_harness.install_stubs()
    from FantasyWorld.fusion.layer.block import BiMultiHeadAttention, CrossModalityBiAttentionBlock
    torch.manual_seed(0)
    blk = CrossModalityBiAttentionBlock(64, 32, 192, 2).eval()                 # 2 heads x 96, like the real 12 x 96
    with torch.no_grad():
        blk.gamma_m1.normal_()
        blk.gamma_m2.normal_()
    f, h, w, ns = 2, 3, 4, 5
    fb = fw_oracle.precompute_freqs_cis_3d(96)
    fd, fa = fw_oracle.expand_freqs(fb, f, h, w), fw_oracle.build_freqs_3d_with_extra_cis(fb, f, h, w, ns)
    x1, x2 = torch.randn(1, f * h * w, 64), torch.randn(1, f * (ns + h * w), 32)
    with torch.no_grad():
        want = blk([x1, x2], freqs=None, freqs_dit=fd, freqs_agg=fa)
    calls = []
    ops = TorchRefOps()
    orig = ops.attention
    ops.attention = lambda *a, **kw: (calls.append(1), orig(*a, **kw))[1]
    undo = install_bicross_attention(blk, ops=ops)
    try:
        assert isinstance(blk.cross_attn, BiMultiHeadAttention) and blk.cross_attn.attn_implementation == "sdpa"
        with torch.no_grad():
            got = blk([x1, x2], freqs=None, freqs_dit=fd, freqs_agg=fa)
        assert len(calls) == 2
        for g_, w_ in zip(got, want):
            assert g_.shape == w_.shape and rel_l2(g_, w_) < 1e-5
    finally:
        undo()
    with torch.no_grad():
        again = blk([x1, x2], freqs=None, freqs_dit=fd, freqs_agg=fa)
    assert all(torch.equal(a, b) for a, b in zip(again, want)) and "forward_sdpa" not in blk.cross_attn.__dict__


def test_layernorm_kernel_hook_b3():
    """get_layernorm(use_kernel=True) (fusion/layer/block.py:693-708; apex FusedLayerNorm in the reference) builds a HipLayerNorm:
    CrossModalityBiAttentionBlock(enable_layernorm_kernel=True) constructs and computes what the nn.LayerNorm version does."""
    from oracle import ref_harness
    from oracle.ref_ops import TorchRefOps
    from fantasy_world_amd import install_layernorm_kernel, HipLayerNorm
    ref_harness.install_stubs()
    import FantasyWorld.fusion.layer.block as block
    original = block.get_layernorm
    undo = install_layernorm_kernel(block, ops=TorchRefOps())
    try:
        torch.manual_seed(1)
        a = block.CrossModalityBiAttentionBlock(64, 32, 192, 2, enable_layernorm_kernel=True).eval()
        assert isinstance(a.attn_norm_m1, HipLayerNorm) and a.attn_norm_m1.weight is None and a.attn_norm_m1.eps == 1e-6
        ln = block.get_layernorm(48, 1e-5, True, True)
        with torch.no_grad():
            ln.weight.normal_()
            ln.bias.normal_()
        x = torch.randn(3, 7, 48)
        want = torch.nn.functional.layer_norm(x, (48,), ln.weight, ln.bias, 1e-5)
        assert rel_l2(ln(x), want) < 1e-6
        assert isinstance(block.get_layernorm(48, 1e-5, True, False), torch.nn.LayerNorm)
    finally:
        undo()
    assert block.get_layernorm is original


@pytest.mark.parametrize("fp8", [False, True])
def test_hip_linear_in_enable_vram_management_module_map(fp8, monkeypatch):
    """The reference's module swap (diffsynth_wan21/vram_management/layers.py:145-166) with HipLinear as the target of nn.Linear:
    same outputs as the reference's own AutoWrappedLinear, in bf16 and -- computation_dtype float8_e4m3fn -- as fp8_linear
    (diffsynth_wan22/vram_management/layers.py:115-151; torch._scaled_mm replaced by its definition, as in tests/test_fp8_cpu.py)."""
    from oracle import ref_harness
    from oracle.ref_ops import TorchRefOps
    from fantasy_world_amd import HipLinear
    ref_harness.install_stubs()
    from FantasyWorld.diffsynth_wan22.vram_management.layers import AutoWrappedLinear, enable_vram_management

    def build():
        torch.manual_seed(3)
        net = torch.nn.Sequential(torch.nn.Linear(96, 128), torch.nn.GELU(), torch.nn.Linear(128, 40)).bfloat16()
        return net
    cdt = torch.float8_e4m3fn if fp8 else torch.bfloat16
    cfg = dict(offload_dtype=torch.bfloat16, offload_device="cpu", onload_dtype=torch.bfloat16, onload_device="cpu",
               computation_dtype=cdt, computation_device="cpu")
    ref_net, hip_net = build(), build()
    enable_vram_management(ref_net, {torch.nn.Linear: AutoWrappedLinear}, cfg)
    enable_vram_management(hip_net, {torch.nn.Linear: HipLinear}, dict(cfg, ops=TorchRefOps(emulate_bf16=True)))
    assert isinstance(hip_net[0], HipLinear) and hip_net[0].enable_fp8 == fp8 and hip_net.vram_management_enabled
    if fp8:
        monkeypatch.setattr(torch, "_scaled_mm", lambda a, b, scale_a=None, scale_b=None, bias=None, out_dtype=None, **kw:
                            ((a.float() @ b.float()) * scale_a.float() * scale_b.float() + bias.float()).to(out_dtype))
    x = torch.randn(2, 5, 96).bfloat16()
    with torch.no_grad():
        one_want, one_got = ref_net[0](x), hip_net[0](x)
        want, got = ref_net(x), hip_net(x)
    assert one_got.dtype == one_want.dtype and rel_l2(one_got.float(), one_want.float()) < (1e-6 if fp8 else 4e-3), rel_l2(one_got.float(), one_want.float())
    assert got.shape == want.shape and got.dtype == want.dtype
    # chained: a bf16 rounding difference of the hidden activation can flip an e4m3 quantisation step of the second layer
    assert rel_l2(got.float(), want.float()) < (3e-2 if fp8 else 6e-3)


def test_install_leaves_no_reference_cycle_and_watches_every_packed_tensor(case_l2):
    """(a) a model dropped WHILE installed must free its packed copy by reference counting (the rebound entries hold the
    model weakly), (b) a change to ANY packed parameter -- not just a sampled few -- is detected at the next call, (c) install() after
    release_reference_weights refuses instead of packing 0-element tensors."""
    import gc
    import weakref
    from oracle import ref_harness
    from oracle.ref_ops import TorchRefOps
    from fantasy_world_amd import install
    case = case_l2
    model = ref_harness.build_reference_wan21(case.cfg, weights=case.weights)
    ins = case.inputs
    kw = dict(timestep=ins["timestep"], context=ins["context"], clip_feature=ins["clip_feature"], y=ins["y"],
              use_gradient_checkpointing=False, camera_token=None, plucker_fea=ins["plucker_fea"],
              plucker_context_lens=ins["plucker_context_lens"], return_prediction=False)
    eng = install(model, ops=TorchRefOps())
    model.joint_forward(ins["x"], **kw)
    # (b) every packed parameter is watched: touch ONE tensor of ONE block in place
    names = [s[0] for s in eng.weight_watch.slots]
    assert len(names) > 100 and "IRGBlock.0.x_dit.ffn.2.bias" in names and "vggt.aggregator.frame_blocks.0.mlp.fc1.weight" in names
    with torch.no_grad():
        model.IRGBlock[0].x_dit.ffn[2].bias.add_(0.0)          # an in-place op bumps the version counter
    with pytest.raises(RuntimeError, match=r"IRGBlock\.0\.x_dit\.ffn\.2\.bias"):
        model.joint_forward(ins["x"], **kw)
    eng = install(model, ops=TorchRefOps())                      # "install again" works and re-packs
    model.joint_forward(ins["x"], **kw)
    # (a) no cycle through the rebound method: with the cyclic collector OFF, dropping the model frees the engine
    ref_engine, ref_model = weakref.ref(eng), weakref.ref(model)
    gc.collect()
    gc.disable()
    try:
        del eng, model
        assert ref_model() is None and ref_engine() is None
    finally:
        gc.enable()
    # (c) re-install after a release refuses
    model = ref_harness.build_reference_wan21(case.cfg, weights=case.weights)
    install(model, ops=TorchRefOps(), release_reference_weights=True)
    kept = [n for n, p in model.named_parameters() if p.numel() > 0]
    assert any(n.startswith("camera_condition.pose_encoder.") for n in kept)      # not packed by install(): keeps its storage
    assert not any(n.startswith("pipe.dit.blocks.0.") for n in kept)              # packed: released
    with pytest.raises(RuntimeError, match="released"):
        install(model, ops=TorchRefOps())
