"""-m gpu: the model bench.py TIMES -- 40 DiT blocks (16 preconditioning + 24 inside IRG blocks), 24 + 24 VGGT frame / global blocks,
24 bicross blocks, `start_index` 16, `cross_attention_list` range(24), full-width geometry heads -- against the REAL reference at
that depth (VERDICT r03 "J0": every other end-to-end comparison builds 2-8 blocks).

The checker is the reference's own FantasyWorldFusionModel (oracle/_ref bundle, unmodified) built the way its __init__ does
(oracle/ref_harness.py; model_wan21.py:24-102 / model_wan22.py:122-229), on PyTorch-ROCm in fp32 on the same GPU:
  * `CameraConditionModel(dit, ...)` installs the camera adapters through ITS OWN filter (wan_video_dit.py:505-527, blocks <= 24), the
    IRG assembly loop then moves DiT blocks 16..39 into the IRG blocks: adapters end up on PCB 0..15 + IRG 0..8, none on IRG 9..23 --
    what fantasy_world_amd.config.FWConfig.has_adapter hard-codes;
  * VGGT() builds its heads with the reference's default `intermediate_layer_idx` (dpt_head.py:44: 23, 17, 11, 7).
Weights are the synthetic ones (fantasy_world_amd.synth), drawn per name ON the device (LazyWeights: 18.5 B parameters, 74 GB fp32).
Then install() (B1) and the same calls on the HIP path; compared: noise_pred, both residual streams after DiT blocks 15 / 24 / 25 / 39
(last PCB, last IRG block with an adapter, first without, last), and the prediction dict of joint_forward(return_prediction=True).

Grids: a small one (seconds), and one whose token count (L = 8190) puts every GEMM and attention on the kernels the benchmark runs
(256 x 256 ping-pong GEMM with the M-tail peel, 64 query blocks per head, split-KV tails); for Wan2.1 -- the model bench.py times --
additionally ONE forward at the headline grid itself (L = 32 760; measured 2.97e-3, profiles/r04/parity.json).  Round 5: the headline
leg runs for BOTH flavours by default, the checker's attention runs head chunk by head chunk (oracle/ref_harness.py:sdpa_by_head_chunks --
24 GB of fp32 scores at a time instead of 172 GB; heads are independent, the module tree stays unmodified), and a checker that still
does not fit FAILS the test (FW_FULL_DEPTH_HEADLINE_OPTIONAL=1 turns that into a recorded note; FW_FULL_DEPTH_HEADLINE=0 skips the leg;
FW_FULL_DEPTH_CONFIG4=1 adds the Wan2.2 81f x 720p grid, L = 75 600, minutes of fp32 reference time).

Round 5 also: (i) the bf16-rounding YARDSTICK at this depth (opt-in, FW_FULL_DEPTH_YARDSTICK=1) -- the engine's own host code on torch ops with activations rounded to bf16
where the HIP path stores bf16 (oracle/ref_ops.py:TorchRefOps(emulate_bf16=True)) against the same fp32 reference, 40 / 24 / 24 blocks,
L = 8190 -- recorded next to the HIP path's numbers (the "floor" docs/parity.md quotes was an 8-block CPU number until now);
(ii) test_full_depth_fp8_*: BASELINE configs[4]'s arithmetic ("fp8 attention + FFN") at the benchmarked depth.
"""
import os

import pytest
import torch

from conftest import rel_l2, PRED_KEYS
from oracle import ref_locate

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not ref_locate.available(), reason="reference not mounted / staged")]

DEV = "cuda:0"
WATCH = {15: "last_pcb", 24: "last_block_with_adapter", 25: "first_block_without_adapter", 39: "last_block"}
# physical bounds: the reference's OWN bf16 inference configuration is 1.2e-2 from its fp32 run after 8 blocks;
# the tight (2.5 x measured) bounds live in tests/golden/parity_bounds_gpu.json
STREAM_TOL, OUT_TOL, PRED_TOL = 3e-2, 3e-2, 8e-2


def _sync():
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def _build(flavour):
    from fantasy_world_amd import config as fwc, synth
    from oracle import ref_harness
    cfg = fwc.wan21_14b() if flavour == "wan21" else fwc.wan22_a14b()
    specs = [synth.weight_spec(cfg), synth.heads_weight_spec(fwc.HeadsConfig())]
    if flavour == "wan21":
        specs.append(synth.pose_encoder_weight_spec())
    weights = synth.LazyWeights(*specs, device=DEV)
    build = ref_harness.build_reference_wan21 if flavour == "wan21" else ref_harness.build_reference_wan22
    model = build(cfg, weights=weights)                      # heads_cfg=None: the reference's own default heads
    assert not model._fw_missing and not model._fw_unused, (model._fw_missing[:4], model._fw_unused[:4])
    model.to(device=DEV, dtype=torch.float32)
    model.pipe.device, model.pipe.torch_dtype, model.device = DEV, torch.float32, DEV
    # the shape of the benchmarked model, read off the REFERENCE's module tree
    assert len(model.pipe.dit.blocks) == 40 and len(model.IRGBlock) == 24 and model.start_index == 16
    assert list(model.vggt.depth_head.intermediate_layer_idx) == [23, 17, 11, 7]
    if flavour == "wan21":
        adapters = [type(b.cross_attn.processor).__name__ == "CrossAttentionAdapterProcessor"
                    for b in list(model.pipe.dit.blocks)[:16]] + \
                   [type(ib.x_dit.cross_attn.processor).__name__ == "CrossAttentionAdapterProcessor" for ib in model.IRGBlock]
        assert adapters == [b <= 24 for b in range(40)], adapters
    return cfg, model


def _kwargs(cfg, ins, return_prediction):
    if cfg.control_adapter:
        return dict(timestep=ins["timestep"], context=ins["context"], y=ins["y"], use_gradient_checkpointing=False, camera_token=None,
                    control_camera_latents_input=ins["control_camera_latents_input"], uncond=False,
                    return_prediction=return_prediction)
    return dict(timestep=ins["timestep"], context=ins["context"], clip_feature=ins["clip_feature"], y=ins["y"],
                use_gradient_checkpointing=False, camera_token=None, plucker_fea=ins["plucker_fea"],
                plucker_context_lens=ins["plucker_context_lens"], return_prediction=return_prediction)


def _reference_forward(model, cfg, ins, return_prediction):
    """The reference's own joint_forward with forward hooks on the four watched blocks."""
    cap, hooks = {"x": {}, "tok": {}}, []
    for b in WATCH:
        if b < cfg.start_index:
            hooks.append(model.pipe.dit.blocks[b].register_forward_hook(
                lambda m, a, out, b=b: cap["x"].__setitem__(b, out[0].float().clone())))
        else:
            def hook(m, a, out, b=b):
                cap["x"][b] = out[0][0].float().clone()
                cap["tok"][b] = out[1][0].float().clone()
            hooks.append(model.IRGBlock[b - cfg.start_index].register_forward_hook(hook))
    from oracle import ref_harness
    try:
        with torch.no_grad(), ref_harness.sdpa_by_head_chunks() as chunks:
            out, pred = model.joint_forward(ins["x"], **_kwargs(cfg, ins, return_prediction))
        assert chunks.calls > 0                 # the reference's attention sites did go through the call that is chunked
    finally:
        for h in hooks:
            h.remove()
    _sync()
    return out, pred, cap


def _hip_forward(model, eng, cfg, ins, return_prediction):
    got = {"x": {}, "tok": {}}

    def per_block(kind, i, t):
        b = i if kind == "x" else cfg.start_index + i
        if b in WATCH:
            got[kind][b] = t.clone()
    kw = _kwargs(cfg, ins, return_prediction)
    for k in ("use_gradient_checkpointing", "timestep", "context"):
        kw.pop(k)
    out, pred = eng.joint_forward(ins["x"], ins["timestep"], ins["context"], collect={"per_block": per_block}, **kw)
    _sync()
    return out, pred, got


def _compare(tag, parity, want, got, cfg, with_pred):
    wout, wpred, wcap = want
    gout, gpred, gcap = got
    assert gout.shape == wout.shape
    rows = {}
    rows["noise_pred"] = parity.check(f"{tag}/noise_pred", rel_l2(gout.float(), wout.float()), OUT_TOL)
    for b, what in WATCH.items():
        rows[f"x@{b}"] = parity.check(f"{tag}/x_stream_after_block_{b}_{what}", rel_l2(gcap["x"][b], wcap["x"][b]), STREAM_TOL)
        if b >= cfg.start_index:
            w = wcap["tok"][b]
            rows[f"tok@{b}"] = parity.check(f"{tag}/vggt_stream_after_block_{b}_{what}",
                                            rel_l2(gcap["tok"][b].reshape(w.shape), w), STREAM_TOL)
    if with_pred:
        assert set(PRED_KEYS) <= set(gpred) and set(PRED_KEYS) <= set(wpred)
        for k in PRED_KEYS:
            assert gpred[k].shape == wpred[k].shape, (k, gpred[k].shape, wpred[k].shape)
            # the reference computes its heads under ITS bf16 autocast on a GPU (vggt.py:136): the checker carries bf16 noise here
            rows[k] = parity.check(f"{tag}/{k}", rel_l2(gpred[k].float(), wpred[k].float()), PRED_TOL)
    print(tag, {k: f"{v:.2e}" for k, v in rows.items()})
    # a broken adapter boundary shows as a jump between blocks 24 and 25, far above the depth curve's slope
    assert rows["x@25"] < rows["x@24"] + 2.5 * rows["x@15"] + 1e-3, rows
    return rows


GRIDS = {"small_f3_12x16": (3, 12, 16), "production_kernels_f21_30x52": (21, 30, 52)}
YARDSTICK_GRID = "production_kernels_f21_30x52"


def _grids(flavour):
    grids = dict(GRIDS)
    # the benchmarked model AT the benchmarked grid (~40 s of reference time per flavour on the box)
    if os.environ.get("FW_FULL_DEPTH_HEADLINE", "1") == "1":
        grids["headline_f21_60x104"] = (21, 60, 104)
    if flavour == "wan22" and os.environ.get("FW_FULL_DEPTH_CONFIG4", "0") == "1":
        grids["config4_f21_90x160"] = (21, 90, 160)
    return grids


def _reference_or_fail(model, cfg, ins, with_pred, parity, tag):
    """The checker at a big grid must not vanish from a green run: out of memory fails the test unless the
    leg was declared optional."""
    try:
        return _reference_forward(model, cfg, ins, with_pred)                  # the reference, fp32, PyTorch-ROCm
    except torch.OutOfMemoryError as e:
        torch.cuda.empty_cache()
        if os.environ.get("FW_FULL_DEPTH_HEADLINE_OPTIONAL", "0") == "1" and not tag.split("/")[-1].startswith(("small", "production")):
            parity.note(f"{tag}/checker_did_not_fit", str(e)[:160])
            return None
        pytest.fail(f"{tag}: the fp32 reference did not fit on the device ({str(e)[:200]}); "
                    "FW_FULL_DEPTH_HEADLINE_OPTIONAL=1 records this as a note instead")


@pytest.mark.parametrize("flavour", ["wan21", "wan22"])
def test_full_depth_model_matches_reference(flavour, parity):
    from fantasy_world_amd import install, uninstall, synth
    from fantasy_world_amd.hip_ops import HipOps
    cfg, model = _build(flavour)
    inputs, want = {}, {}
    for name, (f, h2, w2) in _grids(flavour).items():
        ins = synth.make_inputs(cfg, f, h2, w2, seed=11, device=DEV, dtype=torch.float32)
        w = _reference_or_fail(model, cfg, ins, name.startswith("small"), parity, f"full_depth/{flavour}/{name}")
        if w is not None:
            inputs[name], want[name] = ins, w
    eng = install(model, ops=HipOps(DEV), merge_cfg=False)
    assert eng.heads_cfg is not None and list(eng.heads_cfg.layer_idx) == [23, 17, 11, 7]
    assert [eng.cfg.has_adapter(b) for b in range(40)] == [flavour == "wan21" and b <= 24 for b in range(40)]
    hip_rows = {}
    try:
        for name, ins in inputs.items():
            with_pred = name.startswith("small")
            got = _hip_forward(model, eng, cfg, ins, with_pred)
            keep = want[name] if (flavour == "wan21" and name == YARDSTICK_GRID and
                                  os.environ.get("FW_FULL_DEPTH_YARDSTICK", "0") == "1") else want.pop(name)
            hip_rows[name] = _compare(f"full_depth/{flavour}/{name}", parity, keep, got, cfg, with_pred)
            # and through the rebound method itself (B1), the call the reference's loop makes
            out, pred = model.joint_forward(ins["x"], **_kwargs(cfg, ins, False))
            torch.cuda.synchronize()
            assert pred is None and torch.equal(out, got[0])
    finally:
        uninstall(model)
    del eng
    torch.cuda.empty_cache()
    # opt-in (FW_FULL_DEPTH_YARDSTICK=1; ~40 s of fp32 torch ops and 74 GB more of weights): measured in round 5 -- floor 3.008e-3, HIP
    # 3.018e-3 (profiles/r05/parity_call1_full_depth.json, docs/parity.md); the default run keeps the suite's wall time where it was
    if flavour == "wan21" and os.environ.get("FW_FULL_DEPTH_YARDSTICK", "0") == "1":
        _yardstick(model, cfg, inputs[YARDSTICK_GRID], want.pop(YARDSTICK_GRID), hip_rows[YARDSTICK_GRID], parity)


def _yardstick(model, cfg, ins, want, hip, parity):
    """What bf16 storage of the activations costs at THIS depth, independent of any HIP kernel: the engine's host code (same op order,
    same fusion of gates / residuals, same fp32 residual streams) on plain torch ops in fp32 with a bf16 rounding wherever the HIP path
    stores bf16, against the same fp32 reference.  The HIP path is then read against that floor (VERDICT r04 "weak 1", "next" 6)."""
    from fantasy_world_amd.engine import FusionEngine
    from oracle.ref_ops import TorchRefOps
    params = dict(model.named_parameters())
    eng = FusionEngine(cfg, params.__getitem__, TorchRefOps(emulate_bf16=True, device=DEV))
    got = _hip_forward(model, eng, cfg, ins, False)
    wout, _, wcap = want
    floor = {"noise_pred": rel_l2(got[0].float(), wout.float())}
    for b in WATCH:
        floor[f"x@{b}"] = rel_l2(got[2]["x"][b], wcap["x"][b])
        if b >= cfg.start_index:
            w = wcap["tok"][b]
            floor[f"tok@{b}"] = rel_l2(got[2]["tok"][b].reshape(w.shape), w)
    tag = f"full_depth/wan21/{YARDSTICK_GRID}/yardstick_bf16_rounding_on_torch_ops"
    for k, v in floor.items():
        parity.note(f"{tag}/{k}", {"floor": v, "hip": hip[k], "hip_over_floor": hip[k] / v})
    print(tag, {k: f"{v:.2e} (hip {hip[k]:.2e})" for k, v in floor.items()})
    # the HIP path may not sit far above the rounding floor of its own storage format (10 % is the review's figure for "find the kernel";
    # the assert leaves room for the floor's own run-to-run spread of the torch kernels)
    assert hip["noise_pred"] < 1.25 * floor["noise_pred"] + 2e-4, (hip["noise_pred"], floor["noise_pred"])
    if os.environ.get("FW_FULL_DEPTH_ABLATION", "0") == "1":
        _ablation(model, cfg, ins, want, eng, floor["noise_pred"], parity)


# Which bf16 STORES make the floor?  Groups of rounding sites of oracle/ref_ops.py (prefixes) at the 14B widths.
ABLATION_SITES = {
    "layernorm_outputs_dit": ("ln:C5120",),
    # round 6, second pass: the ONE LayerNorm in front of the output head (its rounding reaches noise_pred without a residual stream
    # to average it) apart from the 160 inside the blocks
    "layernorm_output_in_front_of_the_head_only": ("ln:C5120:head",),
    "layernorm_outputs_inside_the_blocks_only": ("ln:C5120:blk",),
    "dit_qkv_projection_and_qk_pass": ("linear_out:N15360:K5120", "qk:hd128"),
    "dit_self_attention_output": ("attn_o:hd128:long",),
    "dit_ffn_hidden": ("linear_out:N13824:K5120",),
    "dit_cross_attention_q_kv_and_outputs": ("linear_out:N5120:K5120", "linear_out:N10240:K5120", "attn_o:hd128:short"),
    "camera_adapter_and_bridge_linears": ("linear_out:N1024:K5120", "linear_out:N2048:", "linear_out:N448:", "linear_out:N5120:K448", "cast_act"),
    "bicross_operands_and_outputs": ("linear_out:N2304:", "qk:hd96", "attn_o:hd96"),
    "vggt_branch_all_sites": ("ln:C1024", "linear_out:N3072:K1024", "linear_out:N4096:K1024", "qk:hd64", "attn_o:hd64"),
    "inputs_and_context_embeddings": ("input", "linear_out:N5120:K4096", "linear_out:N1280:", "linear_out:N5120:K1280", "ln:C1280"),
}


def _ablation(model, cfg, ins, want, eng, floor_all, parity):
    """Per-site decomposition of the bf16-storage floor at 40 / 24 / 24 blocks: the SAME torch-op engine (weights packed once), noise_pred
    rel-L2 against the fp32 reference with (a) ONE group of rounding sites left in fp32 (leave-one-out: what an fp32 / split form of that
    store could buy) and (b) ONLY that group rounded (one-in: what it costs alone); plus the attention probabilities rounded to bf16 for
    PV the way the HIP kernels do (not part of the default floor).  Recorded under full_depth/.../ablation/*; ranked table in docs/parity.md."""
    ops = eng.ops
    wout = want[0].float()
    epi = eng._epilogue

    def epilogue(*a, **k):           # the head's LayerNorm is a site of its own
        ops.site_suffix = ":head"
        try:
            return epi(*a, **k)
        finally:
            ops.site_suffix = ":blk"
    eng._epilogue = epilogue

    def run():
        out, _, _ = _hip_forward(model, eng, cfg, ins, False)
        return rel_l2(out.float(), wout)
    tag = f"full_depth/wan21/{YARDSTICK_GRID}/ablation"
    rows = {}
    try:
        for name, sites in ABLATION_SITES.items():
            ops.fp32_sites, ops.only_sites = tuple(sites), None
            loo = run()
            ops.fp32_sites, ops.only_sites = (), tuple(sites)
            only = run()
            rows[name] = {"all_but_this_in_bf16": loo, "only_this_in_bf16": only,
                          "share_of_floor_squared": 1.0 - (loo / floor_all) ** 2}
            parity.note(f"{tag}/{name}", rows[name])
            print(tag, name, {k: f"{v:.3e}" for k, v in rows[name].items()}, flush=True)
        ops.fp32_sites, ops.only_sites, ops.emulate_p = (), None, True
        with_p = run()
        ops.fp32_sites, ops.only_sites = (), ("attn_p",)
        only_p = run()
        parity.note(f"{tag}/attention_probabilities_rounded_for_pv", {"floor_with_p_rounding": with_p, "only_this_in_bf16": only_p,
                                                                      "floor_without": floor_all})
        print(tag, "attention P rounded to bf16 for PV", f"{with_p:.3e} (floor {floor_all:.3e}), alone {only_p:.3e}", flush=True)
    finally:
        ops.fp32_sites, ops.only_sites, ops.emulate_p = (), None, False


# ---- BASELINE configs[4] ("Wan2.2 A14B fp8 MFMA path ... fp8 attention + FFN") at the benchmarked depth -------------------------
# physical bounds (e4m3 carries 3 mantissa bits: one rounding is ~3e-2 of an element; the reference's own fp8 linears put it 2e-2 from its
# fp32 run after TWO blocks, tests/test_config5_gpu.py); the tight bounds come from tests/golden/parity_bounds_gpu.json
FP8_OUT_TOL, FP8_STREAM_TOL = 6e-2, 6e-2
# fp8 attention has NO reference semantics (the reference defines fp8 for linears only, vram_management/layers.py:115-151): its pin is the
# distance from the SAME engine with bf16 attention, fp8 linears in both -- STATED and enforced here:
FP8_ATTENTION_VS_BF16_ATTENTION_TOL = 2e-2


def test_full_depth_fp8_linears_and_fp8_attention(parity):
    """Wan2.2 flavour, 40 / 24 / 24 blocks.  (a) engine with precision="fp8" against the reference whose DiT-block nn.Linears are computed
    by the reference's own fp8 linear (the modules enable_vram_management would wrap, oracle/ref_harness.py:swap_fp8_linears), streams
    after blocks 15 / 24 / 25 / 39 so that e4m3's behaviour over depth is a curve; (b) fp8 attention on top against the fp8-linear
    engine with bf16 attention, under FP8_ATTENTION_VS_BF16_ATTENTION_TOL."""
    from fantasy_world_amd import install, uninstall, synth
    from fantasy_world_amd.engine import FusionEngine
    from fantasy_world_amd.hip_ops import HipOps
    from oracle import ref_harness
    cfg, model = _build("wan22")
    ops = HipOps(DEV)
    inputs, want32 = {}, {}
    # the grid whose token count puts every GEMM / attention on the production kernels; FW_FULL_DEPTH_FP8_SMALL=1 adds the 144-token
    # grid (measured in round 5: 1.51e-2 / 1.55e-2, the same plateau)
    grids = {k: v for k, v in GRIDS.items() if k.startswith("production") or os.environ.get("FW_FULL_DEPTH_FP8_SMALL", "0") == "1"}
    for name, (f, h2, w2) in grids.items():
        inputs[name] = synth.make_inputs(cfg, f, h2, w2, seed=11, device=DEV, dtype=torch.float32)
        want32[name] = _reference_forward(model, cfg, inputs[name], False)

    eng = install(model, ops=ops, merge_cfg=False, precision="fp8")
    try:
        got8 = {name: _hip_forward(model, eng, cfg, ins, False) for name, ins in inputs.items()}
    finally:
        uninstall(model)
    del eng
    # round 6: fp8 attention through the SAME boundary -- install(model, precision="fp8", fp8_attention=True), tolerance stated in
    # INTEGRATION.md and enforced below
    eng = install(model, ops=ops, merge_cfg=False, precision="fp8", fp8_attention=True)
    assert eng.fp8_attention and eng.precision == "fp8"
    try:
        got8a = {name: _hip_forward(model, eng, cfg, ins, False) for name, ins in inputs.items()}
    finally:
        uninstall(model)
    del eng
    torch.cuda.empty_cache()
    # round 6 experiment (opt-in): the bicross attention on e4m3 operands as well -- against the engine with fp8 DiT attention only,
    # under the same stated 2e-2: noise_pred and both streams (the VGGT stream after block 39 is what the geometry heads read)
    if os.environ.get("FW_FULL_DEPTH_FP8_BICROSS", "0") in ("1", "all"):
        # ("all": additionally the VGGT frame / global attention on the head_dim-64 fp8 kernel -- second half of round 6)
        mode = "all" if os.environ["FW_FULL_DEPTH_FP8_BICROSS"] == "all" else "bicross"
        eng = install(model, ops=ops, merge_cfg=False, precision="fp8", fp8_attention=mode)
        assert eng.fp8_bicross and eng.fp8_vggt == (mode == "all")
        try:
            got8b = {name: _hip_forward(model, eng, cfg, ins, False) for name, ins in inputs.items()}
        finally:
            uninstall(model)
        del eng
        torch.cuda.empty_cache()
        for name in inputs:
            tag = f"full_depth_fp8/wan22/{name}/" + ("fp8_bicross_vs_bf16_bicross_fp8_dit_attention_in_both" if mode == "bicross" else "fp8_all_attention_vs_fp8_dit_attention_only")
            (aout, _, acap), (bout, _, bcap) = got8a[name], got8b[name]
            row = {"noise_pred": parity.check(f"{tag}/noise_pred", rel_l2(bout.float(), aout.float()), FP8_ATTENTION_VS_BF16_ATTENTION_TOL)}
            for b, what in WATCH.items():
                row[f"x@{b}"] = parity.check(f"{tag}/x_stream_after_block_{b}_{what}", rel_l2(bcap["x"][b], acap["x"][b]),
                                             FP8_ATTENTION_VS_BF16_ATTENTION_TOL)
                if b >= cfg.start_index:
                    row[f"tok@{b}"] = parity.check(f"{tag}/vggt_stream_after_block_{b}_{what}", rel_l2(bcap["tok"][b], acap["tok"][b]),
                                                   FP8_ATTENTION_VS_BF16_ATTENTION_TOL)
            print(tag, {k: f"{v:.2e}" for k, v in row.items()}, flush=True)
            # and the whole fp8-attention option (DiT self-attention + bicross) against the engine with bf16 attention everywhere: the
            # quantity the stated tolerance is written on
            both = parity.check(f"full_depth_fp8/wan22/{name}/fp8_dit_and_bicross_attention_vs_bf16_attention_engine/noise_pred",
                                rel_l2(bout.float(), got8[name][0].float()), FP8_ATTENTION_VS_BF16_ATTENTION_TOL)
            print(f"full_depth_fp8/wan22/{name} fp8 DiT + bicross attention vs bf16 attention (fp8 linears in both): noise_pred {both:.2e}", flush=True)

    assert ref_harness.swap_fp8_linears(model, cfg.start_index) == 40 * len(ref_harness.FP8_SITES)
    for name, ins in inputs.items():
        tag = f"full_depth_fp8/wan22/{name}"
        w8 = _reference_forward(model, cfg, ins, False)             # the reference with ITS fp8 linear in those modules
        (wout, _, wcap), (w32out, _, w32cap) = w8, want32[name]
        (gout, _, gcap), (aout, _, acap) = got8[name], got8a[name]
        assert torch.isfinite(gout.float()).all() and torch.isfinite(aout.float()).all()
        rows = {"noise_pred": parity.check(f"{tag}/fp8_linear_engine_vs_reference_with_fp8_linears/noise_pred",
                                           rel_l2(gout.float(), wout.float()), FP8_OUT_TOL)}
        parity.note(f"{tag}/reference_with_fp8_linears_vs_reference_fp32/noise_pred", rel_l2(wout.float(), w32out.float()))
        parity.note(f"{tag}/fp8_linear_engine_vs_reference_fp32/noise_pred", rel_l2(gout.float(), w32out.float()))
        att = {"noise_pred": parity.check(f"{tag}/fp8_attention_vs_bf16_attention_engine/noise_pred",
                                          rel_l2(aout.float(), gout.float()), FP8_ATTENTION_VS_BF16_ATTENTION_TOL)}
        parity.note(f"{tag}/fp8_attention_engine_vs_reference_with_fp8_linears__unpinned/noise_pred", rel_l2(aout.float(), wout.float()))
        for b, what in WATCH.items():
            rows[f"x@{b}"] = parity.check(f"{tag}/fp8_linear_engine_vs_reference_with_fp8_linears/x_stream_after_block_{b}_{what}",
                                          rel_l2(gcap["x"][b], wcap["x"][b]), FP8_STREAM_TOL)
            parity.note(f"{tag}/reference_with_fp8_linears_vs_reference_fp32/x_stream_after_block_{b}_{what}",
                        rel_l2(wcap["x"][b], w32cap["x"][b]))
            att[f"x@{b}"] = parity.check(f"{tag}/fp8_attention_vs_bf16_attention_engine/x_stream_after_block_{b}_{what}",
                                         rel_l2(acap["x"][b], gcap["x"][b]), FP8_ATTENTION_VS_BF16_ATTENTION_TOL)
            if b >= cfg.start_index:
                w = wcap["tok"][b]
                rows[f"tok@{b}"] = parity.check(f"{tag}/fp8_linear_engine_vs_reference_with_fp8_linears/vggt_stream_after_block_{b}_{what}",
                                                rel_l2(gcap["tok"][b].reshape(w.shape), w), FP8_STREAM_TOL)
                att[f"tok@{b}"] = parity.check(f"{tag}/fp8_attention_vs_bf16_attention_engine/vggt_stream_after_block_{b}_{what}",
                                               rel_l2(acap["tok"][b], gcap["tok"][b]), FP8_ATTENTION_VS_BF16_ATTENTION_TOL)
        print(tag, "fp8 linears vs reference-with-fp8-linears", {k: f"{v:.2e}" for k, v in rows.items()})
        print(tag, "fp8 attention vs bf16 attention (fp8 linears in both)", {k: f"{v:.2e}" for k, v in att.items()})


# ---- the FULL sampling schedule at the benchmarked depth (VERDICT r05 missing 3 / next 4b) ---------------------------------------------
@pytest.mark.skipif(os.environ.get("FW_FULL_DEPTH_SCHEDULE", "0") != "1",
                    reason="opt-in (FW_FULL_DEPTH_SCHEDULE=1): ~10 minutes of fp32 reference time on the box; recorded once per round under profiles/")
def test_full_depth_full_schedule_latents(parity):
    """The quantity north_star's tolerance is written on -- the LATENTS the VAE would decode -- after the reference's own 50-step loop
    (`generate_video`, model_wan21.py:226-324: scheduler, CFG combine with scale 5, Euler updates) at 40 / 24 / 24 blocks on the grid
    whose token count puts every GEMM / attention on the production kernels (L = 8190):
      (1) the reference in fp32 on PyTorch-ROCm                                  -> the truth, latents after EVERY step
      (2) the SAME call on the HIP path (install(), fp32 I/O, bf16 inside)       -> rel-L2 against (1), per step
      (3) the reference in its OWN inference configuration (bf16 weights + autocast, inference_wan21.py:164,310) -> against (1): what the
          reference's users actually run, as the yardstick for (2)
      (4) the HIP path under that bf16 configuration                             -> against (3) and (1).
    Steps via FW_FULL_DEPTH_SCHEDULE_STEPS (default 50).  Recorded per step under full_depth_schedule/*; only a physical bound is
    enforced (a 50-step CFG loop amplifies a forward's 3e-3 by ~6.4 per step before the schedule's shrinking step sizes average it)."""
    from fantasy_world_amd import install, uninstall, synth
    from fantasy_world_amd.hip_ops import HipOps
    from oracle import ref_harness
    steps = int(os.environ.get("FW_FULL_DEPTH_SCHEDULE_STEPS", "50"))
    cfg, model = _build("wan21")
    f, h2, w2 = GRIDS[YARDSTICK_GRID]
    ins = synth.make_inputs(cfg, f, h2, w2, seed=11, device=DEV, dtype=torch.float32)
    frames = 4 * (f - 1) + 1
    kw = dict(context_pos=ins["context"], context_neg=ins["context_neg"], clip_feature=ins["clip_feature"], y=ins["y"], height=8 * h2,
              width=8 * w2, num_frames=frames, num_inference_steps=steps, cfg_scale=5.0, seed=0, device=DEV,
              plucker_embedding=synth.make_plucker(frames, 8 * h2, 8 * w2).to(DEV))

    def run(model, kw, autocast=False):
        """generate_video with the latents after every scheduler.step recorded (a wrapper on the scheduler OBJECT; the loop is untouched)."""
        import contextlib
        import time
        rec, sched = [], model.pipe.scheduler
        orig = sched.step
        sched.step = lambda *a, **k: (lambda o: (rec.append(o.detach().float().cpu()), o)[1])(orig(*a, **k))
        t0 = time.time()
        try:
            with (torch.autocast("cuda", dtype=torch.bfloat16) if autocast else contextlib.nullcontext()), ref_harness.sdpa_by_head_chunks():
                lat, pred = model.generate_video(**kw)
        finally:
            del sched.step
        _sync()
        assert len(rec) == steps and torch.equal(rec[-1], lat.detach().float().cpu())
        return rec, time.time() - t0

    ref32, t_ref32 = run(model, kw)                                                   # (1)
    eng = install(model, ops=HipOps(DEV))
    try:
        hip32, t_hip32 = run(model, kw)                                               # (2)
    finally:
        uninstall(model)
    del eng
    torch.cuda.empty_cache()
    model.to(device=DEV, dtype=torch.bfloat16)
    model.pipe.device, model.pipe.torch_dtype, model.device = DEV, torch.bfloat16, DEV
    kwb = {k: (v.to(torch.bfloat16) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in kw.items()}
    ref16, t_ref16 = run(model, kwb, autocast=True)                                   # (3)
    eng = install(model, ops=HipOps(DEV))
    try:
        hip16, t_hip16 = run(model, kwb, autocast=True)                               # (4)
    finally:
        uninstall(model)
    curves = {"hip_fp32_io_vs_reference_fp32": [rel_l2(a, b) for a, b in zip(hip32, ref32)],
              "reference_bf16_config_vs_reference_fp32": [rel_l2(a, b) for a, b in zip(ref16, ref32)],
              "hip_bf16_config_vs_reference_bf16_config": [rel_l2(a, b) for a, b in zip(hip16, ref16)],
              "hip_bf16_config_vs_reference_fp32": [rel_l2(a, b) for a, b in zip(hip16, ref32)]}
    tag = f"full_depth_schedule/wan21/{YARDSTICK_GRID}/{steps}_steps"
    for k, v in curves.items():
        parity.note(f"{tag}/latents_rel_l2_per_step/{k}", v)
        print(tag, k, " ".join(f"{i + 1}:{v[i]:.2e}" for i in sorted(set([0, 1, 4, 9, 19, 29, 39, steps - 2, steps - 1])) if 0 <= i < steps), flush=True)
    parity.note(f"{tag}/wall_seconds", {"reference_fp32": t_ref32, "hip_fp32_io": t_hip32, "reference_bf16_config": t_ref16, "hip_bf16_config": t_hip16})
    parity.check(f"{tag}/final_latents/hip_fp32_io_vs_reference_fp32", curves["hip_fp32_io_vs_reference_fp32"][-1], 1e-1)
    # the claim that matters for a user of the reference: on the quantity the VAE decodes, the HIP path is CLOSER to the fp32 truth than
    # the reference's own inference configuration is
    assert curves["hip_fp32_io_vs_reference_fp32"][-1] < curves["reference_bf16_config_vs_reference_fp32"][-1], \
        (curves["hip_fp32_io_vs_reference_fp32"][-1], curves["reference_bf16_config_vs_reference_fp32"][-1])
