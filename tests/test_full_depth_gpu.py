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
additionally ONE forward at the headline grid itself (L = 32 760; the reference's fp32 attention fits the 288 GB device next to 74 GB of
fp32 weights and 36 GB of packed ones: measured 2.97e-3, profiles/r04/parity.json).
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
    try:
        with torch.no_grad():
            out, pred = model.joint_forward(ins["x"], **_kwargs(cfg, ins, return_prediction))
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


GRIDS = {"small_f3_12x16": (3, 12, 16), "production_kernels_f21_30x52": (21, 30, 52)}


@pytest.mark.parametrize("flavour", ["wan21", "wan22"])
def test_full_depth_model_matches_reference(flavour, parity):
    from fantasy_world_amd import install, uninstall, synth
    from fantasy_world_amd.hip_ops import HipOps
    cfg, model = _build(flavour)
    grids = dict(GRIDS)
    # the benchmarked model AT the benchmarked grid (Wan2.1: ~40 s of reference time on the box; FW_FULL_DEPTH_HEADLINE=0 skips it,
    # =1 adds it for the Wan2.2 flavour too)
    if os.environ.get("FW_FULL_DEPTH_HEADLINE", "1" if flavour == "wan21" else "0") == "1":
        grids["headline_f21_60x104"] = (21, 60, 104)
    inputs, want = {}, {}
    for name, (f, h2, w2) in grids.items():
        ins = synth.make_inputs(cfg, f, h2, w2, seed=11, device=DEV, dtype=torch.float32)
        with_pred = name.startswith("small")
        try:
            want[name] = _reference_forward(model, cfg, ins, with_pred)                  # the reference, fp32, PyTorch-ROCm
        except torch.OutOfMemoryError as e:                                                # headline grid: the checker's fp32 attention
            if not name.startswith("headline"):
                raise
            parity.note(f"full_depth/{flavour}/{name}/checker_did_not_fit", str(e)[:160])
            torch.cuda.empty_cache()
            continue
        inputs[name] = ins
    eng = install(model, ops=HipOps(DEV), merge_cfg=False)
    assert eng.heads_cfg is not None and list(eng.heads_cfg.layer_idx) == [23, 17, 11, 7]
    assert [eng.cfg.has_adapter(b) for b in range(40)] == [flavour == "wan21" and b <= 24 for b in range(40)]
    try:
        for name, ins in inputs.items():
            with_pred = name.startswith("small")
            got = _hip_forward(model, eng, cfg, ins, with_pred)
            _compare(f"full_depth/{flavour}/{name}", parity, want.pop(name), got, cfg, with_pred)
            # and through the rebound method itself (B1), the call the reference's loop makes
            out, pred = model.joint_forward(ins["x"], **_kwargs(cfg, ins, False))
            torch.cuda.synchronize()
            assert pred is None and torch.equal(out, got[0])
    finally:
        uninstall(model)
