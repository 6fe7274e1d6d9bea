"""-m gpu: BASELINE configs[4] at ITS size -- the Wan2.2 flavour on the 121-frame x 720 x 1280 grid (latents [1,16,31,90,160], L = 111 600
DiT tokens, L2 = 111 755 VGGT tokens) -- as parity numbers instead of "ran, finite" (VERDICT r03 "J2").

Checker: the REAL reference (oracle/_ref bundle) -- a 2-block Wan2.2 FantasyWorldFusionModel (1 preconditioning block + 1 IRG block:
every kind of block the 40-block model has; depth is pinned separately, tests/test_full_depth_gpu.py) -- on PyTorch-ROCm in fp32, which
reproduces its own CPU golden to 1.3e-6 on this box (`ref_on_gpu/wan22/reference_rocm_fp32_vs_golden`).  Three comparisons:
  * bf16 engine against the fp32 reference (the same plateau as at every other size is the expectation);
  * fp8-linear engine (`precision="fp8"`) against the reference with the SAME linears computed by the reference's fp8 linear: the modules
    `enable_vram_management(module_map={nn.Linear: AutoWrappedLinear}, computation_dtype=float8_e4m3fn)` would swap
    (diffsynth_wan22/vram_management/layers.py:113-166) are swapped here for a module whose forward is `AutoWrappedLinear.fp8_linear`
    written by its definition (oracle/fw_oracle.py:fp8_linear -- bit-identical to the real torch._scaled_mm call,
    tests/test_reference_on_gpu.py::test_fp8_linear_against_the_real_scaled_mm; the real call refuses fp32 activations);
  * fp8 attention (`fp8_attention=True`) has no reference semantics (the reference defines fp8 for linears only): its distance from the
    fp8-linear checker is RECORDED, under the same physical bound as at config-1 size.

Round 6: this LIVE-reference test is opt-in -- FW_CONFIG5_LIVE=1 runs the bf16 leg (ONE fp32 reference forward at
L = 111 600 measured 315 s on the box: the reference's fp32 attention over 111 600 keys, not the second forward, is what cost the
suite its wall clock), FW_CONFIG5_FP8=1 adds the reference forward with the fp8 linear in its modules and the two fp8 engines;
both are recorded once per round under profiles/rNN/.  The DEFAULT run pins the bf16 engine at this grid against the same
reference's fp32 forward computed once on CPU and committed as a golden (tests/golden/wan22_cfg5_l2_f31_90x160.pt, oracle/make_golden.py;
tests/test_joint_forward_gpu.py::test_full_size_forward_matches_reference_golden, ~15 s); the fp8-linear arithmetic stays pinned at the
benchmarked DEPTH by tests/test_full_depth_gpu.py::test_full_depth_fp8_linears_and_fp8_attention and at config-1 size by tests/test_fp8_gpu.py.
"""
import os

import pytest
import torch

from conftest import rel_l2
from oracle import ref_locate

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not ref_locate.available(), reason="reference not mounted / staged"),
              pytest.mark.skipif(os.environ.get("FW_CONFIG5_LIVE", "0") != "1" and os.environ.get("FW_CONFIG5_FP8", "0") != "1",
                                 reason="opt-in (FW_CONFIG5_LIVE=1 / FW_CONFIG5_FP8=1): 5 minutes of fp32 reference attention at L = 111 600; "
                                        "the default run uses the committed golden of the same forward")]

DEV = "cuda:0"


def test_config5_grid_against_the_reference(parity):
    from fantasy_world_amd import config as fwc, synth, install, uninstall
    from fantasy_world_amd.hip_ops import HipOps
    from oracle import ref_harness
    cfg = fwc.plumbing22(num_layers=2, start_index=1)
    weights = synth.LazyWeights(synth.weight_spec(cfg), device=DEV)
    model = ref_harness.build_reference_wan22(cfg, weights=weights, heads_cfg=fwc.HeadsConfig.e2e_small())
    model.to(device=DEV, dtype=torch.float32)
    model.pipe.device, model.pipe.torch_dtype, model.device = DEV, torch.float32, DEV
    f, h2, w2 = 31, 90, 160
    ins = synth.make_inputs(cfg, f, h2, w2, seed=21, device=DEV, dtype=torch.float32)
    L = f * (h2 // 2) * (w2 // 2)
    assert L == 111600
    kw = dict(timestep=ins["timestep"], context=ins["context"], y=ins["y"], use_gradient_checkpointing=False, camera_token=None,
              control_camera_latents_input=ins["control_camera_latents_input"], uncond=False, return_prediction=False)
    ops = HipOps(DEV)
    with torch.no_grad():
        want32, _ = model.joint_forward(ins["x"], **kw)                        # the reference, fp32, PyTorch-ROCm
    torch.cuda.synchronize()

    fp8_legs = os.environ.get("FW_CONFIG5_FP8", "0") == "1"
    legs = [("bf16", {})]
    if fp8_legs:      # fp8 attention through the SAME boundary (round 6: install(precision="fp8", fp8_attention=True))
        legs += [("fp8_linears", dict(precision="fp8")), ("fp8_all", dict(precision="fp8", fp8_attention=True))]
    got = {}
    for tag, opts in legs:
        eng = install(model, ops=ops, merge_cfg=False, **opts)
        got[tag], _ = model.joint_forward(ins["x"], **kw)
        torch.cuda.synchronize()
        uninstall(model)
        del eng
    torch.cuda.empty_cache()
    tagp = "config5/wan22_l2_f31_90x160"
    assert all(torch.isfinite(t.float()).all() for t in got.values())
    parity.check(f"{tagp}/bf16_engine_vs_reference_fp32", rel_l2(got["bf16"], want32), 8e-3)
    if not fp8_legs:
        return

    assert ref_harness.swap_fp8_linears(model, cfg.start_index) == 2 * len(ref_harness.FP8_SITES)
    with torch.no_grad():
        want8, _ = model.joint_forward(ins["x"], **kw)                         # the reference with the fp8 linear in those modules
    torch.cuda.synchronize()

    e = {"fp8_linears_vs_ref_fp8": rel_l2(got["fp8_linears"], want8),
         "fp8_linears_vs_ref_fp32": rel_l2(got["fp8_linears"], want32), "ref_fp8_vs_ref_fp32": rel_l2(want8, want32),
         "fp8_attention_vs_ref_fp8": rel_l2(got["fp8_all"], want8)}
    print(tagp, {k: f"{v:.2e}" for k, v in e.items()})
    parity.check(f"{tagp}/fp8_linear_engine_vs_reference_with_fp8_linears", e["fp8_linears_vs_ref_fp8"], 2e-2)
    parity.note(f"{tagp}/reference_with_fp8_linears_vs_reference_fp32", e["ref_fp8_vs_ref_fp32"])
    parity.check(f"{tagp}/fp8_linear_engine_vs_reference_fp32", e["fp8_linears_vs_ref_fp32"], 1e-1)
    # parity UNPINNED by construction: recorded, bounded physically only
    parity.check(f"{tagp}/fp8_attention_engine_vs_reference_with_fp8_linears__unpinned", e["fp8_attention_vs_ref_fp8"], 1e-1)
