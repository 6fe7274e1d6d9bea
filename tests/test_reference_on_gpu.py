"""The REAL reference on the MI355X box: its unmodified Python modules arrive as the git-ignored bundle
oracle/_ref/reference_py.tgz (oracle/stage_ref.sh, oracle/ref_locate.py) and run here as TEST INFRASTRUCTURE --

  * as the caller:  FantasyWorldFusionModel.generate_video (model_wan21.py:226-324) / joint_forward (:104-224), unchanged, on
    top of install() (boundary B1) and install_blocks() (boundary B2) with the HIP op set;
  * as the checker: the same reference modules on PyTorch-ROCm in fp32, and in the reference's own bf16-autocast inference
    configuration (inference_wan21.py:310) as the yardstick; AutoWrappedLinear.fp8_linear with the real torch._scaled_mm
    (diffsynth_wan22/vram_management/layers.py:115-151).

Skipped when the bundle is absent.  The product never imports oracle/ (tests/test_abi.py)."""
import pytest
import torch

from conftest import rel_l2, PRED_KEYS
from oracle import ref_locate

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not ref_locate.available(), reason="reference not mounted / staged")]

DEV = "cuda:0"


def _to_dev(model, dtype):
    model.to(device=DEV, dtype=dtype)
    model.pipe.device, model.pipe.torch_dtype, model.device = DEV, dtype, DEV
    return model


def _gen_kwargs(case, steps):
    from fantasy_world_amd import synth
    f, h2, w2 = case.grid
    ins = case.inputs
    frames = 4 * (f - 1) + 1
    return dict(context_pos=ins["context"].to(DEV), context_neg=ins["context_neg"].to(DEV), clip_feature=ins["clip_feature"],
                y=ins["y"], height=8 * h2, width=8 * w2, num_frames=frames, num_inference_steps=steps, cfg_scale=5.0, seed=0,
                device=DEV, plucker_embedding=synth.make_plucker(frames, 8 * h2, 8 * w2).to(DEV))


def _cast(kw, dtype):
    return {k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in kw.items()}


@pytest.mark.parametrize("steps", [3, 5])
def test_reference_generate_video_runs_on_hip_path(case_pred, parity, steps):
    """The reference's own sampling loop (scheduler, CFG combine, `latents.to('cuda')`, return_prediction on the last step)
    around the rebound joint_forward + get_pose_fea, against the reference itself on PyTorch-ROCm fp32; the reference's
    bf16-autocast configuration (what inference_wan21.py runs) is measured beside it as the yardstick.  3 steps: learn the CFG pair,
    one merged pass, the last step with the prediction; 5 steps: CfgPairing's STEADY state (steps 2-4 all merged passes answered from
    the stash) is on the path before the last step."""
    from oracle import ref_harness
    from fantasy_world_amd import install, uninstall, synth
    from fantasy_world_amd.hip_ops import HipOps
    c = case_pred
    W = dict(c.weights.items())
    W.update(synth.make_pose_encoder_weights())
    model = _to_dev(ref_harness.build_reference_wan21(c.cfg, weights=W, heads_cfg=c.hc), torch.float32)
    assert not model._fw_unused
    kw = _gen_kwargs(c, steps=steps)        # step 1 learns the CFG pair, steps 2.. run the merged pass, the last step returns the prediction
    sfx = "" if steps == 3 else f"/{steps}_steps"
    want, wpred = model.generate_video(**kw)                               # the reference, fp32, PyTorch-ROCm kernels
    torch.cuda.synchronize()

    eng = install(model, ops=HipOps(DEV))
    passes = []
    orig = eng._forward
    eng._forward = lambda x, t, contexts, *a, **k: (passes.append(len(contexts)), orig(x, t, contexts, *a, **k))[1]
    got, pred = model.generate_video(**kw)                                 # the SAME call on the HIP path
    torch.cuda.synchronize()
    uninstall(model)
    assert passes == [1, 1] + [2] * (steps - 1), passes                    # two plain forwards, then ONE merged pass per step
    assert got.shape == want.shape and got.dtype == want.dtype and eng.heads_cfg is not None
    e_lat = rel_l2(got, want)

    # yardstick: the reference in its own inference configuration (bf16 weights + autocast, inference_wan21.py:164,310)
    _to_dev(model, torch.bfloat16)
    kwb = _cast(kw, torch.bfloat16)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ref16, _ = model.generate_video(**kwb)
    install(model, ops=HipOps(DEV))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        got16, _ = model.generate_video(**kwb)                             # exactly what a user of the reference would run
    torch.cuda.synchronize()
    uninstall(model)
    assert got16.dtype == ref16.dtype == torch.bfloat16
    e_ref16_vs_fp32, e_16 = rel_l2(ref16.float(), want), rel_l2(got16.float(), ref16.float())
    parity.note("ref_on_gpu/generate_video/latents_reference_bf16_config_vs_reference_fp32" + sfx, e_ref16_vs_fp32)
    print(f"latents: HIP vs reference, fp32 I/O {e_lat:.2e}; HIP vs reference, both in the bf16 inference configuration {e_16:.2e}; "
          f"reference bf16 configuration vs reference fp32 {e_ref16_vs_fp32:.2e}")
    # Two sampling steps: the CFG combine neg + 5 (pos - neg) multiplies a forward's relative error by ~sqrt(5^2 + 4^2) = 6.4 and the
    # 2-step schedule's last update has sigma 0.83, so the latents carry ~1e-2 of the forwards' 2.7e-3.  Like against like: fp32 I/O
    # against the reference in fp32; the bf16 inference configuration against the reference in ITS bf16 configuration (which
    # rounds the timestep to bf16, model_wan21.py:292-293: 833.3 -> 832, worth 1e-1 on the latents against the fp32 run -- a
    # property of the reference, reproduced, not an error of either side).
    parity.check("ref_on_gpu/generate_video/latents_hip_vs_ref_fp32" + sfx, e_lat, 4e-2)
    parity.check("ref_on_gpu/generate_video/latents_hip_vs_ref_both_bf16_config" + sfx, e_16, 4e-2)
    # the prediction of the reference on a GPU is computed under ITS bf16 autocast (vggt.py:136): both sides carry bf16 noise, on
    # top of the latents' divergence above
    for k in PRED_KEYS:
        assert pred[k].shape == wpred[k].shape
        parity.check(f"ref_on_gpu/generate_video/{k}" + sfx, rel_l2(pred[k].float(), wpred[k].float()), 6e-2)


def test_install_blocks_under_reference_joint_forward_on_hip(case_depth, parity):
    """Boundary B2 on the GPU: every DiTBlock / VGGT Block / IRGBlock forward of the real 8-block model rebound
    (fantasy_world_amd.blocks), the reference's own joint_forward walks its loops; the streams after EVERY block against the
    per-block depth golden (fp32 reference).  Block-granular surfaces pass the DiT stream in the caller's dtype: fp32 here."""
    from oracle import ref_harness
    from fantasy_world_amd.blocks import install_blocks
    from fantasy_world_amd.hip_ops import HipOps
    c, g = case_depth, case_depth.golden
    model = _to_dev(ref_harness.build_reference_wan21(c.cfg, weights=c.weights), torch.float32)
    undo = install_blocks(model, ops=HipOps(DEV))
    cap = {"x": {}, "tok": {}}
    rd, ra = g["rows_dit"].to(DEV), g["rows_agg"].to(DEV)
    for b in range(c.cfg.start_index):
        model.pipe.dit.blocks[b].register_forward_hook(lambda m, a, out, b=b: cap["x"].__setitem__(b, out[0][rd].float()))
    for j in range(c.cfg.n_irg):
        def hook(m, a, out, j=j):
            cap["x"][c.cfg.start_index + j] = out[0][0][rd].float()
            cap["tok"][j] = out[1][0][ra].float()
        model.IRGBlock[j].register_forward_hook(hook)
    ins = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in c.inputs.items()}
    with torch.no_grad():
        out, pred = model.joint_forward(ins["x"], timestep=ins["timestep"], context=ins["context"], clip_feature=ins["clip_feature"],
                                        y=ins["y"], use_gradient_checkpointing=False, camera_token=None,
                                        plucker_fea=ins["plucker_fea"], plucker_context_lens=ins["plucker_context_lens"],
                                        return_prediction=False)
    torch.cuda.synchronize()
    undo()
    ex = [rel_l2(cap["x"][b], g["x_blocks"][b]) for b in range(c.cfg.num_layers)]
    et = [rel_l2(cap["tok"][j], g["tok_blocks"][j]) for j in range(c.cfg.n_irg)]
    print("B2 x per block  ", [f"{v:.2e}" for v in ex])
    print("B2 tok per block", [f"{v:.2e}" for v in et])
    parity.note("ref_on_gpu/b2/x_stream_rel_l2_per_block", ex)
    parity.note("ref_on_gpu/b2/vggt_stream_rel_l2_per_block", et)
    parity.check("ref_on_gpu/b2/noise_pred", rel_l2(out.float(), g["noise_pred"]), 1.6e-2)
    parity.check("ref_on_gpu/b2/x_stream_last_block", ex[-1], 1.6e-2)
    parity.check("ref_on_gpu/b2/vggt_stream_last_block", et[-1], 1.6e-2)


def test_install_on_reference_wan22_model_on_hip(case_w22, parity):
    """B1 with the M22 signature (model_wan22.py:231-242) on the GPU: the real Wan2.2-flavour model, install(), its golden."""
    from oracle import ref_harness
    from fantasy_world_amd import install, uninstall
    from fantasy_world_amd.hip_ops import HipOps
    c = case_w22
    model = _to_dev(ref_harness.build_reference_wan22(c.cfg, weights=c.weights), torch.float32)
    ins = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in c.inputs.items()}
    kw = dict(timestep=ins["timestep"], context=ins["context"], y=ins["y"], use_gradient_checkpointing=False, camera_token=None,
              control_camera_latents_input=ins["control_camera_latents_input"], uncond=False, return_prediction=False)
    with torch.no_grad():
        ref_gpu, _ = model.joint_forward(ins["x"], **kw)                    # the reference itself on PyTorch-ROCm, fp32
    install(model, ops=HipOps(DEV))
    got, p = model.joint_forward(ins["x"], **kw)
    torch.cuda.synchronize()
    uninstall(model)
    assert p is None and got.shape == ref_gpu.shape
    parity.check("ref_on_gpu/wan22/reference_rocm_fp32_vs_golden", rel_l2(ref_gpu, c.golden["noise_pred"]), 1e-3)
    parity.check("ref_on_gpu/wan22/hip_vs_golden", rel_l2(got.float(), c.golden["noise_pred"]), 8e-3)


def test_reference_wan22_dual_expert_loop_runs_on_hip_path(parity):
    """The Wan2.2 sampler's own loop (inference_wan22.py:164-283 generate_video_with_dual_models, unmodified; boundary 900: two
    steps on the high-noise expert, two on the low-noise one, return_prediction on the last) with install() on BOTH experts and
    the HIP op set, against the same loop on the reference itself (PyTorch-ROCm, fp32).  Two engines side by side on one device;
    each learns its CFG pair on its first step and runs the merged pass on its second (the last one with the prediction)."""
    from oracle import ref_harness
    from fantasy_world_amd import install, uninstall
    from fantasy_world_amd.hip_ops import HipOps
    from test_install_dropin import _wan22_two_experts
    high, low, kw = _wan22_two_experts(
        lambda cfg, w, hc: _to_dev(ref_harness.build_reference_wan22(cfg, weights=w, heads_cfg=hc), torch.float32))
    kw = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in kw.items()}
    sampler = ref_harness.build_reference_wan22_sampler(high, low, seed=3, cfg_scale=5.0, timestep_boundary=900, device=DEV)
    with torch.no_grad():
        want, wpred = sampler.generate_video_with_dual_models(**kw)
    torch.cuda.synchronize()
    ops = HipOps(DEV)
    engines = [install(m, ops=ops) for m in (high, low)]
    passes = ([], [])
    for eng, log in zip(engines, passes):
        orig = eng._forward
        eng._forward = (lambda orig, log: lambda x, t, contexts, *a, **k: (log.append(len(contexts)), orig(x, t, contexts, *a, **k))[1])(orig, log)
    got, pred = sampler.generate_video_with_dual_models(**kw)
    torch.cuda.synchronize()
    for m in (high, low):
        uninstall(m)
    assert passes == ([1, 1, 2], [1, 1, 2]), passes
    assert got.shape == want.shape and got.dtype == want.dtype
    # four CFG-combined steps carry ~1e-2 of the forwards' 2.7e-3 (see test_reference_generate_video_runs_on_hip_path)
    parity.check("ref_on_gpu/wan22_dual_expert_loop/latents_hip_vs_ref_fp32", rel_l2(got, want), 4e-2)
    for k in PRED_KEYS:
        assert pred[k].shape == wpred[k].shape
        parity.check(f"ref_on_gpu/wan22_dual_expert_loop/{k}", rel_l2(pred[k].float(), wpred[k].float()), 6e-2)


@pytest.mark.parametrize("M,N,K,amp", [(256, 384, 512, 3.0), (4096, 5120, 5120, 1.0), (2048, 13824, 5120, 30.0)])
def test_fp8_linear_against_the_real_scaled_mm(M, N, K, amp, parity):
    """A19 with the REAL checker: AutoWrappedLinear.forward -> fp8_linear -> torch._scaled_mm (layers.py:115-151,154-166) on this
    GPU against fw_fp8_quant_rows + fw_gemm_fp8.  Same e4m3 operands on both sides (bit-exact quantiser, tests/test_fp8_gpu.py),
    fp32 accumulation, one bf16 rounding of the output: the results may differ by summation order only."""
    from oracle import ref_harness
    from fantasy_world_amd.hip_ops import HipOps
    ref_harness.install_stubs()
    from FantasyWorld.diffsynth_wan22.vram_management.layers import AutoWrappedLinear
    g = torch.Generator().manual_seed(M + N)
    lin = torch.nn.Linear(K, N)
    lin.weight.data = torch.randn(N, K, generator=g) * K ** -0.5
    lin.bias.data = torch.randn(N, generator=g) * 0.1
    lin = lin.to(device=DEV, dtype=torch.bfloat16)
    wrapped = AutoWrappedLinear(lin, offload_dtype=torch.bfloat16, offload_device=DEV, onload_dtype=torch.bfloat16,
                                onload_device=DEV, computation_dtype=torch.float8_e4m3fn, computation_device=DEV, vram_limit=None)
    x = (torch.randn(M, K, generator=g) * amp).to(device=DEV, dtype=torch.bfloat16)
    x[0, 0] = 1000.0 * amp
    try:
        with torch.no_grad():
            want = wrapped(x)
    except (RuntimeError, NotImplementedError) as e:        # a torch build whose _scaled_mm lacks row-wise e4m3fn scaling here
        pytest.skip(f"torch._scaled_mm unavailable for this call on this box: {str(e)[:200]}")
    ops = HipOps(DEV)
    got = ops.linear_fp8(x, ops.pack_linear_fp8(lin.weight.float(), lin.bias.float()))
    torch.cuda.synchronize()
    assert want.dtype == torch.bfloat16 and got.shape == want.shape
    same = (got == want).float().mean().item()
    parity.note(f"ref_on_gpu/fp8/scaled_mm_{M}x{N}x{K}/bit_identical_fraction", same)
    parity.check(f"ref_on_gpu/fp8/scaled_mm_{M}x{N}x{K}", rel_l2(got.float(), want.float()), 3e-3)
    assert same > 0.9, same


def test_flash_attention_hook_on_real_dit_block_on_hip(parity):
    """B3 on the GPU with the real module: wan_video_dit.flash_attention rebound, the reference's own DiTBlock (bf16, as under
    the inference autocast) calls it three times (self, text, image); against the same block through SDPA."""
    from oracle import ref_harness, fw_oracle
    from fantasy_world_amd import install_flash_attention
    from fantasy_world_amd.hip_ops import HipOps
    ref_harness.install_stubs()
    import FantasyWorld.diffsynth_wan21.models.wan_video_dit as dit
    torch.manual_seed(0)
    D, H = 512, 4
    blk = dit.DiTBlock(True, D, H, 1024).eval().to(device=DEV, dtype=torch.bfloat16)
    g = torch.Generator().manual_seed(1)
    f, h, w = 3, 8, 12
    L = f * h * w
    x = torch.randn(1, L, D, generator=g).to(DEV, torch.bfloat16)
    ctx = torch.randn(1, 257 + 64, D, generator=g).to(DEV, torch.bfloat16)
    t_mod = (torch.randn(1, 6, D, generator=g) * 0.2).to(DEV, torch.bfloat16)
    freqs = fw_oracle.expand_freqs(fw_oracle.precompute_freqs_cis_3d(D // H), f, h, w).to(DEV)
    with torch.no_grad():
        want = blk(x, ctx, t_mod, freqs)
        ref32 = blk.float()(x.float(), ctx.float(), t_mod.float(), freqs)
        blk.to(torch.bfloat16)
    calls = []
    ops = HipOps(DEV)
    orig = ops.attention
    ops.attention = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    undo = install_flash_attention([dit], ops=ops)
    try:
        with torch.no_grad():
            got = blk(x, ctx, t_mod, freqs)
    finally:
        undo()
    torch.cuda.synchronize()
    assert len(calls) == 3 and got.dtype == want.dtype
    e_hook, e_sdpa = rel_l2(got.float(), ref32), rel_l2(want.float(), ref32)
    parity.check("ref_on_gpu/b3/dit_block_flash_attention_hook_vs_fp32", e_hook, 2e-2)
    parity.note("ref_on_gpu/b3/dit_block_sdpa_bf16_vs_fp32", e_sdpa)
    assert e_hook < 1.5 * e_sdpa + 2e-3
