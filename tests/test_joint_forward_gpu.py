"""-m gpu: the whole HIP joint_forward (through the C ABI) against the reference goldens and the CPU oracle.

Tolerance: activations between kernels are bf16 (the reference's own inference dtype), accumulation / statistics /
residual streams fp32.  The reference's own bf16-vs-fp32 deviation is 3.2e-3 after ONE DiT block;
emulating exactly our rounding points on CPU gives 2.6e-3 end-to-end on this 2-block model
(tests/test_engine_cpu.py::test_engine_bf16_emulation_yardstick).  The end-to-end bound is therefore 8e-3 relative L2
against the fp32 reference; the <= 1e-3 north-star bound is enforced per kernel in tests/test_hip_ops.py.
"""
import pytest
import torch

from conftest import rel_l2, forward_kwargs

pytestmark = pytest.mark.gpu

E2E_TOL = 8e-3
DEPTH_TOL = 1.6e-2        # 8 blocks; the reference's own bf16-vs-fp32 gap is 1.2e-2 after 8 blocks
# per-(case, tensor) bounds = 1.5 x the value measured on MI355X (profiles/r02/parity.json); anything not listed: E2E_TOL
E2E_BOUNDS = {}


def _run_hip(case, **over):
    from fantasy_world_amd.engine import FusionEngine
    from fantasy_world_amd.hip_ops import HipOps
    ops = HipOps("cuda:0")
    eng = FusionEngine(case.cfg, case.weights.__getitem__, ops)
    ins = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in case.inputs.items()}
    ins.update(over)
    col = {}
    kw = forward_kwargs(case, "cuda")
    kw.update({k: v for k, v in over.items() if k in kw})
    out, pred = eng.joint_forward(ins["x"], ins["timestep"], ins["context"], collect=col, **kw)
    torch.cuda.synchronize()
    assert pred is None
    col["noise_pred"] = out
    return col, eng


@pytest.mark.parametrize("case_name", ["case_l2", "case_l3", "case_w22", "case_camtok"])
def test_hip_joint_forward_matches_reference_golden(case_name, request, parity):
    case = request.getfixturevalue(case_name)
    col, _ = _run_hip(case)
    errs = {k: rel_l2(col[k].float().reshape(case.golden[k].shape), case.golden[k])
            for k in ("x_after_pcb", "x_final", "tokens_final", "noise_pred")}
    print(case.name, {k: f"{v:.2e}" for k, v in errs.items()})
    for k, v in errs.items():
        parity.check(f"e2e/{case.name}/{k}", v, E2E_BOUNDS.get((case.name, k), E2E_TOL))
    assert torch.isfinite(col["noise_pred"].float()).all()


def test_hip_joint_forward_config1_golden(case_cfg1, parity):
    """BASELINE.json configs[0] (2-block model, latents [1,16,9,64,64], L = 9216, L2 = 9261) against the REAL reference's fp32
    joint_forward: the first oracle comparison in which the engine runs what it runs at full size -- the default 256x256
    ping-pong GEMM (M >= 2048, with its fused q|k|v / gate / residual epilogues and the 128-row M-tail peel of the VGGT
    GEMMs), 36 attention query blocks per head with the XCD remap, K/V ring wrap-around, the frame-batched hd-64 attention."""
    case, g = case_cfg1, case_cfg1.golden
    col, _ = _run_hip(case)
    L2 = col["tokens_final"].shape[0]
    errs = {"noise_pred": rel_l2(col["noise_pred"].float(), g["noise_pred"]),
            "x_after_pcb": rel_l2(col["x_after_pcb"][g["rows_dit"].cuda()], g["x_after_pcb"]),
            "x_final": rel_l2(col["x_final"][g["rows_dit"].cuda()], g["x_final"]),
            "tokens_final": rel_l2(col["tokens_final"].reshape(L2, -1)[g["rows_agg"].cuda()], g["tokens_final"])}
    print(case.name, {k: f"{v:.2e}" for k, v in errs.items()})
    for k, v in errs.items():
        parity.check(f"e2e/{case.name}/{k}", v, E2E_BOUNDS.get((case.name, k), E2E_TOL))


def test_hip_error_growth_with_depth(case_depth, parity):
    """4 PCB + 4 IRG blocks: rel-L2 of the bf16-activation HIP path against the fp32 reference after EVERY block (sampled rows
    of both residual streams).  The product runs 40 blocks; this is the measured slope, recorded in parity_gpu.json next to the
    bf16-emulation yardstick of the same host code on CPU (tests/test_engine_cpu.py::test_engine_depth_yardstick)."""
    case, g = case_depth, case_depth.golden
    from fantasy_world_amd.engine import FusionEngine
    from fantasy_world_amd.hip_ops import HipOps
    eng = FusionEngine(case.cfg, case.weights.__getitem__, HipOps("cuda:0"))
    ins = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in case.inputs.items()}
    got = {"x": {}, "tok": {}}
    rd, ra = g["rows_dit"].cuda(), g["rows_agg"].cuda()
    col = {"per_block": lambda kind, i, t: got[kind].__setitem__(i, t[rd if kind == "x" else ra].clone())}
    out, _ = eng.joint_forward(ins["x"], ins["timestep"], ins["context"], collect=col, **forward_kwargs(case, "cuda"))
    torch.cuda.synchronize()
    ex = [rel_l2(got["x"][b], g["x_blocks"][b]) for b in range(case.cfg.num_layers)]
    et = [rel_l2(got["tok"][j], g["tok_blocks"][j]) for j in range(case.cfg.n_irg)]
    print("x per block  ", [f"{v:.2e}" for v in ex])
    print("tok per block", [f"{v:.2e}" for v in et])
    parity.note("depth/x_stream_rel_l2_per_block", ex)
    parity.note("depth/vggt_stream_rel_l2_per_block", et)
    parity.check("depth/noise_pred", rel_l2(out.float(), g["noise_pred"]), DEPTH_TOL)
    parity.check("depth/x_stream_last_block", ex[-1], DEPTH_TOL)
    parity.check("depth/vggt_stream_last_block", et[-1], DEPTH_TOL)
    # random-walk growth: no block may add more than the first one did by a wide margin (a broken block shows as a jump)
    for b in range(1, len(ex)):
        assert ex[b] < ex[b - 1] + 2.5 * ex[0] + 1e-3, (b, ex)


def test_hip_step_invariant_cache_is_bit_identical(case_l2):
    """install() turns the step-invariant cache ON for real generations: on the GPU, with the HIP op set, the cached engine
    must reproduce the uncached one bit for bit over a CFG pair and a second step (cached K/V, context embeddings and the
    adapter's Pluecker term are REUSED tensors; the GEMM that accumulates into its residual operand must not corrupt them)."""
    from fantasy_world_amd.engine import FusionEngine
    from fantasy_world_amd.hip_ops import HipOps
    case = case_l2
    ops = HipOps("cuda:0")
    plain = FusionEngine(case.cfg, case.weights.__getitem__, ops)
    cached = FusionEngine(case.cfg, case.weights.__getitem__, ops, cache_step_invariants=True)
    ins = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in case.inputs.items()}
    kw = forward_kwargs(case, "cuda")
    t2 = ins["timestep"] * 0.5
    for t, ctx in ((ins["timestep"], ins["context"]), (ins["timestep"], ins["context_neg"]), (t2, ins["context"]),
                   (t2, ins["context_neg"]), (ins["timestep"], ins["context"])):
        want, _ = plain.joint_forward(ins["x"], t, ctx, **kw)
        got, _ = cached.joint_forward(ins["x"], t, ctx, **kw)
        torch.cuda.synchronize()
        assert torch.equal(got, want)
    assert cached.invariants.entries and not plain.invariants.entries


def test_hip_fp8_engine_matches_fp8_oracle(case_l2, case_cfg1, parity):
    """BASELINE config 5's arithmetic: FusionEngine(precision="fp8") routes the DiT blocks' linears (q/k/v/o of both attentions,
    FFN) through the fp8 linear, exactly the modules enable_vram_management would swap for AutoWrappedLinear (layers.py:113-151).
    Parity target = the CPU oracle with the SAME linears computed by the reference's fp8_linear (oracle fp8_linears=True).
    Both sides quantise almost identical activations, so they agree far better (bf16-level) than either agrees with the
    full-precision forward (a few percent: the reference's own fp8 trade-off, recorded, not bounded tightly)."""
    from oracle import fw_oracle
    from fantasy_world_amd.engine import FusionEngine
    from fantasy_world_amd.hip_ops import HipOps
    for case in (case_l2, case_cfg1):
        ins = case.inputs
        want = fw_oracle.joint_forward(case.weights, case.cfg, ins["x"], ins["timestep"], ins["context"], ins["clip_feature"],
                                       ins["y"], ins["plucker_fea"], ins["plucker_context_lens"], fp8_linears=True)
        eng = FusionEngine(case.cfg, case.weights.__getitem__, HipOps("cuda:0"), precision="fp8")
        d = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in ins.items()}
        got, _ = eng.joint_forward(d["x"], d["timestep"], d["context"], **forward_kwargs(case, "cuda"))
        torch.cuda.synchronize()
        parity.check(f"fp8/{case.name}/engine_vs_fp8_oracle", rel_l2(got.float(), want), 2.0e-2)
        parity.check(f"fp8/{case.name}/engine_vs_fp32_reference", rel_l2(got.float(), case.golden["noise_pred"]), 1.0e-1)
        parity.note(f"fp8/{case.name}/fp8_oracle_vs_fp32_reference", rel_l2(want, case.golden["noise_pred"]))
        del eng


def test_hip_merged_cfg_pair_equals_two_forwards(case_l2, case_cfg1):
    """CFG batch-2 merge on the GPU (FusionEngine.joint_forward_pair): one pass over 2L rows, attention batch 2 -- every kernel
    computes a row / a (batch, head, q-block) exactly as in the separate forwards, so the results are BIT-identical (small case
    and BASELINE config 1, where the 256x256 GEMMs, multi-block attention and the frame-batched VGGT attention are in play)."""
    from fantasy_world_amd.engine import FusionEngine
    from fantasy_world_amd.hip_ops import HipOps
    ops = HipOps("cuda:0")
    for case in (case_l2, case_cfg1):
        eng = FusionEngine(case.cfg, case.weights.__getitem__, ops, cache_step_invariants=True)
        d = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in case.inputs.items()}
        kw = forward_kwargs(case, "cuda")
        pos, _ = eng.joint_forward(d["x"], d["timestep"], d["context"], **kw)
        neg, _ = eng.joint_forward(d["x"], d["timestep"], d["context_neg"], **kw)
        mp, mn, pred = eng.joint_forward_pair(d["x"], d["timestep"], d["context"], d["context_neg"], **kw)
        torch.cuda.synchronize()
        assert pred is None and torch.equal(mp, pos) and torch.equal(mn, neg), case.name
        assert not torch.equal(pos, neg)
        del eng


def test_hip_denoise_step_matches_oracle(case_l2, parity):
    """A18 on the GPU: one sampling step (2 joint_forward + CFG combine + flow-match Euler update, M21:289-322) through
    sampler.denoise_step against the same step assembled from the CPU oracle's two forwards."""
    from oracle import fw_oracle
    from fantasy_world_amd.engine import FusionEngine
    from fantasy_world_amd.hip_ops import HipOps
    from fantasy_world_amd.sampler import FlowMatchScheduler, denoise_step
    case, ins = case_l2, case_l2.inputs
    sched = FlowMatchScheduler()
    sched.set_timesteps(50)
    step_id = 7
    t = sched.timesteps[step_id].reshape(1)
    fwd = lambda ctx: fw_oracle.joint_forward(case.weights, case.cfg, ins["x"], t, ctx, ins["clip_feature"], ins["y"],
                                              ins["plucker_fea"], ins["plucker_context_lens"])
    pos, neg = fwd(ins["context"]), fwd(ins["context_neg"])
    want = ins["x"] + (neg + 5.0 * (pos - neg)) * float(sched.sigmas[step_id + 1] - sched.sigmas[step_id])
    eng = FusionEngine(case.cfg, case.weights.__getitem__, HipOps("cuda:0"))
    d = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in ins.items()}
    cond = dict(clip_feature=d["clip_feature"], y=d["y"], plucker_fea=d["plucker_fea"],
                plucker_context_lens=d["plucker_context_lens"])
    got, pred = denoise_step(eng, sched, step_id, d["x"], d["context"], d["context_neg"], cond)
    torch.cuda.synchronize()
    assert pred is None and got.shape == want.shape
    # the update is a small multiple of noise_pred added to the latents: compare the UPDATE, not the (dominated) sum
    upd_err = rel_l2(got.float().cpu() - ins["x"], want - ins["x"])
    parity.check("sampler/denoise_step_update", upd_err, 2.0e-2)      # CFG amplifies pos-neg differences 5x
    parity.check("sampler/denoise_step_latents", rel_l2(got.float(), want), 1e-3)


def test_hip_joint_forward_bf16_inputs_and_determinism(case_l2):
    """The inference scripts feed bf16 latents/context and a bf16 timestep (I21:310, M21:292-293); the output comes back
    in the latents' dtype; two runs are bit-identical (no atomics on the path)."""
    case = case_l2
    ins = case.inputs
    over = dict(x=ins["x"].cuda().bfloat16(), y=ins["y"].cuda().bfloat16(), context=ins["context"].cuda().bfloat16(),
                clip_feature=ins["clip_feature"].cuda().bfloat16(), plucker_fea=ins["plucker_fea"].cuda().bfloat16(),
                timestep=ins["timestep"].cuda().bfloat16())
    a, _ = _run_hip(case, **over)
    b, _ = _run_hip(case, **over)
    assert a["noise_pred"].dtype == torch.bfloat16
    assert torch.equal(a["noise_pred"], b["noise_pred"])
    assert rel_l2(a["noise_pred"].float(), case.golden["noise_pred"]) < 1.2e-2   # + bf16 rounding of the output


def test_hip_matches_cpu_oracle_on_negative_prompt_and_uncond(case_l2):
    """Second context draw (the CFG negative pass) and uncond=True (bicross skipped, IRG:70-72) vs the CPU oracle."""
    from oracle import fw_oracle
    case = case_l2
    ins = case.inputs
    for uncond in (False, True):
        want = fw_oracle.joint_forward(case.weights, case.cfg, ins["x"], ins["timestep"], ins["context_neg"],
                                       ins["clip_feature"], ins["y"], ins["plucker_fea"], ins["plucker_context_lens"],
                                       uncond=uncond)
        from fantasy_world_amd.engine import FusionEngine
        from fantasy_world_amd.hip_ops import HipOps
        eng = FusionEngine(case.cfg, case.weights.__getitem__, HipOps("cuda:0"))
        d = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in ins.items()}
        got, _ = eng.joint_forward(d["x"], d["timestep"], d["context_neg"], clip_feature=d["clip_feature"], y=d["y"],
                                   plucker_fea=d["plucker_fea"], plucker_context_lens=d["plucker_context_lens"],
                                   uncond=uncond)
        assert rel_l2(got.float(), want) < E2E_TOL, uncond
        del eng


# (round 6: + BASELINE config 5's grid, 121f x 720p, L = 111 600 -- the reference's CPU fp32 forward as a golden instead of a live
#  315-s fp32 reference forward on the box in every run, tests/test_config5_gpu.py)
FULL_SIZE_GOLDENS = ["wan21_cfg2_l2_f21_60x104", "wan22_cfg4_l2_f21_90x160", "wan21_cli_l2_f21_42x74", "wan22_cfg5_l2_f31_90x160"]


@pytest.mark.parametrize("name", FULL_SIZE_GOLDENS)
def test_full_size_forward_matches_reference_golden(name, parity):
    """BASELINE sizes pinned to the REAL reference (round 3): config 2 / 3's token grid (81f x 480 x 832 -> latents
    [1,16,21,60,104], L = 32760, L2 = 32865) and config 4's (Wan2.2, 81f x 720p -> [1,16,21,90,160], L = 75600, L2 = 75705) on the
    2-block model.  Golden = the reference's own FantasyWorldFusionModel.joint_forward in fp32 on CPU (oracle/make_golden.py
    main_sized: 146 s / ~15 min of reference time): noise_pred in full, the two residual streams as 64 sampled rows.
    First assert: the HIP path against that golden, at the level of the small goldens (2.6-2.8e-3, the bf16 rounding of the
    activations; physical bound 8e-3).  Second assert (kept from round 2): agreement between INDEPENDENT implementations of the
    hot kernels -- the default path (8-wave ping-pong GEMM, log2-domain single-stream attention) against the four-wave GEMM and the
    first-generation attention kernel -- so an addressing error at this size is attributed to a kernel, not to the wiring."""
    import os
    from conftest import Case, GOLDEN_DIR
    if not os.path.exists(os.path.join(GOLDEN_DIR, name + ".pt")):
        pytest.skip(f"{name}.pt not generated (oracle/make_golden.py {name})")
    from fantasy_world_amd.engine import FusionEngine
    from fantasy_world_amd.hip_ops import HipOps
    case = Case(name)
    g = case.golden
    ops = HipOps("cuda:0")
    eng = FusionEngine(case.cfg, case.weights.__getitem__, ops)
    case.weights.clear()
    ins = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in case.inputs.items()}
    kw = forward_kwargs(case, "cuda")
    col = {}
    a, _ = eng.joint_forward(ins["x"], ins["timestep"], ins["context"], collect=col, **kw)
    torch.cuda.synchronize()
    L2 = col["tokens_final"].shape[0]
    rd, ra = g["rows_dit"].cuda(), g["rows_agg"].cuda()
    errs = {"noise_pred": rel_l2(a.float(), g["noise_pred"]),
            "x_after_pcb": rel_l2(col["x_after_pcb"][rd], g["x_after_pcb"]),
            "x_final": rel_l2(col["x_final"][rd], g["x_final"]),
            "tokens_final": rel_l2(col["tokens_final"].reshape(L2, -1)[ra], g["tokens_final"])}
    print(name, {k: f"{v:.2e}" for k, v in errs.items()})
    del col
    for k, v in errs.items():
        parity.check(f"e2e/{name}/{k}", v, E2E_TOL)
    try:
        ops.set_option("gemm_kernel", 5)
        ops.set_option("attn_var", 0)
        b, _ = eng.joint_forward(ins["x"], ins["timestep"], ins["context"], **kw)
        torch.cuda.synchronize()
    finally:
        ops.set_option("gemm_kernel", 9)
        ops.set_option("attn_var", 192)
    assert torch.isfinite(b.float()).all()
    parity.check(f"e2e/{name}/noise_pred_independent_kernels", rel_l2(b.float(), g["noise_pred"]), E2E_TOL)
    # two independent bf16 rounding realisations of the same forward: each ~2.7e-3 from the fp32 truth, so ~sqrt(2) of that apart
    parity.check(f"e2e/{name}/default_vs_independent_kernels", rel_l2(a.float(), b.float()), E2E_TOL)


PRED_TOL = 1.5e-2


def test_hip_joint_forward_return_prediction(case_pred):
    """Last sampling step (M21:303-305): noise_pred and the geometry prediction dict from one HIP joint_forward against the
    REAL reference's joint_forward(return_prediction=True) in fp32.  The heads read bf16-path tokens (a few 1e-3 off after
    two IRG layers) and chain ~25 more bf16 convolutions; 1.5e-2 bounds the sum (measured 8e-4 .. 7.9e-3; world_points goes through sign*expm1)."""
    from conftest import PRED_KEYS
    from fantasy_world_amd.engine import FusionEngine
    from fantasy_world_amd.hip_ops import HipOps
    c = case_pred
    eng = FusionEngine(c.cfg, c.weights.__getitem__, HipOps("cuda:0"), heads_cfg=c.hc)
    ins = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in c.inputs.items()}
    out, pred = eng.joint_forward(ins["x"], ins["timestep"], ins["context"], return_prediction=True, **forward_kwargs(c, "cuda"))
    torch.cuda.synchronize()
    assert rel_l2(out.float(), c.golden["noise_pred"]) < E2E_TOL
    errs = {k: rel_l2(pred[k], c.golden[k]) for k in PRED_KEYS}
    print(c.name, {k: f"{v:.2e}" for k, v in errs.items()})
    for k in PRED_KEYS:
        assert pred[k].shape == c.golden[k].shape and pred[k].dtype == torch.float32
        assert torch.isfinite(pred[k]).all()
        assert errs[k] < PRED_TOL, (k, errs[k])


def test_hip_sequence_shard_over_rccl_single_rank_group(case_l3):
    """The multi-GPU code path on the hardware that is available to a test: a ONE-rank RCCL process group.  Every collective
    of the sequence shard (all_to_all_single with split sizes, all_gather_into_tensor, async work handles waited on the compute
    stream) runs through RCCL on device tensors in bf16 / fp32 and degenerates to a copy, so the sharded engine must
    reproduce the unsharded one bit for bit.  (World sizes 2-4 are covered on CPU over gloo, tests/test_sequence_shard_cpu.py.)"""
    import os
    import socket
    import torch.distributed as dist
    from fantasy_world_amd.engine import FusionEngine
    from fantasy_world_amd.hip_ops import HipOps
    from fantasy_world_amd.parallel import SequenceShard
    case = case_l3
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    created = False
    if not dist.is_initialized():
        try:
            torch.cuda.set_device(0)
            dist.init_process_group("nccl", rank=0, world_size=1)
            created = True
        except Exception as e:                                   # no usable RCCL transport on this box
            pytest.skip(f"RCCL process group unavailable: {e}")
    try:
        ops = HipOps("cuda:0")
        ins = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in case.inputs.items()}
        kw = forward_kwargs(case, "cuda")
        want, _ = FusionEngine(case.cfg, case.weights.__getitem__, ops).joint_forward(ins["x"], ins["timestep"], ins["context"], **kw)
        eng = FusionEngine(case.cfg, case.weights.__getitem__, ops, shard=SequenceShard(0, 1))
        got, _ = eng.joint_forward(ins["x"], ins["timestep"], ins["context"], **kw)
        torch.cuda.synchronize()
        assert torch.equal(got, want)
        del eng
        # round 6: BASELINE config 5's arithmetic through the same transport -- the head exchange then carries e4m3 BYTES (uint8
        # all_to_all_single over RCCL) and the TP-style MAX all-reduce of the fp8 row maxima runs on the device
        opts = dict(precision="fp8", fp8_attention=True)
        want8, _ = FusionEngine(case.cfg, case.weights.__getitem__, ops, **opts).joint_forward(ins["x"], ins["timestep"], ins["context"], **kw)
        got8, _ = FusionEngine(case.cfg, case.weights.__getitem__, ops, shard=SequenceShard(0, 1), **opts).joint_forward(
            ins["x"], ins["timestep"], ins["context"], **kw)
        amax = torch.tensor([1.0, 5.0, 3.0], device="cuda")
        dist.all_reduce(amax, op=dist.ReduceOp.MAX)
        torch.cuda.synchronize()
        assert torch.equal(got8, want8) and amax.tolist() == [1.0, 5.0, 3.0]
    finally:
        if created:
            dist.destroy_process_group()


def test_hip_fp8_attention_engine_is_close_to_the_bf16_engine(case_cfg1, parity):
    """FusionEngine(precision="fp8", fp8_attention=True): BASELINE config 5's arithmetic (fp8 linears + fp8 DiT self-attention) at
    config-1 size.  PARITY UNPINNED for the attention part (no reference semantics): the check is a stated distance to the bf16
    engine and to the reference golden -- fp8 noise (3-bit mantissas on q / k / v / probabilities) through 2 blocks."""
    from fantasy_world_amd.engine import FusionEngine
    from fantasy_world_amd.hip_ops import HipOps
    case = case_cfg1
    ins = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in case.inputs.items()}
    kw = forward_kwargs(case, "cuda")
    outs = {}
    for tag, opts in (("bf16", {}), ("fp8_linears", dict(precision="fp8")), ("fp8_all", dict(precision="fp8", fp8_attention=True))):
        eng = FusionEngine(case.cfg, case.weights.__getitem__, HipOps("cuda:0"), **opts)
        outs[tag], _ = eng.joint_forward(ins["x"], ins["timestep"], ins["context"], **kw)
        del eng
    torch.cuda.synchronize()
    assert torch.isfinite(outs["fp8_all"].float()).all()
    parity.check("fp8attn/cfg1/vs_bf16_engine", rel_l2(outs["fp8_all"].float(), outs["bf16"].float()), 1e-1)
    parity.check("fp8attn/cfg1/vs_fp8_linears_engine", rel_l2(outs["fp8_all"].float(), outs["fp8_linears"].float()), 1e-1)
    parity.check("fp8attn/cfg1/vs_reference_golden", rel_l2(outs["fp8_all"].float(), case.golden["noise_pred"]), 1e-1)


def test_hip_split_head_layernorm_moves_the_forward_closer_to_the_reference(case_cfg1, parity):
    """Round 6: the LayerNorm in front of the output head keeps its bf16 rounding remainder (fw_layernorm_mod_split; the head GEMM runs
    on hi and lo).  The per-site ablation at 40 / 24 / 24 blocks priced that ONE store at a third of the bf16 floor (docs/parity.md);
    here, on the HIP path at config-1 size against the reference golden: the split form must be closer than the round-1..5 form."""
    from fantasy_world_amd.engine import FusionEngine
    from fantasy_world_amd.hip_ops import HipOps
    case = case_cfg1
    ins = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in case.inputs.items()}
    kw = forward_kwargs(case, "cuda")
    eng = FusionEngine(case.cfg, case.weights.__getitem__, HipOps("cuda:0"))
    assert eng.split_head_norm
    errs = {}
    for split in (True, False):
        eng.split_head_norm = split
        out, _ = eng.joint_forward(ins["x"], ins["timestep"], ins["context"], **kw)
        errs[split] = rel_l2(out.float(), case.golden["noise_pred"])
    parity.check("head_split/cfg1/noise_pred_vs_reference_split_head_layernorm", errs[True], E2E_TOL)
    parity.note("head_split/cfg1/noise_pred_vs_reference_round5_form", errs[False])
    print("head LayerNorm split:", errs)
    assert errs[True] < 0.95 * errs[False], errs


def test_hip_fp8_bicross_attention_is_measured_against_the_bf16_bicross(case_cfg1, parity):
    """fp8_attention="bicross" (round 6 experiment, VERDICT r05 missing 2): the two directions of the bicross attention (hd 96) on
    e4m3 operands through the hd-128 kernel (heads zero-padded to 128).  No reference semantics; the record is its distance from the
    SAME engine with the bf16 bicross, under the fp8 path's stated tolerance 2e-2, at config-1 size (the benchmarked depth:
    tests/test_full_depth_gpu.py with FW_FULL_DEPTH_FP8_BICROSS=1)."""
    from fantasy_world_amd.engine import FusionEngine
    from fantasy_world_amd.hip_ops import HipOps
    case = case_cfg1
    ins = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in case.inputs.items()}
    kw = forward_kwargs(case, "cuda")
    outs = {}
    for tag, fa in (("dit", True), ("bicross", "bicross"), ("all", "all")):
        eng = FusionEngine(case.cfg, case.weights.__getitem__, HipOps("cuda:0"), precision="fp8", fp8_attention=fa)
        assert eng.fp8_bicross == (fa != True) and eng.fp8_vggt == (fa == "all")
        outs[tag], _ = eng.joint_forward(ins["x"], ins["timestep"], ins["context"], **kw)
        del eng
    torch.cuda.synchronize()
    assert torch.isfinite(outs["bicross"].float()).all() and torch.isfinite(outs["all"].float()).all()
    parity.check("fp8attn/cfg1/fp8_bicross_vs_bf16_bicross", rel_l2(outs["bicross"].float(), outs["dit"].float()), 2e-2)
    parity.check("fp8attn/cfg1/fp8_bicross_vs_reference_golden", rel_l2(outs["bicross"].float(), case.golden["noise_pred"]), 1e-1)
    # fp8_attention="all" (second half of round 6): additionally the VGGT frame / global attention (hd 64) on fw_attention_fp8's
    # head_dim-64 kernel; same record, same stated tolerance
    parity.check("fp8attn/cfg1/fp8_all_vs_fp8_dit_attention_only", rel_l2(outs["all"].float(), outs["dit"].float()), 2e-2)
    parity.check("fp8attn/cfg1/fp8_all_vs_reference_golden", rel_l2(outs["all"].float(), case.golden["noise_pred"]), 1e-1)


class _ThreadComm:
    """In-process rendezvous of `world` rank threads that share ONE GPU (and its default stream, so enqueue order = execution
    order: what a rank deposited before the barrier is complete before anything a peer enqueues after it)."""

    def __init__(self, world):
        import threading
        self.world, self.slots, self.barrier = world, [None] * world, threading.Barrier(world)

    def exchange(self, rank, value, pick):
        """Every rank deposits `value`; returns pick(all values).  The reads are ENQUEUED between the two barriers: a peer that
        has left the second barrier may free or overwrite what it deposited."""
        self.slots[rank] = value
        self.barrier.wait()
        out = pick(list(self.slots))
        self.barrier.wait()
        return out


def _thread_shard(rank, world, comm):
    """SequenceShard whose three collectives really move every rank's data -- between threads of this process instead of over
    RCCL (which refuses two ranks on one GPU).  Same layouts in and out as parallel.SequenceShard's RCCL versions."""
    from fantasy_world_amd.parallel import Ready, SequenceShard

    class ThreadShard(SequenceShard):
        def all_gather_rows_async(self, t, counts):
            assert t.shape[0] == counts[self.rank]
            return Ready(comm.exchange(self.rank, t.contiguous(), lambda vals: torch.cat(vals, dim=0)))

        def rows_to_heads_async(self, t, parts, counts, cols=None):
            rows, width = t.shape
            c = width // (parts * self.world)
            a, b = cols if cols is not None else (0, c)
            return Ready(comm.exchange(self.rank, t.reshape(rows, parts, self.world, c),
                                       lambda vals: torch.cat([v[:, :, self.rank, a:b] for v in vals], dim=0).contiguous()))

        def heads_to_rows_async(self, o, counts):
            rows, start = counts[self.rank], sum(counts[:self.rank])
            return Ready(comm.exchange(self.rank, o.contiguous(),
                                       lambda vals: torch.cat([v[start:start + rows] for v in vals], dim=1).contiguous()))

    return ThreadShard(rank, world)


@pytest.mark.parametrize("world,split_kv", [(2, False), (4, False), (2, True)])
def test_hip_sequence_sharded_engine_equals_unsharded(case_cfg1, world, split_kv, parity):
    """The sequence-sharded engine ON THE HIP KERNELS at BASELINE config-1 size (L = 9216 rows -> 4608 / 2304 per rank, 9 frames ->
    5+4 / 3+2+2+2, 40 / 16 heads -> 20+20 / 4 x 4 in the head exchange, bicross K/V all-gather): `world` rank threads share the
    one GPU a test box has and exchange through an in-process rendezvous, so every kernel runs at exactly the shapes a rank of a
    real group sees.  Every rank must reproduce the unsharded forward BIT FOR BIT: rows do not depend on the tile or work-group
    that computes them (GEMM, LayerNorm, q/k prep), and attention is per head and per query row.  That holds with the split-KV
    route of tail q-blocks switched off; with it on (the default) the 4-frame rank's frame attention takes that route for its
    5 tail rows per frame where the 9-frame unsharded call does not, the probabilities are rounded to bf16 against a different
    running maximum, and the outputs differ by bf16 roundings (measured 8e-4 rel-L2 end to end, both equally close to the fp32
    reference) -- that case is bounded, and held against the reference golden like the unsharded forward.  What this cannot
    cover is the RCCL transport itself (one-rank group: the test above; world 2-4 over gloo on CPU:
    tests/test_sequence_shard_cpu.py)."""
    import threading
    from fantasy_world_amd.engine import FusionEngine
    from fantasy_world_amd.hip_ops import HipOps
    case = case_cfg1
    ins = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in case.inputs.items()}
    kw = forward_kwargs(case, "cuda")
    def make_ops():
        o = HipOps("cuda:0")
        o.split_kv = split_kv
        return o

    want, _ = FusionEngine(case.cfg, case.weights.__getitem__, make_ops()).joint_forward(
        ins["x"], ins["timestep"], ins["context"], **kw)
    torch.cuda.synchronize()
    comm = _ThreadComm(world)
    outs, errors = [None] * world, []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            eng = FusionEngine(case.cfg, case.weights.__getitem__, make_ops(), shard=_thread_shard(rank, world, comm))
            outs[rank], _ = eng.joint_forward(ins["x"], ins["timestep"], ins["context"], **kw)
        except BaseException as e:                               # a dead rank must not leave its peers in the barrier
            errors.append((rank, e))
            comm.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    torch.cuda.synchronize()
    assert not errors, errors
    for r in range(world):
        assert outs[r] is not None
        if not split_kv:
            assert torch.equal(outs[r], want), (r, rel_l2(outs[r].float(), want.float()))
        else:
            assert torch.equal(outs[r], outs[0])
    if split_kv:
        parity.check(f"shard/world{world}/noise_pred_vs_unsharded", rel_l2(outs[0].float(), want.float()), 2e-3)
        parity.check(f"shard/world{world}/noise_pred_vs_reference", rel_l2(outs[0].float(), case.golden["noise_pred"]), E2E_TOL)


def test_flash_attention_hook_on_hip():
    """Boundary B3 on the GPU: the rebound `flash_attention(q, k, v, num_heads)` hook ([b, s, heads*hd] tensors, bf16 as in
    the inference scripts) against softmax(QK^T/sqrt(hd))V in fp32."""
    import types
    from fantasy_world_amd import install_flash_attention
    mod = types.SimpleNamespace(flash_attention=None)
    undo = install_flash_attention([mod], device="cuda:0")
    g = torch.Generator(device="cuda").manual_seed(0)
    b, heads, hd, lq, lk = 2, 5, 128, 300, 333
    q, k, v = (torch.randn(b, n, heads * hd, device="cuda", generator=g).bfloat16() for n in (lq, lk, lk))
    out = mod.flash_attention(q, k, v, heads)
    assert out.shape == q.shape and out.dtype == q.dtype
    qh, kh, vh = (t.float().view(b, -1, heads, hd).transpose(1, 2) for t in (q, k, v))
    want = torch.softmax(qh @ kh.transpose(-1, -2) / hd ** 0.5, dim=-1) @ vh
    assert rel_l2(out.float(), want.transpose(1, 2).reshape(b, lq, heads * hd)) < 4e-3
    undo()
    assert mod.flash_attention is None


# ---------------------------------------------------------------------------------------------------------------
# finer-grained B3 hooks (fantasy_world_amd/hooks.py) on the GPU.  The reference tree does not exist on the GPU box, so the
# modules below are stand-ins with the reference's attribute names and arithmetic (the hooks are duck-typed: they look for a
# class called BiMultiHeadAttention and call its projections); the REAL reference modules run through the same hooks on the CPU
# op set in tests/test_install_dropin.py.
# ---------------------------------------------------------------------------------------------------------------
class BiMultiHeadAttention(torch.nn.Module):
    """fusion/layer/block.py:315-625 reduced to what forward_sdpa touches (no RoPE: freqs_dit=None)."""

    def __init__(self, m1_dim, m2_dim, embed_dim, num_heads):
        super().__init__()
        self.embed_dim, self.num_heads, self.head_dim = embed_dim, num_heads, embed_dim // num_heads
        self.m1_proj, self.m2_proj = torch.nn.Linear(m1_dim, embed_dim), torch.nn.Linear(m2_dim, embed_dim)
        self.values_m1_proj, self.values_m2_proj = torch.nn.Linear(m1_dim, embed_dim), torch.nn.Linear(m2_dim, embed_dim)
        self.out_m1_proj, self.out_m2_proj = torch.nn.Linear(embed_dim, m1_dim), torch.nn.Linear(embed_dim, m2_dim)
        self.attn_implementation = "sdpa"

    def forward(self, x1, x2, **kw):
        return self.forward_sdpa(x1, x2, **kw)

    def forward_sdpa(self, x1, x2, attention_mask_1=None, attention_mask_2=None, freqs=None, freqs_dit=None, freqs_agg=None):
        b, L1, _ = x1.shape
        L2 = x2.shape[1]
        sh = lambda t, n: t.view(b, n, self.num_heads, self.head_dim).transpose(1, 2)
        q, k = sh(self.m1_proj(x1), L1), sh(self.m2_proj(x2), L2)
        v1, v2 = sh(self.values_m1_proj(x1), L1), sh(self.values_m2_proj(x2), L2)
        o1 = torch.nn.functional.scaled_dot_product_attention(q, k, v2).transpose(1, 2).reshape(b, L1, self.embed_dim)
        o2 = torch.nn.functional.scaled_dot_product_attention(k, q, v1).transpose(1, 2).reshape(b, L2, self.embed_dim)
        return self.out_m1_proj(o1), self.out_m2_proj(o2)


def test_b3_hooks_on_hip(parity):
    from fantasy_world_amd import install_bicross_attention, HipLayerNorm, HipLinear
    from fantasy_world_amd.hip_ops import HipOps
    ops = HipOps("cuda:0")
    torch.manual_seed(0)
    # bicross attention: 12 heads x 96 like the real block, both directions
    m = BiMultiHeadAttention(256, 128, 1152, 12).cuda().eval()
    x1, x2 = torch.randn(1, 700, 256, device="cuda"), torch.randn(1, 333, 128, device="cuda")
    with torch.no_grad():
        want = m(x1, x2)
        undo = install_bicross_attention(m, ops=ops)
        got = m(x1, x2)
        undo()
        again = m(x1, x2)
    for i, (g_, w_) in enumerate(zip(got, want)):
        parity.check(f"hook/bicross_attention/out{i + 1}", rel_l2(g_, w_), 6e-3)      # bf16 q/k/v/P against fp32 SDPA
    assert all(torch.equal(a, b) for a, b in zip(again, want))
    # LayerNorm kernel hook
    ln = HipLayerNorm(5120, eps=1e-6, elementwise_affine=False, ops=ops)
    x = torch.randn(3, 50, 5120, device="cuda")
    parity.check("hook/layernorm_noaffine", rel_l2(ln(x), torch.nn.functional.layer_norm(x, (5120,), None, None, 1e-6)), 4e-3)
    lna = HipLayerNorm(1024, eps=1e-5, elementwise_affine=True, ops=ops).cuda()
    with torch.no_grad():
        lna.weight.normal_()
        lna.bias.normal_()
    xa = torch.randn(70, 1024, device="cuda")
    parity.check("hook/layernorm_affine", rel_l2(lna(xa), torch.nn.functional.layer_norm(xa, (1024,), lna.weight, lna.bias, 1e-5)), 4e-3)
    # nn.Linear module-map target, bf16 and fp8 computation dtypes
    lin = torch.nn.Linear(200, 320).cuda().bfloat16()              # K = 200: padded to 256 inside
    xl = torch.randn(4, 33, 200, device="cuda").bfloat16()
    hl = HipLinear(lin, computation_dtype=torch.bfloat16, computation_device="cuda", ops=ops)
    with torch.no_grad():
        want = torch.nn.functional.linear(xl.float(), lin.weight.float(), lin.bias.float())
        got = hl(xl)
    assert got.dtype == torch.bfloat16 and got.shape == want.shape
    parity.check("hook/hip_linear_bf16", rel_l2(got.float(), want), 4e-3)
    h8 = HipLinear(lin, computation_dtype=torch.float8_e4m3fn, computation_device="cuda", ops=ops)
    from oracle.ref_ops import TorchRefOps
    ref = TorchRefOps()
    w8 = ref.linear_fp8(xl.float().cpu().reshape(-1, 200), ref.pack_linear_fp8(lin.weight.float().cpu(), lin.bias.float().cpu()),
                        out_f32=True)
    with torch.no_grad():
        g8 = h8(xl)
    parity.check("hook/hip_linear_fp8", rel_l2(g8.float().reshape(-1, 320), w8), 4e-3)


def test_hip_graphed_denoise_step_equals_eager(case_l2):
    """one whole sampling step captured in a HIP graph (sampler.GraphedDenoiseStep: two forwards +
    fw_cfg_euler_step, per-step values fed through device buffers) replays to the SAME bits as the eager step, for
    consecutive steps with different timesteps, and in the merged-CFG form."""
    from fantasy_world_amd.engine import FusionEngine
    from fantasy_world_amd.hip_ops import HipOps
    from fantasy_world_amd.sampler import FlowMatchScheduler, GraphedDenoiseStep, denoise_step
    case = case_l2
    eng = FusionEngine(case.cfg, case.weights.__getitem__, HipOps("cuda:0"))
    d = {k: (v.cuda().to(torch.bfloat16) if torch.is_tensor(v) and v.is_floating_point() else (v.cuda() if torch.is_tensor(v) else v))
         for k, v in case.inputs.items()}
    cond = dict(clip_feature=d["clip_feature"], y=d["y"], plucker_fea=d["plucker_fea"], plucker_context_lens=d["plucker_context_lens"])
    sched = FlowMatchScheduler()
    sched.set_timesteps(50)
    for merge in (False, True):
        lat_e = lat_g = d["x"]
        g = GraphedDenoiseStep(eng, sched, d["x"], d["context"], d["context_neg"], cond, merge_cfg=merge)
        for step in (3, 4, 49):
            lat_e, _ = denoise_step(eng, sched, step, lat_e, d["context"], d["context_neg"], cond, merge_cfg=merge)
            lat_g = g.step(step, lat_g).clone()
            torch.cuda.synchronize()
            assert lat_g.dtype == torch.bfloat16 and torch.equal(lat_g, lat_e), (merge, step)
        del g


def _thread_tensor_shard(rank, world, comm, reduce_dtype):
    """TensorShard whose all-reduce / row all-gather really combine every rank thread's data (in-process, see _ThreadComm)."""
    from fantasy_world_amd.parallel import Ready
    from fantasy_world_amd.tensor_parallel import TensorShard

    class ThreadTensorShard(TensorShard):
        def all_reduce_async(self, t, kind="all_reduce", op=None):
            # sum in fp32 in RANK order on every rank (deterministic, identical everywhere), rounded once to the message dtype
            # (op given: MAX of the fp8 row maxima)
            comb = (lambda vals: torch.stack(vals).amax(0)) if op is not None else (
                lambda vals: torch.stack([v.float() for v in vals]).sum(0).to(t.dtype))
            tot = comm.exchange(self.rank, t, comb)
            t.copy_(tot)
            return Ready(t)

        def all_gather_rows_async(self, t, counts):
            return Ready(comm.exchange(self.rank, t.contiguous(), lambda vals: torch.cat(vals, dim=0)))

    return ThreadTensorShard(rank, world, reduce_dtype=reduce_dtype, chunk_rows=2048)


@pytest.mark.parametrize("world,reduce_dtype", [(2, torch.bfloat16), (4, torch.bfloat16), (4, torch.float32)])
def test_hip_tensor_parallel_engine_matches_unsharded(case_cfg1, world, reduce_dtype, parity):
    """north_star's head / FFN-column partition ON THE HIP KERNELS at BASELINE config-1 size: `world` rank threads share the one
    GPU of a test box and all-reduce through an in-process rendezvous, so every kernel runs at a real TP rank's shapes (20 / 10
    DiT heads, 8 / 4 VGGT heads, 6 / 3 bicross heads with the 64-padded out-projection slab, 6912 / 3456 FFN columns, fw_row_sumsq
    -> all-reduce -> fw_qk_prep_tp, row-blocked reductions, fw_residual_add epilogues).  TP is not bit-identical to the unsharded
    forward: the partial sums add up in another order (and are rounded to the message dtype first in the bf16 mode), and a 1e-7
    perturbation of the fp32 stream decorrelates the bf16 rounding realisation of everything downstream within a few GEMMs
    (measured on CPU with bf16 emulation too: 2.7e-3 between the two, both 2.6e-3 from the fp32 truth).  So the two forwards are
    two independent bf16 realisations: <= sqrt(2) x 2.7e-3 apart (+ the partial-sum rounding), and -- the claim that matters -- TP
    is as close to the fp32 reference golden as the unsharded engine is."""
    import threading
    from fantasy_world_amd.engine import FusionEngine
    from fantasy_world_amd.hip_ops import HipOps
    from fantasy_world_amd.tensor_parallel import TPFusionEngine
    case = case_cfg1
    ins = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in case.inputs.items()}
    kw = forward_kwargs(case, "cuda")
    want, _ = FusionEngine(case.cfg, case.weights.__getitem__, HipOps("cuda:0")).joint_forward(
        ins["x"], ins["timestep"], ins["context"], **kw)
    torch.cuda.synchronize()
    comm = _ThreadComm(world)
    outs, errors = [None] * world, []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            eng = TPFusionEngine(case.cfg, case.weights.__getitem__, HipOps("cuda:0"),
                                 _thread_tensor_shard(rank, world, comm, reduce_dtype))
            outs[rank], _ = eng.joint_forward(ins["x"], ins["timestep"], ins["context"], **kw)
        except BaseException as e:
            errors.append((rank, e))
            comm.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=900)
    torch.cuda.synchronize()
    assert not errors, errors
    for r in range(1, world):
        assert torch.equal(outs[r], outs[0]), r                     # replicated streams: every rank computes the same bits
    tag = f"tp/world{world}/{'bf16' if reduce_dtype == torch.bfloat16 else 'fp32'}_reduce"
    parity.check(f"{tag}/noise_pred_vs_unsharded", rel_l2(outs[0].float(), want.float()), 6e-3)
    parity.check(f"{tag}/noise_pred_vs_reference", rel_l2(outs[0].float(), case.golden["noise_pred"]), E2E_TOL)


def _run_rank_threads(world, make_engine, ins, kw):
    import threading
    comm = _ThreadComm(world)
    outs, errors = [None] * world, []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            outs[rank], _ = make_engine(rank, comm).joint_forward(ins["x"], ins["timestep"], ins["context"], **kw)
        except BaseException as e:                               # a dead rank must not leave its peers in the barrier
            errors.append((rank, e))
            comm.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=900)
    torch.cuda.synchronize()
    assert not errors, errors
    return outs


@pytest.mark.parametrize("world", [2, 4])
def test_hip_fp8_sequence_sharded_engine_equals_unsharded(case_cfg1, world, parity):
    """BASELINE config 5's arithmetic under the sequence shard ON THE HIP KERNELS: fp8 linears on L/n local rows,
    q | k written as e4m3 by fw_qk_prep_fp8, v cast raw, the head exchange carrying BYTES, fw_v_transpose_e4m3 + fw_attention_fp8 on
    the received heads (20 / 10 per rank, at world 4 in the groups (4, 6)).  Every rank must return the unsharded fp8 engine's BITS:
    the quantiser, the fp8 GEMM and the q/k pass are per row, the attention per head and query row, and the bytes a rank receives
    are the bytes the unsharded engine hands its kernel.  (Split-KV tails of the bf16 VGGT frame attention switched off, as in the
    bf16 twin of this test: that route depends on the frame count of the launch.)"""
    from fantasy_world_amd.engine import FusionEngine
    from fantasy_world_amd.hip_ops import HipOps
    case = case_cfg1
    ins = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in case.inputs.items()}
    kw = forward_kwargs(case, "cuda")
    opts = dict(precision="fp8", fp8_attention=True)

    def make_ops():
        o = HipOps("cuda:0")
        o.split_kv = False
        return o
    want, _ = FusionEngine(case.cfg, case.weights.__getitem__, make_ops(), **opts).joint_forward(ins["x"], ins["timestep"], ins["context"], **kw)
    torch.cuda.synchronize()
    outs = _run_rank_threads(world, lambda r, comm: FusionEngine(case.cfg, case.weights.__getitem__, make_ops(),
                                                                 shard=_thread_shard(r, world, comm), **opts), ins, kw)
    for r in range(world):
        assert outs[r] is not None and torch.equal(outs[r], want), (r, rel_l2(outs[r].float(), want.float()))
    parity.check(f"shard_fp8/world{world}/noise_pred_vs_reference", rel_l2(outs[0].float(), case.golden["noise_pred"]), 1e-1)


@pytest.mark.parametrize("world", [2, 4])
def test_hip_fp8_tensor_parallel_engine_matches_unsharded(case_cfg1, world, parity):
    """The same arithmetic under north_star's partition on the HIP kernels: column-parallel fw_gemm_fp8 on the
    replicated activation, row-parallel fw_gemm_fp8 on K-slices quantised with the FULL row's scale (fw_row_absmax -> MAX all-reduce
    -> fw_fp8_quant_rows_amax), fw_qk_prep_fp8 with external statistics, fw_attention_fp8 on the rank's heads.  Not bit-identical
    (partial sums in another order; e4m3 rounding amplifies last-bit differences): another realisation of the same arithmetic, within
    the fp8 path's stated 2e-2 of the unsharded fp8 engine and as far from the fp32 reference golden as that engine is."""
    from fantasy_world_amd.engine import FusionEngine
    from fantasy_world_amd.hip_ops import HipOps
    from fantasy_world_amd.tensor_parallel import TPFusionEngine
    case = case_cfg1
    ins = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in case.inputs.items()}
    kw = forward_kwargs(case, "cuda")
    opts = dict(precision="fp8", fp8_attention=True)
    want, _ = FusionEngine(case.cfg, case.weights.__getitem__, HipOps("cuda:0"), **opts).joint_forward(ins["x"], ins["timestep"], ins["context"], **kw)
    torch.cuda.synchronize()
    outs = _run_rank_threads(world, lambda r, comm: TPFusionEngine(case.cfg, case.weights.__getitem__, HipOps("cuda:0"),
                                                                   _thread_tensor_shard(r, world, comm, torch.float32), **opts), ins, kw)
    for r in range(1, world):
        assert torch.equal(outs[r], outs[0]), r
    gold = case.golden["noise_pred"]
    parity.check(f"tp_fp8/world{world}/noise_pred_vs_unsharded_fp8_engine", rel_l2(outs[0].float(), want.float()), 2e-2)
    e_tp, e_one = rel_l2(outs[0].float(), gold), rel_l2(want.float(), gold)
    parity.check(f"tp_fp8/world{world}/noise_pred_vs_reference", e_tp, 1e-1)
    assert e_tp < 1.25 * e_one, (e_tp, e_one)


def _rccl_two_rank_worker(rank, world, port, mode, outdir):
    import os
    import sys
    sys.path.insert(0, ROOT_DIR)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY="0", FW_PARALLEL=mode)
    import torch
    from conftest import Case, forward_kwargs as fkw
    from fantasy_world_amd import parallel
    from fantasy_world_amd.hip_ops import HipOps
    torch.cuda.set_device(rank)
    topo = parallel.init_topology(backend="nccl", cfg_parallel=False)
    case = Case("wan21_l3_f2_12x8")
    dev = f"cuda:{rank}"
    eng = parallel.make_engine(case.cfg, case.weights.__getitem__, HipOps(dev), topo)
    ins = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in case.inputs.items()}
    out, _ = eng.joint_forward(ins["x"], ins["timestep"], ins["context"], **fkw(case, dev))
    torch.cuda.synchronize()
    torch.save(out.cpu(), os.path.join(outdir, f"rccl_{mode}_{rank}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


ROOT_DIR = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))


@pytest.mark.parametrize("mode", ["sp", "tp"])
def test_two_rank_rccl_on_two_gpus(mode, case_l3, tmp_path, parity):
    """The first box with more than one GPU exercises the REAL transport: two processes, one per GPU, RCCL over xGMI, the
    sequence shard (head all-to-all + all-gathers) and the tensor-parallel partition (all-reduces), against the golden.
    Skipped on the 1-GPU boxes this build has had (covered there by the one-rank RCCL group and the rank-thread tests)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL refuses two ranks on one device)")
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_rccl_two_rank_worker, args=(2, port, mode, str(tmp_path)), nprocs=2, join=True)
    a = torch.load(str(tmp_path / f"rccl_{mode}_0.pt"))
    b = torch.load(str(tmp_path / f"rccl_{mode}_1.pt"))
    assert torch.equal(a, b)
    parity.check(f"rccl2/{mode}/noise_pred_vs_reference", rel_l2(a.float(), case_l3.golden["noise_pred"]), E2E_TOL)
