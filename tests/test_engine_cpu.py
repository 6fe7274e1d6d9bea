"""Host logic of the engine (op order, weight packing/padding, modulation algebra, rotary tables, layout glue) checked
on CPU: the engine is driven with the plain-torch op set of oracle/ref_ops.py (test infrastructure) and compared with
the oracle and with the reference goldens.  The HIP op set is covered by the -m gpu tests."""
import pytest
import torch

from conftest import rel_l2, forward_kwargs
from fantasy_world_amd.engine import FusionEngine
from oracle.ref_ops import TorchRefOps


def _run(case, emulate_bf16):
    eng = FusionEngine(case.cfg, case.weights.__getitem__, TorchRefOps(emulate_bf16=emulate_bf16))
    ins = case.inputs
    col = {}
    out, pred = eng.joint_forward(ins["x"], ins["timestep"], ins["context"], collect=col, **forward_kwargs(case))
    assert pred is None
    col["noise_pred"] = out
    return col


@pytest.mark.parametrize("case_name", ["case_l2", "case_l3", "case_w22", "case_camtok"])
def test_engine_fp32_matches_golden(case_name, request):
    case = request.getfixturevalue(case_name)
    col = _run(case, emulate_bf16=False)
    for k in ("x_after_pcb", "x_final", "tokens_final", "noise_pred"):
        err = rel_l2(col[k].reshape(case.golden[k].shape), case.golden[k])
        assert err < 2e-5, f"{case.name}:{k} rel-L2 {err:.3e}"


def test_engine_bf16_emulation_yardstick(case_l2):
    """With activations rounded to bf16 where the HIP path stores bf16, the end-to-end error is the yardstick the GPU
    parity test is held to (the reference's own bf16-vs-fp32 gap is 3e-3 per block)."""
    col = _run(case_l2, emulate_bf16=True)
    err = rel_l2(col["noise_pred"], case_l2.golden["noise_pred"])
    assert err < 1e-2, err


def test_engine_depth_yardstick(case_depth, parity):
    """The 8-block depth golden on the engine's host logic: fp32 op set = the reference after every block; with activations
    rounded to bf16 where the HIP path stores bf16 = the yardstick the GPU's per-block errors are read against."""
    case, g = case_depth, case_depth.golden
    for emulate in (False, True):
        eng = FusionEngine(case.cfg, case.weights.__getitem__, TorchRefOps(emulate_bf16=emulate))
        got = {"x": {}, "tok": {}}
        col = {"per_block": lambda kind, i, t: got[kind].__setitem__(i, t[g["rows_dit"] if kind == "x" else g["rows_agg"]].clone())}
        out, _ = eng.joint_forward(case.inputs["x"], case.inputs["timestep"], case.inputs["context"], collect=col,
                                   **forward_kwargs(case))
        ex = [rel_l2(got["x"][b], g["x_blocks"][b]) for b in range(case.cfg.num_layers)]
        et = [rel_l2(got["tok"][j], g["tok_blocks"][j]) for j in range(case.cfg.n_irg)]
        if emulate:
            parity.note("depth/yardstick_bf16_emulation_x_per_block", ex)
            parity.note("depth/yardstick_bf16_emulation_vggt_per_block", et)
            parity.note("depth/yardstick_bf16_emulation_noise_pred", rel_l2(out, g["noise_pred"]))
            assert max(ex) < 1.6e-2 and max(et) < 1.6e-2, (ex, et)
        else:
            assert max(ex) < 2e-5 and max(et) < 2e-5 and rel_l2(out, g["noise_pred"]) < 2e-5, (ex, et)


def test_engine_fp8_mode_matches_fp8_oracle(case_l2):
    """precision="fp8" on the CPU op set (fp8_linear by its definition) against the oracle with the reference's fp8_linear at the
    same module sites: host logic of the mode (which linears are packed fp8, fused q|k|v quantised row-wise ONCE for all three,
    bias rounding)."""
    from oracle import fw_oracle
    case, ins = case_l2, case_l2.inputs
    want = fw_oracle.joint_forward(case.weights, case.cfg, ins["x"], ins["timestep"], ins["context"], ins["clip_feature"],
                                   ins["y"], ins["plucker_fea"], ins["plucker_context_lens"], fp8_linears=True)
    eng = FusionEngine(case.cfg, case.weights.__getitem__, TorchRefOps(), precision="fp8")
    got, _ = eng.joint_forward(ins["x"], ins["timestep"], ins["context"], **forward_kwargs(case))
    err = rel_l2(got, want)
    full = rel_l2(want, case.golden["noise_pred"])
    print(f"engine(fp8, torch ops) vs fp8 oracle {err:.3e}; fp8 oracle vs fp32 reference {full:.3e}")
    # the oracle rounds the fp8 linear's input and output to bf16 (the reference's dtypes there), the fp32 op set does not
    assert err < 1.5e-2 and 1e-3 < full < 0.2, (err, full)


def test_all_zero_plucker_skips_adapter(case_l2):
    """camera_control.py:111,124-127: an all-zero plucker feature leaves the attention output untouched."""
    case = case_l2
    eng = FusionEngine(case.cfg, case.weights.__getitem__, TorchRefOps())
    ins = case.inputs
    z = torch.zeros_like(ins["plucker_fea"])
    a, _ = eng.joint_forward(ins["x"], ins["timestep"], ins["context"], clip_feature=ins["clip_feature"], y=ins["y"],
                             plucker_fea=z, plucker_context_lens=ins["plucker_context_lens"])
    b, _ = eng.joint_forward(ins["x"], ins["timestep"], ins["context"], clip_feature=ins["clip_feature"], y=ins["y"],
                             plucker_fea=None, plucker_context_lens=None)
    assert torch.equal(a, b)


def test_wan22_control_features_are_cached_per_tensor(case_w22):
    """The control adapter output depends only on control_camera_latents_input: with the step-invariant cache it is computed
    once, reused for the negative pass and the following steps, recomputed when the tensor changes (in place or a new one);
    with the cache off it is recomputed in every call, like the reference (wan_video_dit.py:390-396)."""
    case = case_w22
    ins = case.inputs
    kw = forward_kwargs(case)

    class CountingOps(TorchRefOps):
        n_ctl = 0

        def control_patchify(self, *a, **k):
            self.n_ctl += 1
            return super().control_patchify(*a, **k)

    ops = CountingOps()
    eng = FusionEngine(case.cfg, case.weights.__getitem__, ops, cache_step_invariants=True)
    a, _ = eng.joint_forward(ins["x"], ins["timestep"], ins["context"], **kw)
    b, _ = eng.joint_forward(ins["x"], ins["timestep"], ins["context_neg"], **kw)
    assert ops.n_ctl == 1
    kw2 = dict(kw, control_camera_latents_input=kw["control_camera_latents_input"] * 2.0)
    c, _ = eng.joint_forward(ins["x"], ins["timestep"], ins["context"], **kw2)
    assert ops.n_ctl == 2 and not torch.equal(a, c)
    kw["control_camera_latents_input"].mul_(2.0)                      # in-place edit bumps the version counter
    d, _ = eng.joint_forward(ins["x"], ins["timestep"], ins["context"], **kw)
    assert ops.n_ctl == 3 and torch.equal(c, d)
    kw["control_camera_latents_input"].mul_(0.5)
    plain_ops = CountingOps()
    plain = FusionEngine(case.cfg, case.weights.__getitem__, plain_ops)
    e, _ = plain.joint_forward(ins["x"], ins["timestep"], ins["context"], **kw)
    f, _ = plain.joint_forward(ins["x"], ins["timestep"], ins["context"], **kw)
    assert plain_ops.n_ctl == 2 and torch.equal(e, a) and torch.equal(f, a)


def test_caches_survive_address_recycling_between_generations(case_w22, case_l2):
    """a cache keyed on (address, shape, version) alone serves stale data when the allocator hands the next
    generation's conditioning tensor the address of the previous one.  The caches hold the source tensor itself: a second
    generation whose control / Pluecker tensor is a DIFFERENT object (even one that would compare equal by address after the
    first was freed) is recomputed, and the all-zero verdict of one Pluecker tensor is never applied to another."""
    import gc
    case = case_w22
    ins = case.inputs
    eng = FusionEngine(case.cfg, case.weights.__getitem__, TorchRefOps(), cache_step_invariants=True)
    plain = FusionEngine(case.cfg, case.weights.__getitem__, TorchRefOps())
    kw = forward_kwargs(case)
    ctl1 = kw["control_camera_latents_input"].clone()
    a, _ = eng.joint_forward(ins["x"], ins["timestep"], ins["context"], **dict(kw, control_camera_latents_input=ctl1))
    ptr, shape = ctl1.data_ptr(), ctl1.shape
    del ctl1
    gc.collect()
    # the cache still references the first tensor, so the allocator cannot give its storage to the next one
    ctl2 = torch.empty(shape).copy_(kw["control_camera_latents_input"] * -1.5)
    assert ctl2.data_ptr() != ptr
    b, _ = eng.joint_forward(ins["x"], ins["timestep"], ins["context"], **dict(kw, control_camera_latents_input=ctl2))
    want, _ = plain.joint_forward(ins["x"], ins["timestep"], ins["context"], **dict(kw, control_camera_latents_input=ctl2))
    assert torch.equal(b, want) and not torch.equal(a, b)
    # Pluecker all-zero verdict (Wan2.1): zero tensor first, then a non-zero one of the same shape / version
    c = case_l2
    e2 = FusionEngine(c.cfg, c.weights.__getitem__, TorchRefOps(), cache_step_invariants=True)
    p2 = FusionEngine(c.cfg, c.weights.__getitem__, TorchRefOps())
    k2 = forward_kwargs(c)
    z = torch.zeros_like(k2["plucker_fea"])
    e2.joint_forward(c.inputs["x"], c.inputs["timestep"], c.inputs["context"], **dict(k2, plucker_fea=z))
    del z
    gc.collect()
    nz = k2["plucker_fea"].clone()
    got, _ = e2.joint_forward(c.inputs["x"], c.inputs["timestep"], c.inputs["context"], **dict(k2, plucker_fea=nz))
    want, _ = p2.joint_forward(c.inputs["x"], c.inputs["timestep"], c.inputs["context"], **dict(k2, plucker_fea=nz))
    assert torch.equal(got, want)


def test_engine_return_prediction_matches_reference_golden(case_pred):
    """joint_forward(return_prediction=True) with heads_cfg returns the reference's prediction dict (M21:217-224)."""
    from conftest import PRED_KEYS, forward_kwargs
    c, ins = case_pred, case_pred.inputs
    eng = FusionEngine(c.cfg, c.weights.__getitem__, TorchRefOps(), heads_cfg=c.hc)
    out, pred = eng.joint_forward(ins["x"], ins["timestep"], ins["context"], return_prediction=True, **forward_kwargs(c))
    assert rel_l2(out, c.golden["noise_pred"]) < 2e-5
    assert set(pred.keys()) == set(PRED_KEYS)
    for k in PRED_KEYS:
        assert pred[k].shape == c.golden[k].shape
        assert rel_l2(pred[k], c.golden[k]) < 5e-5, k
    # without heads_cfg the aggregated tokens come back instead (the layers the heads would read)
    eng2 = FusionEngine(c.cfg, c.weights.__getitem__, TorchRefOps())
    _, outputs = eng2.joint_forward(ins["x"], ins["timestep"], ins["context"], return_prediction=True, **forward_kwargs(c))
    assert all(v.shape[-1] == 2 * c.cfg.vggt_dim for v in outputs.values()) and (c.cfg.n_irg - 1) in outputs


def test_step_invariant_cache_is_exact_and_skips_work(case_l2):
    """with cache_step_invariants the second and later calls reuse the context embeddings, every
    block's cross-attention K/V and the adapter's Pluecker term -- bit-identical outputs, fewer GEMMs; a new or modified
    context tensor is recomputed."""
    from conftest import forward_kwargs
    c, ins = case_l2, case_l2.inputs
    kw = forward_kwargs(c)

    class CountingOps(TorchRefOps):
        def __init__(self):
            super().__init__()
            self.n_linear = 0

        def linear(self, *a, **k):
            self.n_linear += 1
            return super().linear(*a, **k)

    plain_ops, cached_ops = CountingOps(), CountingOps()
    plain = FusionEngine(c.cfg, c.weights.__getitem__, plain_ops)
    cached = FusionEngine(c.cfg, c.weights.__getitem__, cached_ops, cache_step_invariants=True)
    t2 = ins["timestep"] * 0.5
    want1, _ = plain.joint_forward(ins["x"], ins["timestep"], ins["context"], **kw)
    n_plain = plain_ops.n_linear
    want2, _ = plain.joint_forward(ins["x"], t2, ins["context"], **kw)
    got1, _ = cached.joint_forward(ins["x"], ins["timestep"], ins["context"], **kw)
    n_first = cached_ops.n_linear
    got2, _ = cached.joint_forward(ins["x"], t2, ins["context"], **kw)
    n_second = cached_ops.n_linear - n_first
    assert torch.equal(got1, want1) and torch.equal(got2, want2)
    assert n_first == n_plain
    # per DiT block: text K/V, image K/V and the adapter's Pluecker GEMM drop out; plus 2 text-embedding and 2 image-embedding GEMMs
    assert n_second == n_plain - (3 * c.cfg.num_layers + 4), (n_plain, n_second)
    # a different context tensor (the negative prompt) gets its own entries, and coming back to the first one hits
    neg = ins["context_neg"]
    want3, _ = plain.joint_forward(ins["x"], t2, neg, **kw)
    got3, _ = cached.joint_forward(ins["x"], t2, neg, **kw)
    assert torch.equal(got3, want3)
    before = cached_ops.n_linear
    got4, _ = cached.joint_forward(ins["x"], t2, ins["context"], **kw)
    assert torch.equal(got4, want2) and cached_ops.n_linear - before == n_second
    # an in-place edit of the context bumps its version counter: recomputed, not served stale
    ctx = ins["context"].clone()
    a, _ = cached.joint_forward(ins["x"], t2, ctx, **kw)
    ctx.mul_(0.5)
    b, _ = cached.joint_forward(ins["x"], t2, ctx, **kw)
    ref_b, _ = plain.joint_forward(ins["x"], t2, ctx, **kw)
    assert torch.equal(a, want2) and torch.equal(b, ref_b) and not torch.equal(a, b)


@pytest.mark.parametrize("case_name", ["case_l2", "case_w22"])
def test_merged_cfg_pair_equals_two_forwards(case_name, request):
    """SURVEY.md 8(f) item 2, CFG batch-2 merge: joint_forward_pair runs the positive and the negative pass as one forward over 2L
    rows (attention batch 2); every op is row-wise or per (batch, head), so the two results equal the two separate forwards
    exactly -- with and without the step-invariant cache, and through sampler.denoise_step(merge_cfg=True)."""
    from fantasy_world_amd.sampler import FlowMatchScheduler, denoise_step
    case = request.getfixturevalue(case_name)
    ins = case.inputs
    kw = forward_kwargs(case)
    for cached in (False, True):
        eng = FusionEngine(case.cfg, case.weights.__getitem__, TorchRefOps(), cache_step_invariants=cached)
        for rep in range(2):                        # second round: served from the cache when it is on
            pos, _ = eng.joint_forward(ins["x"], ins["timestep"], ins["context"], **kw)
            neg, _ = eng.joint_forward(ins["x"], ins["timestep"], ins["context_neg"], **kw)
            mp, mn, pred = eng.joint_forward_pair(ins["x"], ins["timestep"], ins["context"], ins["context_neg"], **kw)
            assert pred is None and rel_l2(mp, pos) < 1e-6 and rel_l2(mn, neg) < 1e-6, (cached, rep)
            assert rel_l2(pos, neg) > 1e-3          # the two prompts really differ
    sched = FlowMatchScheduler()
    sched.set_timesteps(50)
    cond = {k: v for k, v in kw.items() if k != "uncond"}
    a, _ = denoise_step(eng, sched, 3, ins["x"], ins["context"], ins["context_neg"], cond)
    b, _ = denoise_step(eng, sched, 3, ins["x"], ins["context"], ins["context_neg"], cond, merge_cfg=True)
    assert rel_l2(b, a) < 1e-6


def test_fp8_attention_statement_linear_byte_against_the_exact_exponential():
    """Round 6: the kernel's probabilities are the e4m3 BYTE round(8 (s - m + 7) + 56) (2^f taken as 1 + f inside a binade; one integer
    conversion per score, csrc/attention_fp8.hip) -- the CPU statement follows (TorchRefOps.fp8_linear_exp, the default).  Here, on the
    statement alone: every byte stays inside e4m3's finite positive range, the largest probability of a row is 2^7, bytes are monotone
    in the score, the byte decodes to within (1 + f) / 2^f <= 6.2 % above -- never below by more than half a byte -- the exact
    exponential, and the attention output stays within e4m3 noise of the exact-exponential statement."""
    import torch
    from oracle.ref_ops import TorchRefOps
    x = torch.linspace(-20.0, 0.0, 4001, dtype=torch.float64)               # s - m
    bits = torch.round(8.0 * x.float() + 112.0).clamp(0, 255).to(torch.uint8)
    assert int(bits.max()) == 0x70 and int(bits.min()) == 0 and bool((bits[1:] >= bits[:-1]).all())
    val = bits.view(torch.float8_e4m3fn).to(torch.float64)
    exact = torch.exp2(x + 7.0)
    normal = bits >= 8
    ratio = (val / exact)[normal]
    assert float(ratio.max()) < 1.0615 * 1.045 and float(ratio.min()) > 0.955, (float(ratio.min()), float(ratio.max()))   # 1 + f vs 2^f, +- half a byte
    assert float(val[128]) == 0.0                                            # 14 binades below the maximum: flushed like e4m3(2^x) would be

    g = torch.Generator().manual_seed(5)
    H, hd, Lq, Lk = 2, 128, 96, 700
    ops = TorchRefOps(exact=True)
    q = torch.randn(Lq, H * hd, generator=g) * ops.q_scale_fp8(hd)
    k, v = torch.randn(Lk, H * hd, generator=g), torch.randn(Lk, H * hd, generator=g)
    q8, k8 = ops.cast_fp8(q), ops.cast_fp8(k)
    v8, _ = ops.prepare_v_fp8(v, H, hd)
    outs = {}
    for lin in (True, False):
        ops.fp8_linear_exp = lin
        outs[lin] = ops.attention_fp8(q8, k8, v8, H, hd, Lk).float()
    e = rel_l2(outs[True], outs[False])
    assert 0 < e < 6e-2, e                 # two realisations of P's 3-bit rounding (the GPU arms measure 2.1-4.1e-2, same bound)


def test_fp8_attention_engine_on_its_cpu_statement(case_l2):
    """fp8 attention has no reference semantics (the reference defines fp8 for nn.Linear only).  Its CPU statement in the test op set
    (oracle/ref_ops.py: the semantics include/fw_mi355x.h states) exists so that the HOST orchestration -- what is cast where, what the
    exchanges carry -- runs without a GPU (tests/test_sequence_shard_cpu.py, tests/test_tensor_parallel_cpu.py): here the engine with it
    stays within e4m3 noise of the same engine with exact attention, and an op set WITHOUT fw_attention_fp8 is refused at construction,
    not somewhere inside the first forward."""
    import pytest
    from fantasy_world_amd.engine import FusionEngine
    from oracle.ref_ops import TorchRefOps
    kw = forward_kwargs(case_l2)
    ins = case_l2.inputs
    outs = {}
    for tag, opts in (("fp8_linears", dict(precision="fp8")), ("fp8_all", dict(precision="fp8", fp8_attention=True)),
                      ("fp8_every_attention", dict(precision="fp8", fp8_attention="all"))):
        eng = FusionEngine(case_l2.cfg, case_l2.weights.__getitem__, TorchRefOps(), **opts)
        outs[tag], _ = eng.joint_forward(ins["x"], ins["timestep"], ins["context"], **kw)
    e = rel_l2(outs["fp8_all"], outs["fp8_linears"])
    assert 0 < e < 5e-2, e
    # fp8_attention="all" (round 6): the bicross attention (zero-padded heads) and the VGGT frame / global attention (head_dim 64) as well
    e = rel_l2(outs["fp8_every_attention"], outs["fp8_all"])
    assert 0 < e < 5e-2, e

    class NoFp8Attention(TorchRefOps):
        attention_fp8 = property()            # hasattr(...) is False
    with pytest.raises(ValueError, match="HIP op set"):
        FusionEngine(case_l2.cfg, case_l2.weights.__getitem__, NoFp8Attention(), precision="fp8", fp8_attention=True)
