"""-m gpu: every C-ABI kernel against the plain-PyTorch fp32 statement of the same op (oracle/ref_ops.py), on identical
bf16-rounded inputs.  Tolerances (relative L2 over the whole output, stated per test):
  * fp32-accumulate GEMM / attention writing bf16: 4e-3 (one bf16 rounding of the output = 2^-9 ~ 2e-3 per element,
    plus P rounded to bf16 inside attention); writing fp32: 1e-3 (north-star op-level target)
  * element-wise/normalisation kernels writing bf16: 4e-3; fp32 outputs: 1e-5.
"""
import math

import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from fantasy_world_amd.hip_ops import HipOps
    return HipOps("cuda:0")


@pytest.fixture(scope="module")
def ref():
    from oracle.ref_ops import TorchRefOps
    return TorchRefOps()


def bf(t):
    return t.to(torch.bfloat16)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).float()


# --------------------------------------------------------------------------------------------------------- GEMM
GEMM_SHAPES = [(300, 200, 128), (1, 5120, 256), (257, 1280, 1280), (1000, 448, 2048), (513, 64, 5120),
               (128, 128, 64), (129, 130, 192), (2048, 1152, 1024), (777, 3072, 1024)]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_plain_bf16_out(ops, ref, M, N, K, parity, request):
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3, scale=0.1)
    want = ref.linear(x, ref.pack_linear(w, b))
    got = ops.linear(bf(x).cuda(), ops.pack_linear(w, b))
    assert got.dtype == torch.bfloat16 and got.shape == (M, N)
    parity.check(f"op/{request.node.name}/0", rel_l2(got.float(), want), 4e-3)


@pytest.mark.parametrize("act", ["relu", "gelu_tanh", "gelu_erf", "silu"])
def test_gemm_activations(ops, ref, act, parity, request):
    M, N, K = 333, 384, 256
    x, w, b = rnd(M, K, seed=4), rnd(N, K, seed=5, scale=K ** -0.5), rnd(N, seed=6, scale=0.1)
    want = ref.linear(x, ref.pack_linear(w, b), act=act)
    got = ops.linear(bf(x).cuda(), ops.pack_linear(w, b), act=act, out_f32=True)
    parity.check(f"op/{request.node.name}/0", rel_l2(got, want), 1e-3)


@pytest.mark.parametrize("res_dtype", ["f32", "bf16"])
def test_gemm_affine_residual_epilogue(ops, ref, res_dtype, parity, request):
    """y = res + (acc+bias)*g1 + g0: gates (DIT21:246-251), LayerScale, VGGT post-MLP modulation (VB:78-81)."""
    M, N, K = 515, 1024, 4096
    x, w, b = rnd(M, K, seed=7), rnd(N, K, seed=8, scale=K ** -0.5), rnd(N, seed=9, scale=0.1)
    g1, g0 = rnd(N, seed=10), rnd(N, seed=11)
    res = rnd(M, N, seed=12) if res_dtype == "bf16" else torch.randn(M, N, generator=torch.Generator().manual_seed(12))
    want = ref.linear(x, ref.pack_linear(w, b), g1=g1, g0=g0, res=res, out_f32=True)
    r = res.cuda().to(torch.bfloat16 if res_dtype == "bf16" else torch.float32)
    if res_dtype == "f32":
        got = ops.linear(bf(x).cuda(), ops.pack_linear(w, b), g1=g1.cuda(), g0=g0.cuda(), res=r, out_f32=True, out=r)
        assert got.data_ptr() == r.data_ptr()          # in place on the fp32 residual stream
        parity.check(f"op/{request.node.name}/0", rel_l2(got, want), 1e-3)
    else:
        got = ops.linear(bf(x).cuda(), ops.pack_linear(w, b), g1=g1.cuda(), g0=g0.cuda(), res=r, out=r)
        parity.check(f"op/{request.node.name}/1", rel_l2(got.float(), want), 4e-3)


def test_gemm_strided_views(ops, ref, parity, request):
    """A and C may be column slices of wider buffers (fused q|k|v, k|v projections)."""
    M, K, N = 200, 128, 256
    big = rnd(M, 3 * K, seed=13)
    w, b = rnd(N, K, seed=14, scale=K ** -0.5), rnd(N, seed=15, scale=0.1)
    want = ref.linear(big[:, K:2 * K], ref.pack_linear(w, b), out_f32=True)
    outbuf = torch.zeros(M, 2 * N, dtype=torch.float32, device="cuda")
    ops.linear(bf(big).cuda()[:, K:2 * K], ops.pack_linear(w, b), out_f32=True, out=outbuf[:, N:])
    parity.check(f"op/{request.node.name}/0", rel_l2(outbuf[:, N:], want), 1e-3)
    assert outbuf[:, :N].abs().max().item() == 0.0


# the 256x256 kernels (M >= 2048, N >= 1024): 5-deep half-slab ring (default) and the 2-stage staggered kernel accumulate
# in the same k order, so they must agree BIT FOR BIT with each other, and with the fp32 reference to rounding.
RING_SHAPES = [(2048, 1024, 256), (2300, 1280, 320), (4095, 1152, 1024), (2051, 2304, 576), (2560, 1024, 4096),
               (2304, 1280, 2048), (4096 + 97, 1536, 1344)]


@pytest.mark.parametrize("M,N,K", RING_SHAPES)
def test_gemm_big_tile_kernels_agree(ops, ref, M, N, K, parity):
    """The two 256x256 kernels -- 8-wave ping-pong (default) and four-wave 128x128-wave-tile (FW_GEMM_KERNEL=5 selects
    it) -- against the fp32 reference and against each other: same k-order per output element, so they agree BIT FOR BIT."""
    x, w, b = rnd(M, K, seed=21), rnd(N, K, seed=22, scale=K ** -0.5), rnd(N, seed=23, scale=0.1)
    want = ref.linear(x, ref.pack_linear(w, b), out_f32=True)
    lin = ops.pack_linear(w, b)
    xd = bf(x).cuda()
    try:
        outs = {}
        for kern in (4, 5, 7, 9):
            ops.set_option("gemm_kernel", kern)
            outs[kern] = ops.linear(xd, lin, out_f32=True)
        torch.cuda.synchronize()
    finally:
        ops.set_option("gemm_kernel", 9)
    parity.check(f"op/gemm_big_f32out/{M}x{N}x{K}", rel_l2(outs[9], want), 1e-3)
    for kern in (4, 5, 7):             # 9 = the default (two-slot ping-pong, round 4); 4 = four-slot ping-pong; 5 = four waves (LDS-DMA);
                                       # 7 = four waves with VGPR-staged loads (round 6)
        assert torch.equal(outs[kern], outs[9]), kern


@pytest.mark.parametrize("kern", [4, 5, 9])
@pytest.mark.parametrize("res_dtype,out_f32", [("f32", True), ("bf16", False)])
def test_gemm_big_tile_fused_epilogue(ops, ref, res_dtype, out_f32, kern, parity):
    """gelu + per-column affine + residual on the big-tile path, ragged M and N tails, in place on the residual."""
    M, N, K = 2333, 1028, 512
    x, w, b = rnd(M, K, seed=31), rnd(N, K, seed=32, scale=K ** -0.5), rnd(N, seed=33, scale=0.1)
    g1, g0 = rnd(N, seed=34), rnd(N, seed=35)
    res = rnd(M, N, seed=36) if res_dtype == "bf16" else torch.randn(M, N, generator=torch.Generator().manual_seed(36))
    want = ref.linear(x, ref.pack_linear(w, b), act="gelu_tanh", g1=g1, g0=g0, res=res, out_f32=True)
    r = res.cuda().to(torch.bfloat16 if res_dtype == "bf16" else torch.float32)
    ops.set_option("gemm_kernel", kern)
    try:
        got = ops.linear(bf(x).cuda(), ops.pack_linear(w, b), act="gelu_tanh", g1=g1.cuda(), g0=g0.cuda(), res=r,
                         out_f32=out_f32, out=r)
        torch.cuda.synchronize()
    finally:
        ops.set_option("gemm_kernel", 9)
    parity.check(f"op/gemm_big_epilogue/k{kern}/{res_dtype}", rel_l2(got.float(), want), 1e-3 if out_f32 else 4e-3)


@pytest.mark.parametrize("N,K", [(1024, 1024), (1536, 2048)])
@pytest.mark.parametrize("act", ["none", "gelu_tanh"])
def test_gemm_rows_do_not_depend_on_their_tile(ops, N, K, act):
    """A row's result must not depend on WHICH kernel computed it: M = 2 * 301 puts sample 0's last 45 rows in a 256-row tile and
    sample 1's first 211 rows beside them, while the same rows computed alone fall into the 128x128 tail kernel (and vice versa).
    With bias + activation + per-column affine + fp32 residual the stacked call equals the two separate calls BIT FOR BIT -- the
    property the merged CFG pass (two samples stacked along the rows) rests on.  (K = 1024: four-wave kernel; 2048: ping-pong.)"""
    M = 301
    x = bf(rnd(2 * M, K, seed=41)).cuda()
    w, b = rnd(N, K, seed=42, scale=K ** -0.5), rnd(N, seed=43, scale=0.1)
    g1, g0 = rnd(N, seed=44).cuda(), rnd(N, seed=45).cuda()
    res = torch.randn(2 * M, N, generator=torch.Generator().manual_seed(46)).cuda()
    lin = ops.pack_linear(w, b)
    kw = dict(g1=g1, g0=g0, out_f32=True)
    if act != "none":
        kw["act"] = act
    both = ops.linear(x, lin, res=res.clone(), **kw)
    for i in range(2):
        alone = ops.linear(x[i * M:(i + 1) * M].contiguous(), lin, res=res[i * M:(i + 1) * M].clone(), **kw)
        assert torch.equal(both[i * M:(i + 1) * M], alone), f"sample {i}"


# DiT token counts of BASELINE configs 2 / 4 / 5 (81f x 480 x 832; 81f x 720 x 1280; 121f x 720 x 1280) and of config 5's merged CFG pass
# (both samples stacked along the rows: 2 x 111 600; M x N passes 2^31 elements there)
FULL_M = [pytest.param(32760, id="cfg2_L32760"), pytest.param(75600, id="cfg4_L75600"), pytest.param(111600, id="cfg5_L111600"),
          pytest.param(223200, id="cfg5_merged_2x111600")]


@pytest.mark.parametrize("M", FULL_M)
@pytest.mark.parametrize("N,K,tag", [(15360, 5120, "qkv"), (13824, 5120, "ffn0"), (5120, 13824, "ffn2")])
def test_gemm_full_size_properties(ops, M, N, K, tag, parity):
    """BASELINE config-2 / 4 / 5 shapes (M = L rows; every tile, every k-slab of the kernel the forward runs), two size-independent
    properties instead of a CPU reference:
      * selection, BIT-EXACT: with one-hot rows (row m has a single 1.0 at column j(m)) the GEMM returns W[:, j(m)] + bias -- every
        product is exact and every other term is 0, so fp32 out equals the gathered weights bit for bit and bf16 out their rounding;
        a wrong row / column / k-slab address anywhere in the grid shows;
      * a checksum of checksums on random data: sum_mn out[m, n] = sum_k (sum_m x[m, k]) (sum_n W[n, k]) + M sum_n b[n], in fp64."""
    g = torch.Generator(device="cuda").manual_seed(5)
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g) * 0.1
    lin = ops.pack_linear(w.float().cpu(), b.cpu())
    j = (torch.arange(M, device="cuda") * 7919 + 13) % K
    x = torch.zeros(M, K, dtype=torch.bfloat16, device="cuda")
    x[torch.arange(M, device="cuda"), j] = 1.0
    want = w.float().t()[j] + b                                     # [M, N] fp32, exact
    got = ops.linear(x, lin, out_f32=True)
    assert torch.equal(got, want)
    got16 = ops.linear(x, lin)
    assert torch.equal(got16, want.to(torch.bfloat16))
    del got, got16, want, x
    x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    out = ops.linear(x, lin, out_f32=True)
    total = out.double().sum().item()
    expect = (x.double().sum(0) * w.double().sum(0)).sum().item() + M * b.double().sum().item()
    scale = out.double().abs().sum().item()
    parity.check(f"op/gemm_full_size_checksum/{tag}" + ("" if M == 32760 else f"/M{M}"), abs(total - expect) / scale, 1e-6)
    del out, x
    torch.cuda.empty_cache()


def test_empty_inputs(ops):
    """Zero rows are a no-op with the right shapes (no launch with an empty grid, no error); zero KEYS have no softmax and are
    refused by name."""
    lin = ops.pack_linear(rnd(256, 128, seed=1), torch.zeros(256))
    e = torch.zeros(0, 128, dtype=torch.bfloat16, device="cuda")
    k = bf(rnd(64, 128, seed=2)).cuda()
    assert ops.linear(e, lin).shape == (0, 256)
    assert ops.layernorm(torch.zeros(0, 1024, device="cuda"), eps=1e-6).shape == (0, 1024)
    assert ops.attention(e, k, k, 1, 128).shape == (0, 128)
    ops.qk_prep(e, 1, 128, "rms_full", torch.ones(128, device="cuda"), None, 1e-6)
    assert ops.residual_add(torch.zeros(0, 256, device="cuda"), torch.zeros(0, 256, dtype=torch.bfloat16, device="cuda")).shape == (0, 256)
    with pytest.raises(RuntimeError, match="Lk must be > 0"):
        ops.attention(k, e, e, 1, 128)
    torch.cuda.synchronize()


def test_gemm_rejects_bad_k(ops):
    x = torch.zeros(4, 100, dtype=torch.bfloat16, device="cuda")
    from fantasy_world_amd.hip_ops import Linear
    lin = Linear(torch.zeros(8, 100, dtype=torch.bfloat16, device="cuda"), None)
    with pytest.raises(RuntimeError):
        ops.linear(x, lin)


def test_gemv_f32(ops, ref, parity, request):
    K, N = 256, 5120
    x = torch.randn(K, generator=torch.Generator().manual_seed(1))
    w, b = torch.randn(N, K, generator=torch.Generator().manual_seed(2)) * K ** -0.5, torch.randn(N) * 0.1
    for silu_in, act in [(False, "silu"), (True, None)]:
        want = ref.linear_f32(x, ref.pack_linear_f32(w, b), silu_in=silu_in, act=act)
        got = ops.linear_f32(x.cuda(), ops.pack_linear_f32(w, b), silu_in=silu_in, act=act)
        parity.check(f"op/{request.node.name}/0", rel_l2(got, want), 1e-5)


# --------------------------------------------------------------------------------------------------------- attention
DEFAULT_ATTN_VAR = 192


@pytest.fixture(params=[0, 64, 129, 131, 193, 195, 192],
                ids=["attn_v0", "attn_pp3", "attn_sp", "attn_sp_w4", "attn_sp_unrolled", "attn_sp_w4_unrolled", "attn_default"])
def attn_variant(ops, request):
    """Every attention test runs on the generic first kernel (0), on the two-segment ping-pong kernel (64), on the single-stream
    kernel in its 8-wave (129) and 4-wave (131) forms, on their ring-unrolled forms (193 / 195: compile-time LDS ring slots; hd 96
    falls back to the rolled form) and on the default per-head-dim choice (192); 64+ are selected when q carries the softmax scale
    (q_prescaled=True)."""
    ops.set_option("attn_var", request.param)
    yield request.param
    ops.set_option("attn_var", DEFAULT_ATTN_VAR)


def _prescale(ops, variant, q, hd):
    """variant >= 64: fold softmax_scale*log2(e) into q (what qk_prep's out_scale does), one bf16 rounding."""
    if variant >= 64:
        return (q * ops.q_scale(hd)).to(torch.bfloat16).float(), True
    return q, False


ATTN_CASES = [  # heads, hd, batch, Lq, Lk
    (3, 128, 1, 300, 333), (2, 128, 1, 64, 64), (5, 128, 1, 1000, 257), (2, 128, 1, 257, 512),
    (4, 96, 1, 290, 305), (12, 96, 1, 48, 63), (3, 64, 1, 500, 129), (16, 64, 3, 133, 133), (2, 64, 2, 31, 1),
]


@pytest.mark.parametrize("heads,hd,batch,Lq,Lk", ATTN_CASES)
def test_attention_matches_softmax_reference(attn_variant, ops, ref, heads, hd, batch, Lq, Lk, parity, request):
    q, k, v = rnd(batch * Lq, heads * hd, seed=1), rnd(batch * Lk, heads * hd, seed=2), rnd(batch * Lk, heads * hd, seed=3)
    q, pre = _prescale(ops, attn_variant, q, hd)
    want = ref.attention(q, k, v, heads, hd, batch=batch, q_prescaled=pre)
    got = ops.attention(bf(q).cuda(), bf(k).cuda(), bf(v).cuda(), heads, hd, batch=batch, q_prescaled=pre)
    parity.check(f"op/{request.node.name}/0", rel_l2(got.float(), want), 4e-3)


@pytest.mark.parametrize("heads,hd,Lq,Lk", [(16, 64, 256 * 16 + 97, 256 * 16 + 97), (12, 96, 256 * 64 + 97, 2100)])
def test_attention_split_kv_plan_does_not_depend_on_the_batch(ops, heads, hd, Lq, Lk):
    """Round-2 advisor finding: the number of key runs of the split-KV tail came from 256 / (heads * batch), so the merged CFG pass
    (batch 2) merged a tail row's runs in another order than two separate forwards (batch 1) -- joint_forward_pair was not
    bit-identical to two joint_forward calls at shapes that take the split route (VGGT global attention at L2 = 32865).  The plan
    is now a function of (heads, Lq, Lk) only: two samples stacked along the batch equal two separate calls BIT FOR BIT, tail rows
    included."""
    assert ops.lib.fw_attention_workspace_bytes(1, heads, hd, Lq, Lk) > 0
    assert ops.lib.fw_attention_workspace_bytes(2, heads, hd, Lq, Lk) == 2 * ops.lib.fw_attention_workspace_bytes(1, heads, hd, Lq, Lk)
    g = torch.Generator(device="cuda").manual_seed(17)
    D = heads * hd
    q = (torch.randn(2 * Lq, D, device="cuda", generator=g) * ops.q_scale(hd)).to(torch.bfloat16)
    k = torch.randn(2 * Lk, D, device="cuda", generator=g).to(torch.bfloat16)
    v = torch.randn(2 * Lk, D, device="cuda", generator=g).to(torch.bfloat16)
    both = ops.attention(q, k, v, heads, hd, batch=2, q_prescaled=True)
    one = [ops.attention(q[i * Lq:(i + 1) * Lq], k[i * Lk:(i + 1) * Lk], v[i * Lk:(i + 1) * Lk], heads, hd, q_prescaled=True) for i in range(2)]
    assert torch.equal(both[:Lq], one[0]) and torch.equal(both[Lq:], one[1])


@pytest.mark.parametrize("heads,hd,Lq,Lk", [(12, 96, 256 * 64 + 97, 1100), (16, 64, 256 * 16 + 30, 2048), (40, 128, 256 * 32 + 5, 1030)])
def test_attention_split_kv_tail(ops, ref, heads, hd, Lq, Lk, parity, request):
    """Tail q-block through split-KV (the launcher takes the route when the tail work-groups would cost an extra round of the
    256 CUs and the caller passes a workspace): the tail rows against the fp32 reference, the whole output against the
    single-launch path, the accumulate flag, and a spiked key in one run (its shift differs from the other runs' by ~2^100)."""
    assert ops.lib.fw_attention_workspace_bytes(1, heads, hd, Lq, Lk) > 0
    assert ops.lib.fw_attention_workspace_bytes(1, heads, hd, Lq - Lq % 256, Lk) == 0
    g = torch.Generator(device="cuda").manual_seed(11)
    D = heads * hd
    q = (torch.randn(Lq, D, device="cuda", generator=g) * ops.q_scale(hd)).to(torch.bfloat16)
    k = torch.randn(Lk, D, device="cuda", generator=g).to(torch.bfloat16)
    v = torch.randn(Lk, D, device="cuda", generator=g).to(torch.bfloat16)
    k[Lk - 40, :hd] = (q[Lq - 3, :hd].float() * 60.0).to(torch.bfloat16)      # head 0: a late key far above everything else for one tail row
    got = ops.attention(q, k, v, heads, hd, q_prescaled=True)
    try:
        ops.split_kv = False
        base = ops.attention(q, k, v, heads, hd, q_prescaled=True)
    finally:
        ops.split_kv = True
    tail0 = Lq - Lq % 256
    assert torch.equal(got[:tail0], base[:tail0])                             # the full q-blocks are the same launch either way
    rows = torch.arange(tail0, Lq, device="cuda")
    want = ref.attention(q[rows].float().cpu(), k.float().cpu(), v.float().cpu(), heads, hd, q_prescaled=True)
    parity.check(f"op/attention_split_kv_tail/hd{hd}", rel_l2(got[rows].float(), want), 4e-3)
    parity.check(f"op/attention_single_launch_tail/hd{hd}", rel_l2(base[rows].float(), want), 4e-3)
    acc = got.clone()
    ops.attention(q, k, v, heads, hd, out=acc, accumulate=True, q_prescaled=True)
    parity.check(f"op/{request.node.name}/0", rel_l2(acc[rows].float(), 2 * want), 5e-3)


def test_attention_strided_qkv_and_accumulate(attn_variant, ops, ref, parity, request):
    """q/k/v as column slices of one fused buffer; second call accumulates (cross-attn text + image, DIT21:197-200)."""
    heads, hd, L, Lc = 4, 128, 200, 77
    D = heads * hd
    qkv = rnd(L, 3 * D, seed=4)
    qkv[:, :D], pre = _prescale(ops, attn_variant, qkv[:, :D].clone(), hd)
    ctx = rnd(Lc, 2 * D, seed=5)
    want = ref.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], heads, hd, q_prescaled=pre)
    want2 = want + ref.attention(qkv[:, :D], ctx[:, :D], ctx[:, D:], heads, hd, q_prescaled=pre)
    g = bf(qkv).cuda()
    c = bf(ctx).cuda()
    got = ops.attention(g[:, :D], g[:, D:2 * D], g[:, 2 * D:], heads, hd, q_prescaled=pre)
    parity.check(f"op/{request.node.name}/0", rel_l2(got.float(), want), 4e-3)
    ops.attention(g[:, :D], c[:, :D], c[:, D:], heads, hd, out=got, accumulate=True, q_prescaled=pre)
    parity.check(f"op/{request.node.name}/1", rel_l2(got.float(), want2), 5e-3)


@pytest.mark.parametrize("heads,hd,batch,Lq,Lk", [(3, 128, 1, 300, 333), (5, 128, 2, 520, 257), (12, 96, 1, 290, 305), (4, 96, 2, 100, 64),
                                                   (16, 64, 3, 133, 133), (2, 64, 1, 700, 1), (2, 128, 1, 64, 2048)])
def test_attention_descriptor_requests_return_the_bits_of_the_pointer_form(ops, heads, hd, batch, Lq, Lk):
    """Round 4: the production kernels request their K / Vt tiles through an SGPR buffer descriptor (rows past the last key are out of
    its range: zeros, masked anyway; the ping-pong kernel also requests tiles past the last one); FW_ATTN_VAR bit 10 selects the
    round-3 pointer form (clamped rows, conditional requests).  Same bits -- ragged last tiles, one-key sequences, batches and
    k | v column slices of one buffer included."""
    D = heads * hd
    q = bf(rnd(batch * Lq, D, seed=41) * ops.q_scale(hd)).cuda()
    kv = bf(rnd(batch * Lk, 2 * D, seed=42)).cuda()
    outs = []
    try:
        for var in (DEFAULT_ATTN_VAR, DEFAULT_ATTN_VAR + 1024):
            ops.set_option("attn_var", var)
            outs.append(ops.attention(q, kv[:, :D], kv[:, D:], heads, hd, batch=batch, q_prescaled=True).clone())
    finally:
        ops.set_option("attn_var", DEFAULT_ATTN_VAR)
    assert torch.equal(outs[0], outs[1])
    assert torch.isfinite(outs[0].float()).all()


@pytest.mark.parametrize("heads,hd,Lq,Lk", [(3, 128, 700, 1333), (12, 96, 515, 1029), (16, 64, 600, 2100), (2, 128, 256, 64), (2, 64, 31, 7)])
def test_attention_matrix_pipe_row_sums_against_vector_pipe_sums(ops, ref, heads, hd, Lq, Lk, parity, request):
    """Round 6: the single-stream kernels sum the bf16-ROUNDED probabilities on the matrix pipe (v_mfma_f32_16x16x32_bf16 with a per-lane
    ones pattern, csrc/attention.hip) where round 5 summed the unrounded fp32 p on the vector pipe (FW_ATTN_VAR bit 11 keeps that
    choice, incl. the ping-pong kernel at hd 96, for the A/B).  Both against the fp32 softmax definition, and against each other: what
    separates them is P's bf16 rounding in the denominator -- a wrong lane pattern (a query summed with its partner fi ^ 16) would
    show as errors of order 1."""
    D = heads * hd
    q = bf(rnd(Lq, D, seed=51) * ops.q_scale(hd)).cuda()
    k, v = bf(rnd(Lk, D, seed=52)).cuda(), bf(rnd(Lk, D, seed=53)).cuda()
    want = ref.attention(q.cpu().float(), k.cpu().float(), v.cpu().float(), heads, hd, q_prescaled=True).float()
    outs = []
    try:
        for var in (DEFAULT_ATTN_VAR, DEFAULT_ATTN_VAR + 2048):
            ops.set_option("attn_var", var)
            outs.append(ops.attention(q, k, v, heads, hd, q_prescaled=True).float().cpu())
    finally:
        ops.set_option("attn_var", DEFAULT_ATTN_VAR)
    parity.check(f"op/{request.node.name}/matrix_pipe_sums_vs_fp32_softmax", rel_l2(outs[0], want), 8e-3)
    parity.check(f"op/{request.node.name}/vector_pipe_sums_vs_fp32_softmax", rel_l2(outs[1], want), 8e-3)
    parity.check(f"op/{request.node.name}/matrix_vs_vector_pipe_sums", rel_l2(outs[0], outs[1]), 4e-3)


@pytest.mark.parametrize("gain", [3.0, 40.0])
def test_attention_large_score_spike(attn_variant, ops, ref, gain, parity, request):
    """Online-softmax rescale path: a key whose score dwarfs the running max late in the sequence (gain 40: the spike is
    ~2^900 above everything else in the exponent domain -- the probabilities of the stale max overflow to inf and the slow
    path has to recover), a row whose max is set in tile 0 and never changes, and a row whose scores are all very negative."""
    heads, hd, Lq, Lk = 1, 128, 64, 640
    q, k, v = rnd(Lq, hd, seed=6), rnd(Lk, hd, seed=7), rnd(Lk, hd, seed=8)
    k[600] = q[5] * gain        # row 5's max jumps at tile 9
    k[10] = q[9] * (gain + 1)   # row 9's max is set in tile 0 and never changes
    k[3] = -q[20] * 2.0         # row 20 has one strongly negative score in tile 0
    k = k.to(torch.bfloat16).float()
    q, pre = _prescale(ops, attn_variant, q, hd)
    want = ref.attention(q, k, v, heads, hd, q_prescaled=pre)
    got = ops.attention(bf(q).cuda(), bf(k).cuda(), bf(v).cuda(), heads, hd, q_prescaled=pre)
    assert torch.isfinite(got.float()).all()
    parity.check(f"op/{request.node.name}/0", rel_l2(got.float(), want), 4e-3)


def test_attention_all_scores_far_below_zero(attn_variant, ops, ref, parity, request):
    """Every score of every row is << 0 (q = -20 k-ish): exp2 of the unshifted scores underflows; the first-tile max must
    anchor the running max or the row sums vanish."""
    heads, hd, Lq, Lk = 2, 64, 96, 200
    base = rnd(1, heads * hd, seed=9).abs() + 0.5
    q = base.repeat(Lq, 1) * 6.0 + 0.1 * rnd(Lq, heads * hd, seed=10)
    k = -base.repeat(Lk, 1) * 6.0 + 0.1 * rnd(Lk, heads * hd, seed=11)
    v = rnd(Lk, heads * hd, seed=12)
    q, k = q.to(torch.bfloat16).float(), k.to(torch.bfloat16).float()
    q, pre = _prescale(ops, attn_variant, q, hd)
    want = ref.attention(q, k, v, heads, hd, q_prescaled=pre)
    got = ops.attention(bf(q).cuda(), bf(k).cuda(), bf(v).cuda(), heads, hd, q_prescaled=pre)
    assert torch.isfinite(got.float()).all()
    parity.check(f"op/{request.node.name}/0", rel_l2(got.float(), want), 4e-3)


@pytest.mark.parametrize("L,batch", [pytest.param(32760, 1, id="cfg2_L32760"), pytest.param(75600, 1, id="cfg4_L75600"),
                                     pytest.param(111600, 1, id="cfg5_L111600"), pytest.param(111600, 2, id="cfg5_merged_batch2")])
def test_attention_full_length_properties(attn_variant, ops, parity, request, L, batch):
    """BASELINE config-2 / 4 / 5 sizes (L = 32760 / 75600 / 111600 keys, hd 128; batch 2 = the merged CFG pass): rows of softmax sum
    to 1 => V = const gives O = const, and sampled query rows match an fp32 evaluation of the same rows."""
    if L > 32760 and attn_variant not in (0, 64, DEFAULT_ATTN_VAR):
        pytest.skip("configs 4 / 5: the generic kernel, the ping-pong kernel and the default choice (the other variants at L = 32 760)")
    heads, hd = 2, 128
    g = torch.Generator(device="cuda").manual_seed(0)
    q = torch.randn(batch * L, heads * hd, device="cuda", generator=g).to(torch.bfloat16)
    k = torch.randn(batch * L, heads * hd, device="cuda", generator=g).to(torch.bfloat16)
    v = torch.randn(batch * L, heads * hd, device="cuda", generator=g).to(torch.bfloat16)
    pre = attn_variant >= 64
    qs = (q.float() * ops.q_scale(hd)).to(torch.bfloat16) if pre else q
    sc = 0.6931471805599453 if pre else 1.0 / math.sqrt(hd)
    ones = torch.full_like(v, 0.5)
    o1 = ops.attention(qs, k, ones, heads, hd, batch=batch, q_prescaled=pre)
    assert (o1.float() - 0.5).abs().max().item() < 4e-3
    o = ops.attention(qs, k, v, heads, hd, batch=batch, q_prescaled=pre)
    rows = torch.tensor([0, 1, 255, 256, 9999, 16383, 32503, 32759, L - 257, L - 1], device="cuda")
    for b in range(batch):                                  # a sample only attends to its own keys
        kb, vb = k[b * L:(b + 1) * L], v[b * L:(b + 1) * L]
        for h in range(heads):
            sl = slice(h * hd, (h + 1) * hd)
            s = (qs[b * L + rows][:, sl].float() @ kb[:, sl].float().t()) * sc
            want = torch.softmax(s, dim=-1) @ vb[:, sl].float()
            parity.check(f"op/{request.node.name}/0", rel_l2(o[b * L + rows][:, sl].float(), want), 6e-3)


def test_attention_rejects_bad_head_dim(attn_variant, ops):
    q = torch.zeros(8, 80, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(RuntimeError):
        ops.attention(q, q, q, 1, 80)


# --------------------------------------------------------------------------------------------------------- norms / rope
@pytest.mark.parametrize("C,rows,affine,mod,xdt", [(5120, 70, False, True, "f32"), (5120, 33, True, False, "f32"),
                                                   (1024, 129, True, True, "f32"), (1280, 257, True, False, "bf16"),
                                                   (5120, 5, True, False, "bf16"), (2048, 9, True, False, "f32"),
                                                   (1000, 7, True, True, "f32"), (8192, 3, False, True, "f32")])
def test_layernorm_mod(ops, ref, C, rows, affine, mod, xdt, parity, request):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(rows, C, generator=g) * 3 + 0.5
    if xdt == "bf16":
        x = x.to(torch.bfloat16).float()
    w = (1 + 0.1 * torch.randn(C, generator=g)) if affine else None
    b = 0.1 * torch.randn(C, generator=g) if affine else None
    sc = torch.randn(C, generator=g) if mod else None
    sh = torch.randn(C, generator=g) if mod else None
    eps = 1e-6 if not affine else 1e-5
    want = ref.layernorm(x, w, b, sc, sh, eps)
    cu = lambda t: None if t is None else t.cuda()
    xin = x.cuda().to(torch.bfloat16) if xdt == "bf16" else x.cuda()
    got = ops.layernorm(xin, cu(w), cu(b), cu(sc), cu(sh), eps)
    parity.check(f"op/{request.node.name}/0", rel_l2(got.float(), want), 4e-3)


def test_layernorm_mod_large_row_kernel_is_bit_identical_to_the_small_one(ops, ref, parity, request):
    """From 4096 rows on, the modulated LayerNorm of the fp32 DiT stream (C = 5120, no affine weights) runs on the kernel that stages
    1 + scale / shift in LDS and walks the rows; below, on the one-wave-per-row kernel.  Same arithmetic in the same order: a call over
    4100 rows equals the concatenation of two calls over 2050 rows BIT FOR BIT (a sequence shard sees the small kernel, the unsharded
    forward the large one), and both match the fp32 definition."""
    C, rows = 5120, 4100
    x = torch.randn(rows, C, generator=torch.Generator().manual_seed(51))
    sc, sh = rnd(C, seed=52, scale=0.3), rnd(C, seed=53, scale=0.3)
    want = ref.layernorm(x, scale=sc, shift=sh, eps=1e-6)
    xd, scd, shd = x.cuda(), sc.cuda(), sh.cuda()
    big = ops.layernorm(xd, scale=scd, shift=shd, eps=1e-6)
    halves = torch.cat([ops.layernorm(xd[:2050].contiguous(), scale=scd, shift=shd, eps=1e-6),
                        ops.layernorm(xd[2050:].contiguous(), scale=scd, shift=shd, eps=1e-6)], dim=0)
    assert torch.equal(big, halves)
    parity.check(f"op/{request.node.name}/0", rel_l2(big.float(), want), 4e-3)


@pytest.mark.parametrize("C,affine,mod", [(5120, True, False), (5120, True, True), (1024, True, True), (1024, True, False),
                                          (1024, False, True)])
def test_layernorm_lds_staged_forms_are_bit_identical_to_the_row_kernel(ops, ref, C, affine, mod, parity, request):
    """Round 3: every parameterised LayerNorm form of the forward stages its vectors in LDS from 4096 rows on (DiT norm3 = affine
    only; VGGT norm1 = affine + modulation, norm2 = affine only).  Same ln_finish arithmetic as the one-wave-per-row kernel: a call
    over 4100 rows equals two calls over 2050 rows BIT FOR BIT (a sequence shard runs the small kernel), and matches fp32."""
    rows = 4100
    g = torch.Generator().manual_seed(61)
    x = torch.randn(rows, C, generator=g) * 2 + 0.3
    w, b = ((1 + 0.1 * torch.randn(C, generator=g)), 0.1 * torch.randn(C, generator=g)) if affine else (None, None)
    sc, sh = (0.3 * torch.randn(C, generator=g), 0.3 * torch.randn(C, generator=g)) if mod else (None, None)
    want = ref.layernorm(x, w, b, sc, sh, 1e-5)
    cu = lambda t: None if t is None else t.cuda()
    xd = x.cuda()
    kw = dict(w=cu(w), b=cu(b), scale=cu(sc), shift=cu(sh), eps=1e-5)
    big = ops.layernorm(xd, **kw)
    halves = torch.cat([ops.layernorm(xd[:2050].contiguous(), **kw), ops.layernorm(xd[2050:].contiguous(), **kw)], dim=0)
    assert torch.equal(big, halves)
    parity.check(f"op/{request.node.name}/0", rel_l2(big.float(), want), 4e-3)


def test_qk_prep_rms_full_rope3d(ops, ref, parity, request):
    """DiT q/k: RMSNorm over the full 5120 width (DIT21:170-171) + interleaved 3-D RoPE (DIT21:97-102)."""
    from fantasy_world_amd import rope
    heads, hd = 40, 128
    f, h, w = 2, 3, 5
    L = f * h * w
    tab = rope.rope3d_table(hd, f, h, w)
    x = rnd(L, 3 * heads * hd, seed=9)
    nw = 1 + 0.1 * torch.randn(heads * hd, generator=torch.Generator().manual_seed(1))
    xr = x.clone()
    ref.qk_prep(xr[:, heads * hd:2 * heads * hd], heads, hd, "rms_full", nw, None, 1e-6, "interleaved", tab)
    xg = bf(x).cuda()
    ops.qk_prep(xg[:, heads * hd:2 * heads * hd], heads, hd, "rms_full", nw.cuda(), None, 1e-6, "interleaved", tab.cuda())
    parity.check(f"op/{request.node.name}/0", rel_l2(xg.float(), xr), 4e-3)
    assert torch.equal(xg[:, :heads * hd].float().cpu(), x[:, :heads * hd])        # neighbours untouched


def test_qk_prep_ln_head_rope2d(ops, ref, parity, request):
    """VGGT q/k: per-head LayerNorm(64) (VA:43-44) + 2-D rotate-half RoPE base 100 (VR:154-188), table row = row % P."""
    from fantasy_world_amd import rope
    heads, hd, S, h, w = 16, 64, 3, 4, 5
    P = 5 + h * w
    tab = rope.rope2d_table(hd, h, w, 5)
    x = rnd(S * P, heads * hd, seed=10)
    g = torch.Generator().manual_seed(2)
    nw, nb = 1 + 0.1 * torch.randn(hd, generator=g), 0.1 * torch.randn(hd, generator=g)
    xr = x.clone()
    ref.qk_prep(xr, heads, hd, "ln_head", nw, nb, 1e-5, "half2d", tab)
    xg = bf(x).cuda()
    ops.qk_prep(xg, heads, hd, "ln_head", nw.cuda(), nb.cuda(), 1e-5, "half2d", tab.cuda())
    parity.check(f"op/{request.node.name}/0", rel_l2(xg.float(), xr), 4e-3)


@pytest.mark.parametrize("F,gh,gw", [pytest.param(21, 30, 52, id="cfg2_81f_480x832"), pytest.param(21, 45, 80, id="cfg4_81f_720x1280"),
                                     pytest.param(31, 45, 80, id="cfg5_121f_720x1280")])
def test_row_passes_at_full_size_on_sampled_rows(ops, ref, parity, F, gh, gw):
    """BASELINE config-2 / 4 / 5 sizes for the row-wise passes around the GEMMs (every row is independent, so a sample of rows pins the
    launch at full size against the CPU oracle ops): the modulated LayerNorm of the fp32 DiT stream [L, 5120], the affine +
    modulated LayerNorm of the VGGT stream [L2, 1024], the DiT q/k pass (RMSNorm over 5120 + interleaved 3-D RoPE + q scale) on
    [L, 5120] and the VGGT q/k pass (per-head LayerNorm + 2-D RoPE) on two whole frames of [L2, 1024]."""
    from fantasy_world_amd import rope
    g = torch.Generator(device="cuda").manual_seed(71)
    P = 5 + gh * gw
    L, L2 = F * gh * gw, F * P
    tag = "" if (F, gh, gw) == (21, 30, 52) else f"/L{L}"
    rows = torch.tensor([0, 1, 63, 64, 255, 256, 4095, 4096, 8191, 16384, 20000, 32503, L - 2, L - 1])
    # LayerNorm, DiT stream
    x = torch.randn(L, 5120, device="cuda", generator=g) * 2 + 0.3
    sc, sh = rnd(5120, seed=72, scale=0.3), rnd(5120, seed=73, scale=0.3)
    got = ops.layernorm(x, scale=sc.cuda(), shift=sh.cuda(), eps=1e-6)
    want = ref.layernorm(x[rows.cuda()].cpu(), scale=sc, shift=sh, eps=1e-6)
    parity.check("op/full_size_rows/layernorm_mod_5120" + tag, rel_l2(got[rows.cuda()].float(), want), 4e-3)
    # a constant row has zero variance: the output is the shift, exactly (values with exact fp32 sums)
    x[rows.cuda()] = 0.5
    got = ops.layernorm(x, scale=sc.cuda(), shift=sh.cuda(), eps=1e-6)
    assert torch.equal(got[rows.cuda()].cpu(), sh.to(torch.bfloat16).expand(len(rows), -1))
    del x, got
    # LayerNorm, VGGT stream (affine + modulation)
    t = torch.randn(L2, 1024, device="cuda", generator=g)
    w, b = 1 + rnd(1024, seed=74, scale=0.1), rnd(1024, seed=75, scale=0.1)
    sc, sh = rnd(1024, seed=76, scale=0.3), rnd(1024, seed=77, scale=0.3)
    rows2 = torch.cat([rows, torch.tensor([L2 - 105, L2 - 1])])
    got = ops.layernorm(t, w.cuda(), b.cuda(), sc.cuda(), sh.cuda(), 1e-5)
    want = ref.layernorm(t[rows2.cuda()].cpu(), w, b, sc, sh, 1e-5)
    parity.check("op/full_size_rows/layernorm_affine_mod_1024" + tag, rel_l2(got[rows2.cuda()].float(), want), 4e-3)
    del t, got
    # DiT q pass: 40 heads x 128, full-width RMSNorm, 3-D RoPE, q pre-scale
    heads, hd = 40, 128
    tab = rope.rope3d_table(hd, F, gh, gw)
    q = torch.randn(L, heads * hd, device="cuda", generator=g).to(torch.bfloat16)
    nw = 1 + rnd(heads * hd, seed=78, scale=0.1)
    qr = q[rows.cuda()].float().cpu()
    ref.qk_prep(qr, heads, hd, "rms_full", nw, None, 1e-6, "interleaved", tab[rows], out_scale=ops.q_scale(hd))
    ops.qk_prep(q, heads, hd, "rms_full", nw.cuda(), None, 1e-6, "interleaved", tab.cuda(), out_scale=ops.q_scale(hd))
    parity.check("op/full_size_rows/qk_prep_rms_rope3d" + tag, rel_l2(q[rows.cuda()].float(), qr), 4e-3)
    del q
    # VGGT k pass: 16 heads x 64, per-head LayerNorm, 2-D RoPE with table row = row % P: the first and the last frame whole
    heads, hd = 16, 64
    tab2 = rope.rope2d_table(hd, gh, gw, 5)
    k = torch.randn(L2, heads * hd, device="cuda", generator=g).to(torch.bfloat16)
    nw, nb = 1 + rnd(hd, seed=79, scale=0.1), rnd(hd, seed=80, scale=0.1)
    fr = torch.cat([torch.arange(0, P), torch.arange((F - 1) * P, F * P)])
    kr = k[fr.cuda()].float().cpu()
    ref.qk_prep(kr, heads, hd, "ln_head", nw, nb, 1e-5, "half2d", tab2)
    ops.qk_prep(k, heads, hd, "ln_head", nw.cuda(), nb.cuda(), 1e-5, "half2d", tab2.cuda())
    parity.check("op/full_size_rows/qk_prep_ln_head_rope2d" + tag, rel_l2(k[fr.cuda()].float(), kr), 4e-3)


def test_qk_prep_rope_only_hd96_with_identity_rows(ops, ref, parity, request):
    """bicross k: no norm, RoPE-3D hd=96 with 5 un-rotated special tokens per frame (DIT21:105-132)."""
    from fantasy_world_amd import rope
    heads, hd, f, h, w = 12, 96, 2, 3, 4
    tab = rope.rope3d_table_with_extra(hd, f, h, w, 5)
    rows = f * (5 + h * w)
    x = rnd(rows, heads * hd, seed=11)
    xr = x.clone()
    ref.qk_prep(xr, heads, hd, None, None, None, 1e-6, "interleaved", tab)
    xg = bf(x).cuda()
    ops.qk_prep(xg, heads, hd, None, None, None, 1e-6, "interleaved", tab.cuda())
    parity.check(f"op/{request.node.name}/0", rel_l2(xg.float(), xr), 4e-3)
    assert torch.equal(xg[:5].float().cpu(), x[:5])                                # identity rows bit-exact


def test_qk_prep_out_scale(ops, ref, parity, request):
    """out_scale is applied in fp32 before the single bf16 rounding (q carries softmax_scale*log2e into attention)."""
    from fantasy_world_amd import rope
    heads, hd, f, h, w = 40, 128, 1, 3, 4
    tab = rope.rope3d_table(hd, f, h, w)
    x = rnd(f * h * w, heads * hd, seed=13)
    nw = 1 + 0.1 * torch.randn(heads * hd, generator=torch.Generator().manual_seed(1))
    xr = x.clone()
    ref.qk_prep(xr, heads, hd, "rms_full", nw, None, 1e-6, "interleaved", tab, out_scale=ops.q_scale(hd))
    xg = bf(x).cuda()
    ops.qk_prep(xg, heads, hd, "rms_full", nw.cuda(), None, 1e-6, "interleaved", tab.cuda(), out_scale=ops.q_scale(hd))
    parity.check(f"op/{request.node.name}/0", rel_l2(xg.float(), xr), 4e-3)
    # and through the generic (block-per-row) kernel: odd head count -> width not a multiple of 64 chunks is still wave path;
    # force the fallback with a misaligned table pointer
    tab2 = torch.cat([torch.zeros(1), tab.reshape(-1)]).cuda()[1:].view(tab.shape)
    xg2 = bf(x).cuda()
    ops.qk_prep(xg2, heads, hd, "rms_full", nw.cuda(), None, 1e-6, "interleaved", tab2, out_scale=ops.q_scale(hd))
    parity.check(f"op/{request.node.name}/1", rel_l2(xg2.float(), xr), 4e-3)


def test_qk_prep_rms_no_rope(ops, ref, parity, request):
    heads, hd = 40, 128
    x = rnd(50, heads * hd, seed=12)
    nw = 1 + 0.1 * torch.randn(heads * hd, generator=torch.Generator().manual_seed(1))
    xr = x.clone()
    ref.qk_prep(xr, heads, hd, "rms_full", nw, None, 1e-6)
    xg = bf(x).cuda()
    ops.qk_prep(xg, heads, hd, "rms_full", nw.cuda(), None, 1e-6)
    parity.check(f"op/{request.node.name}/0", rel_l2(xg.float(), xr), 4e-3)


# --------------------------------------------------------------------------------------------------------- layout glue
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_control_patchify_is_an_exact_gather(ops, ref, dtype):
    """Wan2.2 control adapter front end (PixelUnshuffle(8) + k2s2 patches): pure data movement, bit exact."""
    C, F_, h, w = 24, 2, 3, 5
    ctl = rnd(1, C, F_, 16 * h, 16 * w, seed=41)
    want = ref.control_patchify(ctl)
    got = ops.control_patchify(ctl.to(dtype).cuda())
    assert got.shape == (F_ * h * w, C * 256) and torch.equal(got.float().cpu(), want)


def test_im2col3x3_is_an_exact_gather(ops, ref, parity, request):
    """3x3 / pad 1 im2col on token-major activations (ResidualBlock convs of the control adapter), zero borders, bit exact."""
    F_, h, w, C = 2, 4, 6, 64
    x = rnd(F_ * h * w, C, seed=42)
    want = ref.im2col3x3(x, F_, h, w)
    got = ops.im2col3x3(bf(x).cuda(), F_, h, w)
    assert torch.equal(got.float().cpu(), want)
    # and it is the convolution: im2col @ W^T == conv2d
    wt = rnd(8, C, 3, 3, seed=43, scale=0.1)
    conv = torch.nn.functional.conv2d(x.view(F_, h, w, C).permute(0, 3, 1, 2), wt, padding=1).permute(0, 2, 3, 1).reshape(-1, 8)
    parity.check(f"op/{request.node.name}/0", rel_l2(got.float().cpu() @ wt.reshape(8, -1).t(), conv), 1e-5)



def test_sinusoid_bf16_timestep(ops, ref):
    """The sampler hands joint_forward a bf16 timestep (M21:292-293): 999.x -> 1000."""
    for t in (torch.tensor([937.5]), torch.tensor([999.3]).to(torch.bfloat16), torch.tensor([0.0])):
        want = ref.sinusoid(t.float() if t.dtype == torch.bfloat16 else t, 256)
        got = ops.sinusoid(t.cuda(), 256)
        assert (got.cpu() - want).abs().max().item() < 2e-6


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_patchify_unpatchify(ops, ref, dt):
    F_, H2, W2 = 3, 8, 12
    x, y = rnd(1, 16, F_, H2, W2, seed=1), rnd(1, 20, F_, H2, W2, seed=2)
    want = ref.patchify(x, y, 192)
    got = ops.patchify(x.cuda().to(dt), y.cuda().to(dt), 192)
    assert torch.equal(got.float().cpu(), want)
    hd = torch.randn(F_ * 4 * 6, 64)
    want_u = ref.unpatchify(hd, F_, 4, 6, torch.float32)
    got_u = ops.unpatchify(hd.cuda(), F_, 4, 6, dt)
    assert got_u.shape == (1, 16, F_, 8, 12)
    assert torch.equal(got_u.float().cpu(), want_u.to(dt).float())


def test_assemble_tokens_and_cast(ops, ref):
    S, hw, C = 4, 7, 1024
    patch = rnd(S * hw, C, seed=3)
    special = torch.randn(2, 5, C)
    want = ref.assemble_tokens(patch, special, S, hw)
    got = ops.assemble_tokens(bf(patch).cuda(), special.cuda(), S, hw)
    assert torch.equal(got.cpu(), want)
    x = torch.randn(37, 5120)
    assert torch.equal(ops.cast_act(x.cuda()).cpu(), x.to(torch.bfloat16))


# ------------------------------------------------------------------------------- round 3: sampler step, tensor-parallel helpers
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_cfg_euler_step_is_bit_identical_to_the_reference_tensor_ops(ops, dtype):
    """fw_cfg_euler_step against the reference's own tensor ops on the GPU (model_wan21.py:318-321, flow_match.py:52): every
    intermediate rounded to the tensors' dtype -> BIT-identical, host values and device parameters alike."""
    g = torch.Generator().manual_seed(0)
    shape = (1, 16, 5, 30, 52)
    pos, neg, lat = (torch.randn(shape, generator=g).to(dtype).cuda() for _ in range(3))
    for s, ds in ((5.0, -0.004065036773681641), (7.5, -0.09259258210659027), (1.0, 0.0)):
        noise_pred = neg + s * (pos - neg)
        want = lat + noise_pred * torch.tensor(ds, dtype=torch.float32)           # sigma_ - sigma is a 0-dim fp32 CPU tensor
        got = ops.cfg_euler_step(pos, neg, lat, s, ds)
        assert got.dtype == dtype and torch.equal(got, want), (dtype, s, (got.float() - want.float()).abs().max().item())
        params = torch.tensor([s, ds], dtype=torch.float32, device="cuda")
        got2 = ops.cfg_euler_step(pos, neg, lat, 0.0, 0.0, dev_params=params)
        assert torch.equal(got2, want)


def test_row_sumsq_and_qk_prep_with_external_statistics(ops, ref, parity):
    """Tensor-parallel q/k norm: per-rank partial sums of squares + the row statistic supplied to fw_qk_prep_tp reproduce the
    full-width RMSNorm + RoPE of fw_qk_prep on every head slice (same arithmetic up to the summation order of the statistic)."""
    L, H, hd, n = 700, 40, 128, 4
    x = rnd(L, H * hd, seed=11)
    w = rnd(H * hd, seed=12, scale=0.2) + 1.0
    ang = torch.randn(L, hd // 2, generator=torch.Generator().manual_seed(13)).double()
    tab = torch.stack([ang.cos(), ang.sin()], dim=-1).float()
    full = ops.qk_prep(bf(x).cuda().clone(), H, hd, norm="rms_full", norm_w=w.cuda(), eps=1e-6, rope="interleaved",
                       table=tab.cuda(), out_scale=0.1275)
    xs = bf(x).cuda()
    wl = H * hd // n
    parts = [ops.row_sumsq(xs[:, r * wl:(r + 1) * wl]) for r in range(n)]
    want_ss = x.double().pow(2).sum(-1)
    tot = torch.stack(parts).sum(0)
    parity.check("op/row_sumsq", rel_l2(tot, want_ss), 1e-5)
    for r in range(n):
        sl = xs[:, r * wl:(r + 1) * wl].clone()
        ops.qk_prep(sl, H // n, hd, norm="rms_full", norm_w=w[r * wl:(r + 1) * wl].contiguous().cuda(), eps=1e-6, rope="interleaved",
                    table=tab.cuda(), out_scale=0.1275, ext_sumsq=tot, norm_width=H * hd)
        # same inputs, same formula; the statistic differs in its last bits -> at most a bf16 ulp on a few elements
        parity.check(f"op/qk_prep_tp/slice{r}", rel_l2(sl.float(), full[:, r * wl:(r + 1) * wl].float()), 1e-3)
    want = ref.qk_prep(x.clone(), H, hd, norm="rms_full", norm_w=w, eps=1e-6, rope="interleaved", table=tab, out_scale=0.1275)
    parity.check("op/qk_prep_tp/vs_fp32", rel_l2(sl.float(), want[:, (n - 1) * wl:]), 4e-3)


@pytest.mark.parametrize("ydt", [torch.bfloat16, torch.float32])
def test_residual_add_epilogue(ops, ref, ydt, parity):
    rows, C = 515, 1024
    x, y = rnd(rows, C, seed=21), rnd(rows, C, seed=22)
    bias, g1, g0 = rnd(C, seed=23, scale=0.1), rnd(C, seed=24), rnd(C, seed=25, scale=0.1)
    for kw in (dict(), dict(bias=bias), dict(bias=bias, g1=g1), dict(bias=bias, g1=g1, g0=g0)):
        want = ref.residual_add(x.clone(), y, **kw)
        got = ops.residual_add(x.clone().cuda(), y.to(ydt).cuda(), **{k: v.cuda() for k, v in kw.items()})
        parity.check(f"op/residual_add/{ydt}/{len(kw)}", rel_l2(got, want), 1e-6)


# ---------------------------------------------------------------------------------------------- bit-identity across BUILDS
# "bit-identical" claims of the MFMA kernels used to hold per compiler version only (-ffast-math lets
# hipcc re-associate the softmax row sums and the epilogue sums per instantiation).  The MFMA files are now built with
# -fno-associative-math (csrc/build.sh), and the outputs of the production kernels on FIXED inputs are held against committed sha256
# digests: a hipcc upgrade (or a flag change) that moves a bit is named here, as such, instead of tripping hundreds of tight
# parity bounds with no kernel at fault.  Inputs come from the CPU generator (the same bits on every machine).
DIGESTS = "kernel_digests_gfx950.json"


def _digest(t):
    import hashlib
    return hashlib.sha256(t.detach().contiguous().cpu().view(torch.uint8).numpy().tobytes()).hexdigest()


def _digest_cases(ops):
    def attn(H, hd, Lq, Lk, seed):
        q = (rnd(Lq, H * hd, seed=seed) * ops.q_scale(hd)).to(torch.bfloat16).cuda()
        k, v = bf(rnd(Lk, H * hd, seed=seed + 1)).cuda(), bf(rnd(Lk, H * hd, seed=seed + 2)).cuda()
        return ops.attention(q, k, v, H, hd, q_prescaled=True)

    def attn8(H, Lq, Lk, seed, hd=128):
        q = (rnd(Lq, H * hd, seed=seed) * ops.q_scale_fp8(hd)).to(torch.bfloat16).cuda()
        k, v = bf(rnd(Lk, H * hd, seed=seed + 1)).cuda(), bf(rnd(Lk, H * hd, seed=seed + 2)).cuda()
        vt8, lk = ops.prepare_v_fp8(v, H, hd)
        return ops.attention_fp8(ops.cast_fp8(q), ops.cast_fp8(k), vt8, H, hd, lk)

    def gemm(M, N, K, seed, fp8=False, **epi):
        x = bf(rnd(M, K, seed=seed)).cuda()
        lin = ops.pack_linear(rnd(N, K, seed=seed + 1, scale=K ** -0.5), rnd(N, seed=seed + 2, scale=0.1), fp8=fp8)
        if epi.pop("residual", False):
            stream = rnd(M, N, seed=seed + 3).cuda()
            return ops.linear(x, lin, g1=rnd(N, seed=seed + 4).cuda(), res=stream, out_f32=True, out=stream, **epi)
        return ops.linear(x, lin, **epi)
    return {
        "attention_sp_kernel<128,577>/H2_2100x2100": lambda: attn(2, 128, 2100, 2100, 11),
        "attention_sp_kernel<96,577>/H3_1500x1565": lambda: attn(3, 96, 1500, 1565, 21),
        "attention_sp_kernel<64,577>/H4_1565x1565": lambda: attn(4, 64, 1565, 1565, 31),
        "attention_fp8_sp_kernel<default>/H2_2100x2100": lambda: attn8(2, 2100, 2100, 41),
        "attention_fp8_hd64_kernel/H4_1565x1565": lambda: attn8(4, 1565, 1565, 45, hd=64),
        "gemm_bf16_two_slot_kernel/2304x1536x1024_bias_gelu_bf16": lambda: gemm(2304, 1536, 1024, 51, act="gelu_tanh"),
        "gemm_bf16_two_slot_kernel/2304x1024x2048_gate_f32_residual": lambda: gemm(2304, 1024, 2048, 61, residual=True),
        "gemm_fp8_pp_kernel/2304x1024x1024_bias_bf16": lambda: gemm(2304, 1024, 1024, 71, fp8=True),
    }


def test_production_kernel_outputs_match_committed_digests(ops):
    """(Round 6: the three bf16 attention entries are the single-stream kernels with their row sums on the matrix pipe.)
    sha256 of the outputs of the three production attention kernels, the fp8 attention kernel, the two-slot GEMM (two epilogues) and
    the fp8 GEMM on fixed seeds against tests/golden/kernel_digests_gfx950.json.  A mismatch means THE BUILD moved bits (hipcc version,
    compile flags, or a kernel edit that was meant to be bit-neutral): regenerate with FW_WRITE_DIGESTS=1 after checking the parity
    numbers, and say so in the commit.  Digests missing from the file are recorded to gpurun_out/ and reported as a skip."""
    import json
    import os
    from conftest import GOLDEN_DIR, ROOT
    path = os.path.join(GOLDEN_DIR, DIGESTS)
    have = json.load(open(path))["digests"] if os.path.exists(path) else {}
    got = {}
    for name, fn in _digest_cases(ops).items():
        a = fn()
        torch.cuda.synchronize()
        b = fn()                                                 # run-to-run first: a digest of a non-deterministic kernel means nothing
        torch.cuda.synchronize()
        assert torch.equal(a, b), f"{name}: two identical launches returned different bits"
        got[name] = _digest(a)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", DIGESTS), "w") as f:
        json.dump({"note": "sha256 of kernel outputs on fixed CPU-generated inputs (tests/test_hip_ops.py::_digest_cases); "
                           "written by the GPU run, committed as tests/golden/" + DIGESTS, "digests": got}, f, indent=1, sort_keys=True)
    missing = [n for n in got if n not in have]
    moved = [n for n in got if n in have and have[n] != got[n]]
    assert not moved, ("the build moved bits in " + ", ".join(moved) + ": compiler / flags / kernel edit -- if intended, copy "
                       "gpurun_out/" + DIGESTS + " to tests/golden/ and say why in the commit")
    if missing:
        pytest.skip("no committed digest for " + ", ".join(missing) + " (recorded in gpurun_out/" + DIGESTS + ")")


def test_layernorm_split_keeps_the_rounding_remainder(ops, ref):
    """fw_layernorm_mod_split (round 6): hi = the bits fw_layernorm_mod writes, hi + lo = the fp32 LayerNorm to ~16 bits -- the form the
    head's LayerNorm uses (its bf16 rounding alone was a third of the forward's bf16 floor: docs/parity.md)."""
    for rows, C in ((4100, 5120), (33, 1024)):
        g = torch.Generator().manual_seed(41)
        x = torch.randn(rows, C, generator=g).cuda() * 3 + 0.5
        scale, shift = torch.randn(C, generator=g).cuda() * 0.3, torch.randn(C, generator=g).cuda()
        hi, lo = ops.layernorm_split(x, scale=scale, shift=shift, eps=1e-6)
        assert torch.equal(hi, ops.layernorm(x, scale=scale, shift=shift, eps=1e-6))
        want = ref.layernorm(x.cpu(), scale=scale.cpu(), shift=shift.cpu(), eps=1e-6)
        e_hi, e_split = rel_l2(hi.float(), want), rel_l2(hi.float() + lo.float(), want)
        assert e_hi > 1e-3 and e_split < 2e-5, (e_hi, e_split)
