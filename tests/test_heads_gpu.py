"""-m gpu: the geometry heads through libfw_mi355x.so.

  * every heads kernel against the torch statement of the same op (oracle/ref_ops.py): pure gathers are bit-exact,
    kernels that write bf16 within one rounding (4e-3 rel-L2), fp32 outputs within 1e-5;
  * the whole prediction dict (pose_enc, depth, world_points and confidences) of fantasy_world_amd.heads on HipOps against
    the golden output of the REAL reference (VGGT._head_predction, fp32 CPU).  Tolerance 1.5e-2 rel-L2: the path stores
    bf16 between ~25 chained convolutions like the reference's autocast; the same host code on the torch ops with bf16
    rounding emulated measures 1e-3 .. 7.4e-3 (world_points goes through sign*expm1, which amplifies);
  * properties at sizes the oracle would take minutes for: frame chunking invisible, causality of the temporal decode.
"""
import pytest
import torch

from conftest import PRED_KEYS, rel_l2

pytestmark = pytest.mark.gpu

HEADS_TOL = 1.5e-2


@pytest.fixture(scope="module")
def ops():
    from fantasy_world_amd.hip_ops import HipOps
    return HipOps("cuda:0")


@pytest.fixture(scope="module")
def ref():
    from oracle.ref_ops import TorchRefOps
    return TorchRefOps()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).float()


def dev(t):
    return t.to(torch.bfloat16).cuda()


IM2COL_CASES = [  # T, H, W, C, kt, kh, kw, sh, sw, t0, nt, relu_in
    (3, 5, 7, 64, 1, 3, 3, 1, 1, 0, None, False),
    (2, 6, 4, 128, 1, 3, 3, 2, 2, 0, None, False),
    (2, 5, 5, 64, 1, 3, 3, 2, 2, 0, None, True),
    (5, 4, 6, 64, 3, 1, 1, 1, 1, 0, None, False),
    (6, 4, 5, 64, 3, 3, 3, 1, 1, 0, None, False),
    (7, 3, 4, 72, 3, 3, 3, 1, 1, 2, 3, True),
    (9, 2, 3, 8, 3, 3, 3, 1, 1, 8, 1, False),
]


@pytest.mark.parametrize("T,H,W,C,kt,kh,kw,sh,sw,t0,nt,relu_in", IM2COL_CASES)
def test_im2col_is_an_exact_gather(ops, ref, T, H, W, C, kt, kh, kw, sh, sw, t0, nt, relu_in):
    x = rnd(T * H * W, C, seed=1)
    want = ref.im2col(x, T, H, W, kt, kh, kw, sh, sw, t0, nt, relu_in)
    got = ops.im2col(dev(x), T, H, W, kt, kh, kw, sh, sw, t0, nt, relu_in)
    assert got.shape == want.shape
    assert torch.equal(got.float().cpu(), want)


def test_im2col_then_gemm_is_the_convolution(ops, parity, request):
    """Tap-major gather + weight [N][kt][kh][kw][C] == F.conv3d with causal time padding (vae_modified.py:17-36)."""
    import torch.nn.functional as F
    T, H, W, C, N = 5, 6, 7, 64, 128
    x, w, b = rnd(T * H * W, C, seed=2), rnd(N, C, 3, 3, 3, seed=3, scale=(27 * C) ** -0.5), rnd(N, seed=4, scale=0.1)
    vol = x.view(T, H, W, C).permute(3, 0, 1, 2)[None]
    want = F.conv3d(F.pad(vol, (1, 1, 1, 1, 2, 0)), w, b)[0].permute(1, 2, 3, 0).reshape(T * H * W, N)
    lin = ops.pack_linear(w.permute(0, 2, 3, 4, 1).reshape(N, 27 * C), b)
    got = ops.linear(ops.im2col(dev(x), T, H, W, 3, 3, 3), lin, out_f32=True)
    parity.check(f"op/{request.node.name}/0", rel_l2(got, want), 1e-3)


CONV_GEMM_CASES = [  # T, H, W, C, N, kt, kh, kw, sh, sw, t0, nt, up, tile
    (3, 10, 12, 64, 64, 3, 3, 3, 1, 1, 0, None, 1, 0),       # CausalConv3d 3x3x3, 128x128 kernel
    (2, 21, 17, 128, 256, 1, 3, 3, 1, 1, 0, None, 1, 0),     # Conv2d 3x3, two k-slabs per tap
    (1, 37, 37, 64, 128, 1, 3, 3, 2, 2, 0, None, 1, 0),      # stride 2 (dpt_head resize_layers[3])
    (3, 9, 11, 128, 128, 1, 3, 3, 1, 1, 0, None, 2, 0),      # on the nearest x2 up-sampled map (VAE Resample)
    (6, 8, 8, 64, 128, 3, 1, 1, 1, 1, 0, None, 1, 0),        # time_conv (3,1,1)
    (7, 6, 5, 64, 64, 3, 3, 3, 1, 1, 2, 3, 1, 0),            # frame window t0 = 2, nt = 3
    (2, 63, 80, 64, 256, 1, 3, 3, 1, 1, 0, None, 1, 256),    # the 256x256 ping-pong kernel (forced), ragged last band
    (3, 48, 40, 128, 512, 3, 3, 3, 1, 1, 1, 2, 1, 256),      # 256x256 kernel, 27 taps x 2 slabs, two column tiles
    (2, 30, 36, 64, 256, 1, 3, 3, 1, 1, 0, None, 2, 256),    # 256x256 kernel on the up-sampled map
]


@pytest.mark.parametrize("T,H,W,C,N,kt,kh,kw,sh,sw,t0,nt,up,tile", CONV_GEMM_CASES)
@pytest.mark.parametrize("epi", ["plain", "relu_res_f32"])
def test_conv_gemm_equals_gather_plus_gemm(ops, T, H, W, C, N, kt, kh, kw, sh, sw, t0, nt, up, tile, epi):
    """fw_conv_gemm_bf16 (implicit GEMM: the tap gather is the A tile's DMA source address) against fw_im2col + fw_gemm_bf16: the
    same kernel body, the same k-order per output element -> BIT-IDENTICAL, on both tile kernels, with strides, the up-sampled
    map, frame windows and the fused epilogue."""
    x = dev(rnd(T * H * W, C, seed=11))
    K = kt * kh * kw * C
    lin = ops.pack_linear(rnd(N, K, seed=12, scale=K ** -0.5), rnd(N, seed=13, scale=0.1))
    cols = ops.im2col(x, T, H, W, kt, kh, kw, sh, sw, t0, nt, False, up=up)
    kw_epi = {}
    if epi != "plain":
        kw_epi = dict(act="relu", out_f32=True, res=torch.randn(cols.shape[0], N, generator=torch.Generator().manual_seed(14)).cuda())
    ops.set_option("gemm_tile", tile)
    try:
        want = ops.linear(cols, lin, **kw_epi)
        got = ops.conv_gemm(x, T, H, W, lin, kt, kh, kw, sh=sh, sw=sw, t0=t0, nt=nt, up=up, **kw_epi)
        torch.cuda.synchronize()
    finally:
        ops.set_option("gemm_tile", 0)
    assert got.shape == want.shape and got.dtype == want.dtype
    assert torch.equal(got, want)


def test_conv_gemm_is_the_convolution(ops, parity, request):
    """... and against F.conv3d itself (fp32, causal time padding, vae_modified.py:17-36)."""
    import torch.nn.functional as F
    T, H, W, C, N = 5, 6, 7, 64, 128
    x, w, b = rnd(T * H * W, C, seed=2), rnd(N, C, 3, 3, 3, seed=3, scale=(27 * C) ** -0.5), rnd(N, seed=4, scale=0.1)
    vol = x.view(T, H, W, C).permute(3, 0, 1, 2)[None]
    want = F.conv3d(F.pad(vol, (1, 1, 1, 1, 2, 0)), w, b)[0].permute(1, 2, 3, 0).reshape(T * H * W, N)
    lin = ops.pack_linear(w.permute(0, 2, 3, 4, 1).reshape(N, 27 * C), b)
    got = ops.conv_gemm(dev(x), T, H, W, lin, 3, 3, 3, out_f32=True)
    parity.check(f"op/{request.node.name}/0", rel_l2(got, want), 1e-3)


def test_conv_gemm_rejects_what_it_cannot_pack(ops):
    lin = ops.pack_linear(rnd(64, 640, seed=1), None)
    with pytest.raises(AssertionError):
        ops.conv_gemm(dev(rnd(16, 72, seed=2)), 1, 4, 4, lin, 1, 3, 3)          # C % 64 != 0: the gather path serves these


@pytest.mark.parametrize("N,h,w,H,W,C", [(2, 4, 6, 8, 12, 64), (1, 15, 26, 30, 52, 72), (3, 5, 3, 20, 12, 64), (1, 7, 7, 7, 7, 8),
                                         (2, 1, 5, 4, 9, 16)])
def test_resize_bilinear_align_corners(ops, ref, N, h, w, H, W, C, parity, request):
    x = rnd(N * h * w, C, seed=5)
    want = ref.resize_bilinear(x, N, h, w, H, W)
    got = ops.resize_bilinear(dev(x), N, h, w, H, W)
    parity.check(f"op/{request.node.name}/0", rel_l2(got.float(), want), 4e-3)


@pytest.mark.parametrize("rows,C,c_true", [(301, 128, 96), (77, 64, 64), (1030, 256, 192), (50, 384, 384), (9, 1024, 1024)])
def test_chan_rmsnorm_silu(ops, ref, rows, C, c_true, parity, request):
    """Both kernels: several rows per wave for C = 64 / 128 / 256, one wave per row otherwise; ragged row counts."""
    x = rnd(rows, C, seed=6, scale=3.0)
    x[:, c_true:] = 0
    g = torch.zeros(C)
    g[:c_true] = 1 + 0.1 * rnd(c_true, seed=7)
    want = ref.chan_rmsnorm_silu(x, g, c_true)
    got = ops.chan_rmsnorm_silu(dev(x), g.cuda(), c_true)
    parity.check(f"op/{request.node.name}/0", rel_l2(got.float(), want), 4e-3)
    assert (got[:, c_true:] == 0).all()


def test_depth_to_space_unfold_time_add_table_add_act(ops, ref, parity, request):
    N, h, w, k, C = 2, 3, 5, 4, 64
    y = rnd(N * h * w, k * k * C, seed=8)
    assert torch.equal(ops.depth_to_space(dev(y), N, h, w, k, C).float().cpu(), ref.depth_to_space(y, N, h, w, k, C))
    n, hw = 3, 10
    y = rnd(n * hw, 2 * C, seed=9)
    assert torch.equal(ops.unfold_time2(dev(y), n, hw, C).float().cpu(), ref.unfold_time2(y, n, hw, C))
    x, tab = rnd(4 * hw, C, seed=10), rnd(hw, C, seed=11, scale=0.1)
    want = ref.add_table(x.clone(), tab)
    got = ops.add_table(dev(x), tab.cuda())
    parity.check(f"op/{request.node.name}/0", rel_l2(got.float(), want), 4e-3)
    a, b = rnd(77, C, seed=12), rnd(77, C, seed=13)
    for relu in (False, True):
        parity.check(f"op/{request.node.name}/1", rel_l2(ops.add_act(dev(a), dev(b), relu=relu).float(), ref.add_act(a, b, relu=relu)), 4e-3)
    assert torch.equal(ops.add_act(dev(a), None, relu=True).float().cpu(), torch.relu(a))


def test_adaln_rows_and_head_activation(ops, ref, parity, request):
    rows, C = 81, 256
    x, mod = rnd(rows, C, seed=14, scale=2.0), rnd(rows, 3 * C, seed=15, scale=0.5)
    parity.check(f"op/{request.node.name}/0", rel_l2(ops.adaln_rows(x.cuda(), mod.cuda()), ref.adaln_rows(x, mod)), 1e-5)
    y = rnd(1000, 4, seed=16, scale=2.0)
    for mode in ("exp", "inv_log"):
        p, c = ops.head_activation(y.cuda(), mode)
        pw, cw = ref.head_activation(y, mode)
        assert rel_l2(p, pw) < 1e-5 and rel_l2(c, cw) < 1e-5
    y = rnd(81, 9, seed=17)
    assert torch.equal(ops.head_activation(y.cuda(), "pose").cpu(), ref.head_activation(y, "pose"))


def _predict(case, ops, **kw):
    from fantasy_world_amd import heads as fw_heads
    gh = fw_heads.GeometryHeads(case.hc, case.weights.__getitem__, ops, **kw)
    ol = {k: v[None].cuda() for k, v in case.output_list.items()}
    out = gh.predict(ol, case.S, case.ph, case.pw)
    torch.cuda.synchronize()
    return out


def test_heads_prediction_matches_reference_golden(heads_case, ops):
    pred = _predict(heads_case, ops)
    for k in PRED_KEYS:
        assert pred[k].shape == heads_case.golden[k].shape, (k, pred[k].shape)
        assert torch.isfinite(pred[k]).all(), k
        err = rel_l2(pred[k], heads_case.golden[k])
        assert err < HEADS_TOL, f"{heads_case.name}:{k} rel-L2 {err:.3e}"


def test_heads_frame_chunking_is_invisible_on_gpu(heads_case, ops):
    a = _predict(heads_case, ops)
    b = _predict(heads_case, ops, max_col_bytes=1, frames_chunk=3)
    for k in PRED_KEYS:
        assert torch.equal(a[k], b[k]), k


def test_heads_temporal_decode_is_causal(ops):
    """Changing the tokens of the LAST latent frame leaves every prediction frame before its first decoded frame unchanged
    (CausalConv3d, vae_modified.py:17-36), at a grid larger than the goldens."""
    from fantasy_world_amd import config as fwc, synth, heads as fw_heads
    hc = fwc.HeadsConfig.small()
    W = synth.make_heads_weights(hc, seed=2)
    S, ph, pw = 5, 6, 9
    ol = synth.make_output_list(hc, S, ph, pw, seed=4)
    gh = fw_heads.GeometryHeads(hc, W.__getitem__, ops)
    a = gh.predict({k: v[None].cuda() for k, v in ol.items()}, S, ph, pw)
    ol2 = {k: v.clone() for k, v in ol.items()}
    for v in ol2.values():
        v[S - 1, 5:] += torch.randn(v[S - 1, 5:].shape, generator=torch.Generator().manual_seed(1))   # last latent frame
    b = gh.predict({k: v[None].cuda() for k, v in ol2.items()}, S, ph, pw)
    T = (S - 1) * 4 + 1
    first_touched = (S - 2) * 4 + 1              # latent frame S-1 decodes to prediction frames (S-2)*4+1 .. T-1
    for k in ("depth", "depth_conf", "world_points", "world_points_conf"):
        assert a[k].shape[1] == T
        assert torch.equal(a[k][:, :first_touched], b[k][:, :first_touched]), k
        assert not torch.equal(a[k][:, first_touched:], b[k][:, first_touched:]), k


def test_heads_full_width_matches_reference_golden(heads_case_full, ops):
    """Real widths through the kernels (K up to 27 648 in the 3x3x3 convolutions at 1024 channels, 16-head hd-128 camera
    trunk, 4x4 / 2x2 transposed convolutions): the reference's own prediction on a 2x3 token grid."""
    pred = _predict(heads_case_full, ops)
    errs = {k: rel_l2(pred[k], heads_case_full.golden[k]) for k in PRED_KEYS}
    print({k: f"{v:.2e}" for k, v in errs.items()})
    for k in PRED_KEYS:
        assert pred[k].shape == heads_case_full.golden[k].shape and torch.isfinite(pred[k]).all(), k
        assert errs[k] < HEADS_TOL, (k, errs[k])
