"""The CPU oracle (oracle/fw_oracle.py) must reproduce the golden fixtures, which are outputs of the REAL reference
(oracle/make_golden.py, run in the build container where /root/reference is mounted)."""
import pytest

from conftest import rel_l2
from oracle import fw_oracle


@pytest.mark.parametrize("case_name", ["case_l2", "case_l3", "case_w22", "case_camtok"])
def test_oracle_matches_reference_golden(case_name, request):
    case = request.getfixturevalue(case_name)
    ins = case.inputs
    col = {}
    out = fw_oracle.joint_forward(case.weights, case.cfg, ins["x"], ins["timestep"], ins["context"],
                                  ins["clip_feature"], ins["y"], ins["plucker_fea"], ins["plucker_context_lens"],
                                  uncond=case.uncond, collect=col,
                                  control_camera_latents_input=ins.get("control_camera_latents_input"),
                                  camera_token=ins["camera_token"] if case.golden["meta"].get("camera_token") else None)
    col["noise_pred"] = out
    for k in ("x_after_pcb", "x_final", "tokens_final", "noise_pred"):
        err = rel_l2(col[k], case.golden[k])
        assert err < 5e-6, f"{case.name}:{k} oracle deviates from the reference golden: rel-L2 {err:.3e}"


@pytest.mark.parametrize("case_name", ["case_depth", "case_cfg1"])
def test_oracle_matches_reference_golden_at_depth_and_size(case_name, request):
    """The 8-block depth golden (streams after every block) and BASELINE config 1 (L = 9216; sampled rows): outputs of the
    real reference, reproduced by the oracle."""
    case = request.getfixturevalue(case_name)
    ins, g = case.inputs, case.golden
    col = {}
    if "x_blocks" in g:
        got = {"x": {}, "tok": {}}
        col["per_block"] = lambda kind, i, t: got[kind].__setitem__(i, t.reshape(-1, t.shape[-1]).clone())
    out = fw_oracle.joint_forward(case.weights, case.cfg, ins["x"], ins["timestep"], ins["context"],
                                  ins["clip_feature"], ins["y"], ins["plucker_fea"], ins["plucker_context_lens"], collect=col)
    assert rel_l2(out, g["noise_pred"]) < 5e-6
    rd, ra = g["rows_dit"], g["rows_agg"]
    L2 = col["tokens_final"].shape[0]
    assert rel_l2(col["x_after_pcb"][rd], g["x_after_pcb"]) < 5e-6
    assert rel_l2(col["x_final"][rd], g["x_final"]) < 5e-6
    assert rel_l2(col["tokens_final"].reshape(L2, -1)[ra], g["tokens_final"]) < 5e-6
    if "x_blocks" in g:
        for b in range(case.cfg.num_layers):
            assert rel_l2(got["x"][b][rd], g["x_blocks"][b]) < 5e-6, b
        for j in range(case.cfg.n_irg):
            assert rel_l2(got["tok"][j][ra], g["tok_blocks"][j]) < 5e-6, j


def test_heads_oracle_matches_reference_golden(heads_case):
    """oracle/fw_heads_oracle.py reproduces the prediction dict of the real VGGT._head_predction (vggt.py:134-154)."""
    from conftest import PRED_KEYS
    from oracle import fw_heads_oracle
    c = heads_case
    pred = fw_heads_oracle.head_prediction(c.weights, c.output_list, c.hc, c.S, c.ph, c.pw)
    for k in PRED_KEYS:
        assert pred[k].shape == c.golden[k].shape, (k, pred[k].shape, c.golden[k].shape)
        err = rel_l2(pred[k], c.golden[k])
        assert err < 5e-6, f"{c.name}:{k} heads oracle deviates from the reference golden: rel-L2 {err:.3e}"


@pytest.mark.parametrize("S,ph,pw", [(4, 3, 4), (2, 1, 1), (3, 2, 7)])
def test_heads_oracle_matches_live_reference(S, ph, pw):
    """In the build container (reference mounted): more grids (a 1x1 token grid, an odd width), straight against the
    reference modules, which decode time frame by frame through the convolution cache -- the oracle's whole-sequence causal
    convolutions and the host logic of fantasy_world_amd.heads (on the torch ops) must both agree.  (A single latent frame is
    not a reference case: CameraHead's time up-sampler fails on an empty sequence, camera_head.py:93.)"""
    from oracle import ref_locate
    if not ref_locate.available():
        pytest.skip("reference not mounted / staged on this machine")
    from conftest import PRED_KEYS
    from fantasy_world_amd import config as fwc, synth, heads as fw_heads
    from oracle import fw_heads_oracle, ref_harness, ref_ops
    hc = fwc.HeadsConfig.small()
    W = synth.make_heads_weights(hc, seed=5)
    ol = synth.make_output_list(hc, S, ph, pw, seed=9)
    ref = ref_harness.run_reference_heads(ref_harness.build_reference_heads(hc, W), ol, S, ph, pw, max(hc.layer_idx) + 1)
    pred = fw_heads_oracle.head_prediction(W, ol, hc, S, ph, pw)
    got = fw_heads.GeometryHeads(hc, W.__getitem__, ref_ops.TorchRefOps()).predict({k: v[None] for k, v in ol.items()}, S, ph, pw)
    for k in PRED_KEYS:
        assert rel_l2(pred[k], ref[k]) < 5e-6, k
        assert rel_l2(got[k], ref[k]) < 2e-5, k


def test_oracle_chain_matches_reference_prediction_golden(case_pred):
    """joint_forward's output_list fed to the heads oracle == the reference's joint_forward(return_prediction=True)."""
    from conftest import PRED_KEYS
    from oracle import fw_heads_oracle
    c, ins = case_pred, case_pred.inputs
    col = {"output_list": {}}
    out = fw_oracle.joint_forward(c.weights, c.cfg, ins["x"], ins["timestep"], ins["context"], ins["clip_feature"], ins["y"],
                                  ins["plucker_fea"], ins["plucker_context_lens"], collect=col)
    assert rel_l2(out, c.golden["noise_pred"]) < 5e-6
    f, h2, w2 = c.grid
    pred = fw_heads_oracle.head_prediction(c.weights, col["output_list"], c.hc, f, h2 // 2, w2 // 2)
    for k in PRED_KEYS:
        assert rel_l2(pred[k], c.golden[k]) < 1e-5, k


def test_heads_oracle_full_width_golden(heads_case_full):
    """The reference's real head widths (dim 2048, layers 23/17/11/7, DPT 256/512/1024/1024) on a synthetic 24-entry
    output_list: oracle and host logic (torch ops) against the reference's prediction."""
    from conftest import PRED_KEYS
    from fantasy_world_amd import heads as fw_heads
    from oracle import fw_heads_oracle, ref_ops
    c = heads_case_full
    pred = fw_heads_oracle.head_prediction(c.weights, c.output_list, c.hc, c.S, c.ph, c.pw)
    got = fw_heads.GeometryHeads(c.hc, c.weights.__getitem__, ref_ops.TorchRefOps()).predict(
        {k: v[None] for k, v in c.output_list.items()}, c.S, c.ph, c.pw)
    for k in PRED_KEYS:
        assert rel_l2(pred[k], c.golden[k]) < 5e-6, k
        assert rel_l2(got[k], c.golden[k]) < 2e-5, k


def test_pose_encoder_oracle_and_host_logic_match_reference_golden(pose_case):
    """CameraPoseEncoder (A21): oracle/fw_pose_oracle.py and fantasy_world_amd.pose_encoder (on the torch ops) reproduce the
    plucker_fea of the real module (pose_adaptor_ac3d.py:83-118) at its real widths."""
    from fantasy_world_amd.pose_encoder import PoseEncoder
    from oracle import fw_pose_oracle, ref_ops
    c = pose_case
    want = c.golden["plucker_fea"]
    got = fw_pose_oracle.camera_pose_encoder(c.weights, c.plucker)
    assert got.shape == want.shape and rel_l2(got, want) < 5e-6
    host = PoseEncoder(c.weights.__getitem__, ref_ops.TorchRefOps()).encode(c.plucker)
    assert host.shape == want.shape and rel_l2(host, want) < 5e-6
    # with bf16 rounding where the HIP path stores bf16: the yardstick for the GPU tolerance
    emu = PoseEncoder(c.weights.__getitem__, ref_ops.TorchRefOps(emulate_bf16=True)).encode(c.plucker)
    assert rel_l2(emu, want) < 1.5e-2


def test_vae_decoder_oracle_matches_reference_golden(vae_case):
    """oracle/fw_vae_oracle.py (whole-sequence causal convolutions) reproduces VideoVAE_.decode, which decodes frame by frame
    through its convolution cache (wan_video_vae.py:552-575)."""
    from oracle import fw_vae_oracle
    got = fw_vae_oracle.vae_decode(vae_case.weights, vae_case.latents)
    want = vae_case.golden["video"]
    assert got.shape == want.shape and rel_l2(got, want) < 1e-5


def test_vae_decoder_host_logic_matches_reference_golden(vae_case):
    """fantasy_world_amd.vae_decoder on the torch ops: weight packing (16 -> 64 and 96 -> 128 channel padding, latent
    un-normalisation folded into conv2), up-sampling folded into the gather, GEMM-based single-head attention, chunking."""
    from fantasy_world_amd.vae_decoder import VaeDecoder
    from oracle import ref_ops
    c = vae_case
    want = c.golden["video"]
    got = VaeDecoder(c.weights.__getitem__, ref_ops.TorchRefOps()).decode(c.latents)
    assert got.shape == want.shape and rel_l2(got, want) < 1e-5
    # scalar scale (VideoVAE_.decode's non-tensor branch): same as the per-channel tensors it stands for
    dec = VaeDecoder(c.weights.__getitem__, ref_ops.TorchRefOps())
    import torch
    a = dec.decode(c.latents, [0.25, 2.0])
    b = dec.decode(c.latents, [torch.full((16,), 0.25), torch.full((16,), 2.0)])
    assert rel_l2(a, b) < 1e-6
    tiny = VaeDecoder(c.weights.__getitem__, ref_ops.TorchRefOps(), max_col_bytes=1).decode(c.latents)
    assert rel_l2(tiny, got) < 1e-5
    emu = VaeDecoder(c.weights.__getitem__, ref_ops.TorchRefOps(emulate_bf16=True)).decode(c.latents)
    assert rel_l2(emu, want) < 2e-2              # the yardstick behind the GPU tolerance


def test_staged_reference_bundle_is_the_unmodified_reference():
    """oracle/_ref/reference_py.tgz (what the GPU box runs as the checker, oracle/stage_ref.sh) holds the reference's Python files
    BYTE FOR BYTE: every member is compared with /root/reference.  Runs only where both exist (the build container)."""
    import hashlib
    import os
    import tarfile
    from oracle import ref_locate
    root = "/root/reference"
    if not (os.path.isfile(ref_locate.BUNDLE) and os.path.isdir(os.path.join(root, "FantasyWorld"))):
        pytest.skip("needs both the staged bundle and /root/reference")
    n = 0
    with tarfile.open(ref_locate.BUNDLE, "r:gz") as tar:
        for m in tar.getmembers():
            if not m.isfile():
                continue
            with open(os.path.join(root, m.name), "rb") as f:
                want = hashlib.sha256(f.read()).hexdigest()
            got = hashlib.sha256(tar.extractfile(m).read()).hexdigest()
            assert got == want, m.name
            n += 1
    assert n > 90, n           # FantasyWorld/ (103 files) + the two inference scripts
