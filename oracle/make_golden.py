"""TEST INFRASTRUCTURE ONLY -- generate tests/golden/*.pt by running the REAL reference (imported from
/root/reference through oracle/ref_harness.py) on deterministic synthetic weights and inputs, and check that
oracle/fw_oracle.py reproduces it.  Runs only in the build container:

    python oracle/make_golden.py            # writes tests/golden/wan21_*.pt, prints oracle-vs-reference errors

The fixtures hold only small tensors (outputs and a few intermediate activations); weights and inputs are
regenerated from seeds by fantasy_world_amd.synth on whatever machine runs the tests.
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from fantasy_world_amd import config as fwc, synth   # noqa: E402
from oracle import fw_oracle, fw_heads_oracle, fw_pose_oracle, fw_vae_oracle, ref_harness   # noqa: E402

CASES = {
    # name: (cfg kwargs, (f, h2, w2), timestep, text_len, uncond)
    "wan21_l2_f3_8x8": (dict(num_layers=2, start_index=1), (3, 8, 8), 500.0, 512, False),
    "wan21_l3_f2_12x8": (dict(num_layers=3, start_index=1), (2, 12, 8), 937.5, 512, False),
    # Wan2.2-Fun-A14B-Control-Camera flavour (model_wan22.py): control adapter in patchify, text-only context, no per-block adapter
    "wan22_l2_f2_8x12": (dict(num_layers=2, start_index=1), (2, 8, 12), 968.75, 512, False),
    # camera_token given: VGGT's per-frame camera token comes from CamTokenProjector instead of the learned parameter
    # (aggregator.py:265-266) -- not used by the inference scripts, part of the joint_forward signature
    "wan21_camtok_l2_f3_8x8": (dict(num_layers=2, start_index=1), (3, 8, 8), 250.0, 512, False),
}


# geometry heads, reduced widths (HeadsConfig.small()): name -> (S, ph, pw)
HEAD_CASES = {
    "heads_small_s3_4x6": (3, 4, 6),
    "heads_small_s2_5x3": (2, 5, 3),
    # the reference's real widths (dim 2048, 16-head trunk of depth 4, DPT 256 / 512 / 1024 / 1024, layers 23/17/11/7, patch 16)
    # on a synthetic 24-entry output_list: 623 M parameters, tiny grid
    "heads_full_s2_2x3": (2, 2, 3),
}


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()


def run_reference(cfg, W, ins, uncond=False, camera_token=None):
    model = (ref_harness.build_reference_wan22 if cfg.control_adapter else ref_harness.build_reference_wan21)(cfg, weights=W)
    cap = {}

    def hook_pcb(m, a, out):
        cap["x_after_pcb"] = out[0].detach().clone()

    def hook_irg(m, a, out):
        cap["x_final"] = out[0][0].detach().clone()
        cap["tokens_final"] = out[1][0].detach().clone()

    model.pipe.dit.blocks[cfg.start_index - 1].register_forward_hook(hook_pcb)
    model.IRGBlock[len(cfg.cross_attention_list) - 1].register_forward_hook(hook_irg)
    with torch.no_grad():
        if cfg.control_adapter:      # model_wan22.py:231-242 signature
            out, pred = model.joint_forward(
                ins["x"], timestep=ins["timestep"], context=ins["context"], y=ins["y"], use_gradient_checkpointing=False,
                camera_token=None, control_camera_latents_input=ins["control_camera_latents_input"], uncond=uncond,
                return_prediction=False)
        else:
            out, pred = model.joint_forward(
                ins["x"], timestep=ins["timestep"], context=ins["context"], clip_feature=ins["clip_feature"],
                y=ins["y"], use_gradient_checkpointing=False, camera_token=camera_token, plucker_fea=ins["plucker_fea"],
                plucker_context_lens=ins["plucker_context_lens"], uncond=uncond, return_prediction=False)
    assert pred is None
    cap["noise_pred"] = out.detach().clone()
    return cap, model


def main():
    torch.manual_seed(0)
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    only = sys.argv[1:]
    for name, (ckw, (f, h2, w2), ts, tl, uncond) in CASES.items():
        if only and name not in only:
            continue
        cfg = fwc.plumbing22(**ckw) if name.startswith("wan22") else fwc.plumbing(**ckw)
        t0 = time.time()
        W = synth.make_weights(cfg)
        ins = synth.make_inputs(cfg, f, h2, w2, seed=1, timestep=ts, text_len=tl)
        print(f"[{name}] weights {sum(v.numel() for v in W.values())/1e9:.2f} B params in {time.time()-t0:.1f}s")
        t0 = time.time()
        camtok = ins["camera_token"] if "camtok" in name else None
        ref, model = run_reference(cfg, W, ins, uncond, camera_token=camtok)
        print(f"[{name}] reference forward {time.time()-t0:.1f}s; unused synth names: {model._fw_unused[:5]} "
              f"(n={len(model._fw_unused)}); hot-path names missing from synth: "
              f"{[m for m in model._fw_missing if not _cold(m)][:8]}")
        assert not model._fw_unused, "synth produced names the reference does not have"
        assert not [m for m in model._fw_missing if not _cold(m)], "synth misses hot-path parameters"
        del model
        col = {}
        t0 = time.time()
        orc = fw_oracle.joint_forward(W, cfg, ins["x"], ins["timestep"], ins["context"], ins["clip_feature"], ins["y"],
                                      ins["plucker_fea"], ins["plucker_context_lens"], uncond=uncond, collect=col,
                                      control_camera_latents_input=ins.get("control_camera_latents_input"), camera_token=camtok)
        print(f"[{name}] oracle forward {time.time()-t0:.1f}s")
        col["noise_pred"] = orc
        for k in ("x_after_pcb", "x_final", "tokens_final", "noise_pred"):
            print(f"   oracle vs reference  {k:14s} rel-L2 = {rel(col[k], ref[k]):.3e}   |ref| = {ref[k].norm():.3f}")
        # a second timestep draw through the negative-prompt context (the CFG pair of one denoise step)
        golden = {k: v.to(torch.float32).contiguous() for k, v in ref.items()}
        golden["meta"] = dict(cfg=ckw, grid=(f, h2, w2), timestep=ts, text_len=tl, uncond=uncond, seed_weights=0,
                              seed_inputs=1, torch=torch.__version__, flavour="wan22" if cfg.control_adapter else "wan21",
                              camera_token=camtok is not None)
        path = os.path.join(ROOT, "tests", "golden", name + ".pt")
        torch.save(golden, path)
        print(f"[{name}] wrote {path} ({os.path.getsize(path)/1e6:.2f} MB)")


def main_pred(only):
    """End to end: the REAL reference's joint_forward(return_prediction=True) on a reduced-depth fusion model whose VGGT
    carries narrow geometry heads (HeadsConfig.e2e_small()) -> noise_pred and the prediction dict."""
    name = "wan21_pred_l3_f2_8x12"
    if only and name not in only:
        return
    ckw, (f, h2, w2), ts, tl = dict(num_layers=3, start_index=1), (2, 8, 12), 125.0, 512
    cfg, hc = fwc.plumbing(**ckw), fwc.HeadsConfig.e2e_small()
    W = synth.make_weights(cfg)
    W.update(synth.make_heads_weights(hc))
    ins = synth.make_inputs(cfg, f, h2, w2, seed=1, timestep=ts, text_len=tl)
    model = ref_harness.build_reference_wan21(cfg, weights=W, heads_cfg=hc)
    assert not model._fw_unused, model._fw_unused[:5]
    t0 = time.time()
    with torch.no_grad():
        out, pred = model.joint_forward(
            ins["x"], timestep=ins["timestep"], context=ins["context"], clip_feature=ins["clip_feature"], y=ins["y"],
            use_gradient_checkpointing=False, camera_token=None, plucker_fea=ins["plucker_fea"],
            plucker_context_lens=ins["plucker_context_lens"], uncond=False, return_prediction=True)
    print(f"[{name}] reference joint_forward(return_prediction=True) {time.time()-t0:.1f}s")
    col = {"output_list": {}}
    orc = fw_oracle.joint_forward(W, cfg, ins["x"], ins["timestep"], ins["context"], ins["clip_feature"], ins["y"],
                                  ins["plucker_fea"], ins["plucker_context_lens"], collect=col)
    opred = fw_heads_oracle.head_prediction(W, col["output_list"], hc, f, h2 // 2, w2 // 2)
    print(f"   oracle vs reference  noise_pred         rel-L2 = {rel(orc, out):.3e}")
    for k in pred:
        print(f"   oracle vs reference  {k:18s} rel-L2 = {rel(opred[k], pred[k]):.3e}   shape {tuple(pred[k].shape)}")
    golden = {k: v.to(torch.float32).contiguous() for k, v in pred.items()}
    golden["noise_pred"] = out.to(torch.float32).contiguous()
    golden["meta"] = dict(cfg=ckw, grid=(f, h2, w2), timestep=ts, text_len=tl, uncond=False, seed_weights=0, seed_inputs=1,
                          torch=torch.__version__, flavour="wan21", heads="e2e_small")
    path = os.path.join(ROOT, "tests", "golden", name + ".pt")
    torch.save(golden, path)
    print(f"[{name}] wrote {path} ({os.path.getsize(path)/1e6:.2f} MB)")


# CameraPoseEncoder at the reference's real widths (6 -> 384 -> 768 -> 5120 -> 2560 -> 2048): name -> (frames, H, W)
POSE_CASES = {
    "pose_full_f9_32x48": (9, 32, 48),
    "pose_full_f13_48x16": (13, 48, 16),
}


def main_pose(only):
    ref_harness.install_stubs()
    from FantasyWorld.diffsynth_wan21.models.pose_adaptor_ac3d import CameraPoseEncoder
    for name, (f, H, Wd) in POSE_CASES.items():
        if only and name not in only:
            continue
        W = synth.make_pose_encoder_weights()
        enc = CameraPoseEncoder(context_dim=2048, dim=5120, patch_size=[1, 2, 2], in_channels=6, downscale_coef=8,
                                pose_inject_method="adaln").eval()
        enc.load_state_dict({k.replace("camera_condition.pose_encoder.", ""): v for k, v in W.items()}, strict=True)
        pl = synth.make_plucker(f, H, Wd)
        with torch.no_grad():
            ref = enc(pl)
        orc = fw_pose_oracle.camera_pose_encoder(W, pl)
        print(f"[{name}] oracle vs reference plucker_fea rel-L2 = {rel(orc, ref):.3e}   shape {tuple(ref.shape)}")
        path = os.path.join(ROOT, "tests", "golden", name + ".pt")
        torch.save({"plucker_fea": ref.float().contiguous(),
                    "meta": dict(grid=(f, H, Wd), seed_weights=0, seed_plucker=5, torch=torch.__version__)}, path)
        print(f"[{name}] wrote {path} ({os.path.getsize(path)/1e6:.2f} MB)")


# Wan video VAE decoder at its real widths (73 M parameters): name -> latent grid (T, h, w)
VAE_CASES = {
    "vae_full_t3_4x6": (3, 4, 6),
    "vae_full_t2_3x5": (2, 3, 5),
}


def main_vae(only):
    ref_harness.install_stubs()
    from FantasyWorld.diffsynth_wan21.models.wan_video_vae import VideoVAE_
    for name, (T, h, w) in VAE_CASES.items():
        if only and name not in only:
            continue
        W = synth.make_vae_decoder_weights()
        m = VideoVAE_(z_dim=16).eval()
        sd = m.state_dict()
        sd.update(W)
        m.load_state_dict(sd)
        z = synth.make_latents(T, h, w)
        scale = [torch.tensor(fw_vae_oracle.VAE_MEAN), 1.0 / torch.tensor(fw_vae_oracle.VAE_STD)]
        with torch.no_grad():
            ref = m.decode(z, scale)
        orc = fw_vae_oracle.vae_decode(W, z)
        print(f"[{name}] oracle vs reference video rel-L2 = {rel(orc, ref):.3e}   shape {tuple(ref.shape)}")
        path = os.path.join(ROOT, "tests", "golden", name + ".pt")
        torch.save({"video": ref.float().contiguous(),
                    "meta": dict(grid=(T, h, w), seed_weights=0, seed_latents=7, torch=torch.__version__)}, path)
        print(f"[{name}] wrote {path} ({os.path.getsize(path)/1e6:.2f} MB)")


def main_heads(only):
    for name, (S, ph, pw) in HEAD_CASES.items():
        if only and name not in only:
            continue
        hc = fwc.HeadsConfig() if "_full_" in name else fwc.HeadsConfig.small()
        W = synth.make_heads_weights(hc)
        ol = synth.make_output_list(hc, S, ph, pw)
        vggt = ref_harness.build_reference_heads(hc, W)
        t0 = time.time()
        ref = ref_harness.run_reference_heads(vggt, ol, S, ph, pw, max(hc.layer_idx) + 1)
        print(f"[{name}] reference heads {time.time()-t0:.1f}s ({sum(v.numel() for v in W.values())/1e6:.1f} M params)")
        orc = fw_heads_oracle.head_prediction(W, ol, hc, S, ph, pw)
        for k in ref:
            print(f"   oracle vs reference  {k:18s} rel-L2 = {rel(orc[k], ref[k]):.3e}   shape {tuple(ref[k].shape)}")
        golden = {k: v.to(torch.float32).contiguous() for k, v in ref.items()}
        golden["meta"] = dict(grid=(S, ph, pw), seed_weights=0, seed_tokens=3, torch=torch.__version__,
                              heads="full" if "_full_" in name else "small")
        path = os.path.join(ROOT, "tests", "golden", name + ".pt")
        torch.save(golden, path)
        print(f"[{name}] wrote {path} ({os.path.getsize(path)/1e6:.2f} MB)")


# Cases whose activations are too large to commit whole: noise_pred is stored in full, the intermediate streams as a fixed
# sample of rows.  name -> (cfg kwargs, grid, timestep, text_len, rows sampled per stream, per-block capture)
SIZED_CASES = {
    # BASELINE.json configs[0]: 2-block model, latents [1,16,9,64,64] -> L = 9216 DiT tokens, L2 = 9261 VGGT tokens.  The first
    # golden in which the engine takes its default 256x256 GEMM (M >= 2048), several attention query blocks, the XCD remap and
    # K/V ring wrap-around.  sample_steps = 1 -> the scheduler's only timestep is 1000.
    "wan21_cfg1_l2_f9_64x64": (dict(num_layers=2, start_index=1), (9, 64, 64), 1000.0, 512, 64, False),
    # depth: 4 PCB + 4 IRG blocks on 96 tokens; the streams after EVERY block are stored, so the error growth with depth of
    # the bf16 path against the fp32 reference is measured, not inferred
    "wan21_depth_l8_s4_f2_12x16": (dict(num_layers=8, start_index=4), (2, 12, 16), 750.0, 512, 24, True),
    # BASELINE.json configs[1] / [2] token grid (81f x 480 x 832 -> latents [1,16,21,60,104], L = 32760, L2 = 32865) on the 2-block
    # model: the headline SIZE pinned to the real reference (round 3; the reference forward is a few CPU-minutes).  Timestep =
    # the second one of the 50-step schedule (995.9 -> bf16 996).
    "wan21_cfg2_l2_f21_60x104": (dict(num_layers=2, start_index=1), (21, 60, 104), 996.0, 512, 64, False),
    # BASELINE.json configs[3] token grid (Wan2.2-Fun-A14B-Control-Camera, 81f x 720p -> latents [1,16,21,90,160], L = 75600,
    # L2 = 75705), control adapter at its real size
    "wan22_cfg4_l2_f21_90x160": (dict(num_layers=2, start_index=1), (21, 90, 160), 996.0, 512, 64, False),
    # the reference CLI's OWN default grid (inference_wan21.py:93-128: --height 336 --width 592 --frames 81 -> latents
    # [1,16,21,42,74]): 21 x 37 = 777 tokens per frame -- odd in both directions, L = 16317 (not a multiple of any tile size), 782
    # VGGT tokens per frame, L2 = 16422 -- the shape a user of the unmodified script gets first (round 4)
    "wan21_cli_l2_f21_42x74": (dict(num_layers=2, start_index=1), (21, 42, 74), 996.0, 512, 64, False),
    # BASELINE.json configs[4] token grid (Wan2.2, 121f x 720p -> latents [1,16,31,90,160], L = 111600, L2 = 111755): round 6 -- the live
    # fp32 reference forward at this size costs the GPU suite 315 s per run (tests/test_config5_gpu.py, now opt-in); as a golden it
    # is ~35 CPU-minutes ONCE here (FW_GOLDEN_SKIP_ORACLE=1 skips the CPU restatement's own forward at this size)
    "wan22_cfg5_l2_f31_90x160": (dict(num_layers=2, start_index=1), (31, 90, 160), 996.0, 512, 64, False),
}


def _sample_rows(n, k, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randperm(n, generator=g)[:min(k, n)].sort().values


def main_sized(only):
    for name, (ckw, (f, h2, w2), ts, tl, nrows, per_block) in SIZED_CASES.items():
        if only and name not in only:
            continue
        wan22 = name.startswith("wan22")
        cfg = (fwc.plumbing22 if wan22 else fwc.plumbing)(**ckw)
        W = synth.make_weights(cfg)
        ins = synth.make_inputs(cfg, f, h2, w2, seed=1, timestep=ts, text_len=tl)
        model = (ref_harness.build_reference_wan22 if wan22 else ref_harness.build_reference_wan21)(cfg, weights=W)
        assert not model._fw_unused
        L = f * (h2 // 2) * (w2 // 2)
        L2 = f * (cfg.n_special + (h2 // 2) * (w2 // 2))
        rows_dit, rows_agg = _sample_rows(L, nrows, 11), _sample_rows(L2, nrows, 12)
        cap = {"x_blocks": {}, "tok_blocks": {}}
        for b in range(cfg.start_index):
            model.pipe.dit.blocks[b].register_forward_hook(
                lambda m, a, out, b=b: cap["x_blocks"].__setitem__(b, out[0].detach()[rows_dit].clone()))
        for j in range(cfg.n_irg):
            def hook(m, a, out, j=j):
                cap["x_blocks"][cfg.start_index + j] = out[0][0].detach()[rows_dit].clone()
                cap["tok_blocks"][j] = out[1][0].detach().reshape(L2, -1)[rows_agg].clone()
            model.IRGBlock[j].register_forward_hook(hook)
        t0 = time.time()
        with torch.no_grad():
            if wan22:       # model_wan22.py:231-242 signature
                out, pred = model.joint_forward(
                    ins["x"], timestep=ins["timestep"], context=ins["context"], y=ins["y"], use_gradient_checkpointing=False,
                    camera_token=None, control_camera_latents_input=ins["control_camera_latents_input"], uncond=False,
                    return_prediction=False)
            else:
                out, pred = model.joint_forward(
                    ins["x"], timestep=ins["timestep"], context=ins["context"], clip_feature=ins["clip_feature"],
                    y=ins["y"], use_gradient_checkpointing=False, camera_token=None, plucker_fea=ins["plucker_fea"],
                    plucker_context_lens=ins["plucker_context_lens"], uncond=False, return_prediction=False)
        t_ref = time.time() - t0
        print(f"[{name}] reference forward {t_ref:.1f}s (L = {L}, L2 = {L2})", flush=True)
        del model
        last = cfg.num_layers - 1
        if os.environ.get("FW_GOLDEN_SKIP_ORACLE", "0") != "1":
            col = {}
            t0 = time.time()
            orc = fw_oracle.joint_forward(W, cfg, ins["x"], ins["timestep"], ins["context"], ins["clip_feature"], ins["y"],
                                          ins["plucker_fea"], ins["plucker_context_lens"], collect=col,
                                          control_camera_latents_input=ins.get("control_camera_latents_input"))
            print(f"[{name}] oracle forward {time.time()-t0:.1f}s", flush=True)
            print(f"   oracle vs reference  noise_pred     rel-L2 = {rel(orc, out):.3e}")
            print(f"   oracle vs reference  x_after_pcb    rel-L2 = {rel(col['x_after_pcb'][rows_dit], cap['x_blocks'][cfg.start_index - 1]):.3e}")
            print(f"   oracle vs reference  x_final        rel-L2 = {rel(col['x_final'][rows_dit], cap['x_blocks'][last]):.3e}")
            print(f"   oracle vs reference  tokens_final   rel-L2 = {rel(col['tokens_final'].reshape(L2, -1)[rows_agg], cap['tok_blocks'][cfg.n_irg - 1]):.3e}")
        golden = {"noise_pred": out.float().contiguous(), "rows_dit": rows_dit, "rows_agg": rows_agg,
                  "x_after_pcb": cap["x_blocks"][cfg.start_index - 1].float(), "x_final": cap["x_blocks"][last].float(),
                  "tokens_final": cap["tok_blocks"][cfg.n_irg - 1].float()}
        if per_block:
            golden["x_blocks"] = torch.stack([cap["x_blocks"][b] for b in range(cfg.num_layers)]).float()
            golden["tok_blocks"] = torch.stack([cap["tok_blocks"][j] for j in range(cfg.n_irg)]).float()
        golden["meta"] = dict(cfg=ckw, grid=(f, h2, w2), timestep=ts, text_len=tl, uncond=False, seed_weights=0,
                              seed_inputs=1, torch=torch.__version__, flavour="wan22" if wan22 else "wan21", sampled_rows=True,
                              reference_forward_s=round(t_ref, 1), cpu_threads=torch.get_num_threads())
        path = os.path.join(ROOT, "tests", "golden", name + ".pt")
        torch.save(golden, path)
        print(f"[{name}] wrote {path} ({os.path.getsize(path)/1e6:.2f} MB)", flush=True)


def _cold(name):
    """Parameters that are not on the per-step hot path (geometry heads, pose encoder, CamTokenProjector)."""
    return (name.startswith("vggt.camera_head") or name.startswith("vggt.depth_head") or
            name.startswith("vggt.point_head") or name.startswith("camera_condition.pose_encoder") or
            "CamTokenProjector" in name)


if __name__ == "__main__":
    args = sys.argv[1:]
    if not args or any(a in CASES for a in args):
        main()
    if not args or any(a in SIZED_CASES for a in args):
        main_sized([a for a in args if a in SIZED_CASES])
    if not args or any(a.startswith("heads") for a in args):
        main_heads([a for a in args if a.startswith("heads")])
    if not args or any("_pred_" in a for a in args):
        main_pred([a for a in args if "_pred_" in a])
    if not args or any(a.startswith("pose") for a in args):
        main_pose([a for a in args if a.startswith("pose")])
    if not args or any(a.startswith("vae") for a in args):
        main_vae([a for a in args if a.startswith("vae")])
