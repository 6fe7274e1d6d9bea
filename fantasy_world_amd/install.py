"""Drop-in boundary B1: rebind FantasyWorldFusionModel.joint_forward on a live reference model.

    from fantasy_world_amd import install
    sampler = FantasyWorldSampler(...)          # the reference's own inference_wan21.py object
    install(sampler.model)                      # weights are read from the live module tree (after .to(bf16), LoRA
                                                # merges, load_state_dict) and packed for the HIP kernels
    sampler.generate_video(...)                 # unchanged reference code now steps through libfw_mi355x.so

The replacement keeps the reference signature and return convention
(FantasyWorld/fusion/model_wan21.py:104-116,217-224): (noise_pred[B,16,F,H,W] in x.dtype, prediction dict | None).
On a Wan2.1 model `camera_condition.get_pose_fea` (the once-per-generation CameraPoseEncoder) is rebound too.
The once-per-generation geometry heads (vggt._head_predction, vggt/models/vggt.py:134-154) run on the same
engine (fantasy_world_amd.heads) when the model carries all three of them; a VGGT with a head switched off keeps the
reference's own head modules on the aggregated tokens the engine returns.
"""
import types

import torch

from .config import FWConfig, HeadsConfig
from .engine import FusionEngine


def config_from_model(model) -> FWConfig:
    """Read the shape description off the live reference module tree."""
    dit = model.pipe.dit
    blocks = list(dit.blocks)
    n_layers = len(blocks)
    first = model.IRGBlock[0].x_dit if len(model.IRGBlock) else blocks[0]
    first_dit = next((b for b in blocks if hasattr(b, "ffn")), first)
    cfg = FWConfig(
        dim=dit.dim, in_dim=dit.patch_embedding.in_channels, ffn_dim=first_dit.ffn_dim,
        out_dim=dit.head.head.out_features // 4, text_dim=dit.text_embedding[0].in_features, freq_dim=dit.freq_dim,
        eps=first_dit.norm1.eps, num_heads=first_dit.num_heads, num_layers=n_layers,
        has_image_input=dit.has_image_input, start_index=model.start_index,
        cross_attention_list=list(model.cross_attention_list), bicross_dim=model.bicross_dim,
        bicross_heads=model.bicross_num_heads, vggt_dim=model.vggt.embed_dim,
        # Wan2.1: per-block adapter processors installed by CameraConditionModel (model_wan21.py:55-60);
        # Wan2.2: camera enters through dit.control_adapter inside patchify (model_wan22.py:254, wan_video_dit.py:385-396)
        camera_adapter=bool(getattr(model, "camera_control", False)) and hasattr(model, "camera_condition"),
        control_adapter=getattr(dit, "control_adapter", None) is not None,
    )
    if cfg.control_adapter:
        cfg.control_in_dim = dit.control_adapter.conv.in_channels // 64
    return cfg


def heads_config_from_model(vggt):
    """HeadsConfig read off the live head modules (vggt/models/vggt.py:31-34), or None when one of them is disabled."""
    cam, dep, pts = (getattr(vggt, n, None) for n in ("camera_head", "depth_head", "point_head"))
    if cam is None or dep is None or pts is None:
        return None
    dim = cam.token_norm.normalized_shape[0]
    return HeadsConfig(
        dim_in=dim, trunk_depth=len(cam.trunk), cam_heads=cam.trunk[0].attn.num_heads,
        cam_mlp_ratio=cam.trunk[0].mlp.fc1.out_features // dim, features=dep.scratch.layer1_rn.out_channels,
        out_channels=[p.out_channels for p in dep.projects], layer_idx=list(dep.intermediate_layer_idx),
        dpt_patch=dep.patch_size, depth_out=dep.scratch.output_conv2[2].out_channels,
        point_out=pts.scratch.output_conv2[2].out_channels)


class _PackedParams:
    """The getter the engine packs from: name -> live Parameter, RECORDING every name it hands out.  The record is what
    `release_reference_weights` releases (exactly the tensors that were packed -- not everything under a prefix) and what
    _WeightWatch watches.  A name asked for AFTER its storage was released fails with a clear message instead of packing a 0-element
    tensor."""

    def __init__(self, params):
        self.params = params
        self.fetched = []
        self._seen = set()
        self.released = set()

    def __call__(self, name):
        if name in self.released:
            raise RuntimeError(f"{name}: the reference's copy of this parameter was released by "
                               "install(..., release_reference_weights=True); build / load the model again")
        if name not in self._seen:
            self._seen.add(name)
            self.fetched.append(name)
        return self.params[name]


class _WeightWatch:
    """Evidence that the live module tree still holds the tensors that were packed: EVERY packed parameter is remembered BY MODULE
    AND NAME at install time and its (storage address, in-place version) re-read per call -- ~1 700 slots, well under a millisecond
    against a forward of seconds, no walk over the module tree: load_state_dict, a LoRA merge done with in-place tensor ops, a partial
    load that touches a single block, `.to()` / `.half()` or re-assigned Parameters all change at least one watched slot.
    NOT detected: edits that go through `.data` (`p.data += delta`, `p.data.add_(...)`) -- autograd's version counter does not
    see them and the address stays; call `install(model)` again after such a merge, or `verify()` (one device sync: compares
    content checksums of a spread of the watched tensors with the ones taken at install time)."""

    def __init__(self, model, names=None, checksum_every=24):
        mods = dict(model.named_modules())
        if names is None:
            names = [k for k, _ in model.named_parameters()]
        self.slots = []
        for name in sorted(names):
            owner, _, leaf = name.rpartition(".")
            self.slots.append((name, mods[owner], leaf))
        self.sum_slots = self.slots[:: max(1, len(self.slots) // checksum_every)]
        self.signature = self._read()
        self.checksums = self._sums()

    def _read(self):
        out = []
        for name, mod, leaf in self.slots:
            p = mod._parameters.get(leaf)
            out.append((None, None) if p is None else (p.data_ptr(), p._version))
        return out

    def _sums(self):
        vals = [mod._parameters[leaf].detach().reshape(-1)[:4096].double().sum() for _, mod, leaf in self.sum_slots
                if mod._parameters.get(leaf) is not None and mod._parameters[leaf].numel()]
        return torch.stack(vals).cpu() if vals else torch.zeros(0)

    def sync(self, model, names):
        """Names packed AFTER install (the geometry heads on the first return_prediction call, the pose encoder on the first
        get_pose_fea) join the watch at the moment they were packed: called right after every engine call (the snapshot
        taken at install time never saw them, so a later load_state_dict touching only the heads went unnoticed)."""
        if len(names) == len(self.slots):
            return
        have = {n for n, _, _ in self.slots}
        mods = dict(model.named_modules())
        for name in names:
            if name in have:
                continue
            owner, _, leaf = name.rpartition(".")
            mod = mods[owner]
            p = mod._parameters.get(leaf)
            self.slots.append((name, mod, leaf))
            self.signature.append((None, None) if p is None else (p.data_ptr(), p._version))

    def unchanged(self):
        return self._read() == self.signature

    def changed_names(self, limit=4):
        now = self._read()
        return [self.slots[i][0] for i in range(len(now)) if now[i] != self.signature[i]][:limit]

    def verify(self):
        """Content check of a spread of the watched tensors (syncs the device once): also catches `.data` edits."""
        return self.unchanged() and torch.equal(self._sums(), self.checksums)


class CfgPairing:
    """CFG parallelism UNDER the reference's unchanged sampling loop (model_wan21.py:289-322; inference_wan22.py:229-277).

    The loop calls joint_forward twice per step -- positive prompt, then negative prompt -- with the SAME latents and timestep
    objects and, step after step, the SAME two context tensors.  With two CFG rank groups (parallel.Topology.cfg_groups == 2)
    the rebound joint_forward learns that pair during the first step (both groups compute both forwards, like one group would),
    and from the second step on runs the two forwards of a step CONCURRENTLY: at the positive call group 0 computes the positive
    and group 1 the negative forward, the two noise predictions are exchanged with one all-gather (4 MB), the positive one is
    returned and the negative one is kept for the call that follows; that call is recognised by identity (same latents object,
    same timestep object, the learned negative context) and answered from the stash.  Anything else -- another prompt,
    uncond=True, and with rank groups return_prediction=True (the last step needs the positive pass's geometry on every rank)
    -- takes the plain path and re-learns (the single-GPU merged pass serves the last step too: it returns the positive
    sample's prediction).  Every rank runs the same script on the same inputs, so every rank takes the same branch and returns
    the same tensors as a single-GPU run."""

    def __init__(self, topo=None):
        # topo = None: ONE GPU -- the pair is computed as one merged pass over 2L rows (FusionEngine.joint_forward_pair: weights read
        # once, bit-identical to two calls) at the loop's first call, the second call is answered from the stash all the same
        self.topo = topo
        self.pair = None          # (first context, second context) of a step, learned
        self.last = None          # (key of the previous plain call, its context)
        self.stash = None         # (key, second-context result) waiting for the second call

    @staticmethod
    def _key(x, timestep, cond):
        """Identity AND in-place version of everything the two forwards of a step share: the latents, the timestep and every
        conditioning tensor (clip_feature, y, camera_token, plucker_fea, plucker_context_lens, control_camera_latents_input).  The
        stashed second result was computed from the FIRST call's conditioning: a loop that hands the negative pass different
        conditioning, or that updates latents / timestep in place between the two calls, must not be answered from it."""
        items = [x, timestep] + [cond[k] for k in sorted(cond)]
        return tuple((id(t), t._version) if torch.is_tensor(t) else (None, t) for t in items), tuple(items)

    @staticmethod
    def _same(a, b):
        return a is not None and b is not None and a[0] == b[0] and all(p is q for p, q in zip(a[1], b[1]))

    def run(self, forward, x, timestep, context, uncond, return_prediction, forward_pair=None, cond=None):
        """forward(context, want_prediction) -> (out, prediction) on this rank's sequence-shard group;
        forward_pair(ctx_first, ctx_second, want_prediction) -> (out_first, out_second, prediction of the first): the merged pass
        (single GPU only).  `cond`: name -> conditioning tensor of this call (part of the stash key)."""
        plain = uncond or (return_prediction and self.topo is not None)
        key = self._key(x, timestep, cond or {})
        st = self.stash
        self.stash = None
        # the second call of a step: same latents / timestep / conditioning objects, unmodified since the first call, and the learned
        # negative context.  A second call that asks for the prediction takes the plain path (the stash holds none for it).
        if (st is not None and not plain and not return_prediction and self._same(st[0], key) and self.pair is not None
                and context is self.pair[1]):
            return st[1], None
        if not plain and self.pair is not None and context is self.pair[0]:
            pred = None
            if self.topo is not None:
                out, _ = forward(self.pair[self.topo.cfg_rank], False)
                first, second = self.topo.gather_cfg(out)
            else:
                first, second, pred = forward_pair(self.pair[0], self.pair[1], return_prediction)
            self.stash = (key, second)
            return first, pred
        if self.last is not None and self._same(self.last[0], key) and self.last[1] is not context:
            self.pair = (self.last[1], context)
        self.last = (key, context)
        return forward(context, return_prediction)


def install(model, ops=None, device=None, cache_step_invariants=True, precision="bf16", topo=None, shard=None,
            release_reference_weights=False, merge_cfg=True, fp8_attention=False):
    """Replace `model.joint_forward` by the MI355X engine.  `ops` defaults to HipOps (raises without a GPU / library);
    tests may inject another op set to exercise this boundary on CPU.  `cache_step_invariants`: keep the context embeddings,
    the per-block cross-attention K/V and the camera adapter's Pluecker term across the calls of a generation (the caller
    passes the same tensors 100 times; results are bit-identical).  `precision`: "bf16", or "fp8" = the
    DiT blocks' linears through the reference's fp8 linear.  `fp8_attention=True` (with precision="fp8":
    BASELINE config 5, "fp8 attention + FFN"): the DiT self-attention on e4m3 q / k / v / probabilities as well.  The reference
    defines fp8 for nn.Linear only, so this option has NO reference semantics to be exact against: its stated, test-enforced
    tolerance is 2e-2 rel-L2 of noise_pred against the same engine with bf16 attention (measured 1.35e-2 at 40 / 24 / 24 blocks,
    INTEGRATION.md).  Both options work on one GPU and under either multi-GPU partition (`topo`).  `fp8_attention="bicross"`
    (one GPU; round-6 experiment): additionally the two directions of the bicross attention through the same kernel on zero-padded
    heads -- measured 1.36e-2 against bf16 attention everywhere at 40 / 24 / 24 blocks, the same stated 2e-2; `fp8_attention="all"`
    (one GPU; second half of round 6): additionally the VGGT frame / global attention (head_dim 64) on fw_attention_fp8's head_dim-64
    kernel.

    Several GPUs (one process per GPU, every process running the SAME reference script on the same inputs): pass
    `topo=fantasy_world_amd.parallel.init_topology()` (or a bare `shard=SequenceShard(...)`).  The forward is then
    sequence-sharded over the ranks of this process's group (head all-to-all, parallel.py) and, with two CFG groups, the two
    forwards of a sampling step run concurrently under the unchanged reference loop (CfgPairing).  Every rank returns the full
    noise prediction (and, on the last step, the full prediction dict), identical across ranks.

    `merge_cfg` (single GPU, default on): the two forwards of a sampling step -- two sequential joint_forward calls of the unchanged
    reference loop that differ only in the text context -- run as ONE pass over 2L rows from the second step on (CfgPairing with
    FusionEngine.joint_forward_pair): bit-identical outputs, +1.2 % steps/s (profiles/r03/bench_r3_merge_cfg.log).

    `release_reference_weights=True`: after packing (geometry heads and pose encoder included, eagerly) the storage of every
    parameter that was packed is released on the reference module tree (the Parameters stay, with empty data): the model then holds
    ONE copy of its weights -- the packed one -- instead of 28 GB + 36 GB per expert.  One-way: `uninstall()` cannot bring the
    reference forward back (it raises), load the checkpoint again for that.

    The weights are SNAPSHOT at install time into packed copies (36 GB for the 14B model, next to the reference's own): call
    install() after checkpoint loading / LoRA merging / .to(dtype).  A later change of the live parameters is detected at the
    next joint_forward (RuntimeError: install again) instead of being silently ignored; installing again drops the previous
    engine and its caches first."""
    if getattr(model, "_fw_released_weights", 0):
        raise RuntimeError("this model's reference weights were released by install(..., release_reference_weights=True): there is "
                           "nothing left to pack from -- build / load the model again, then install()")
    if hasattr(model, "_fw_engine"):
        model._fw_engine.invariants.clear()
        uninstall(model)
    if ops is None:
        from .hip_ops import HipOps
        ops = HipOps(device or "cuda")
    cfg = config_from_model(model)
    if topo is not None:
        assert shard is None or shard is topo.shard, "pass either topo or shard"
        shard = topo.shard
    pairing = CfgPairing(topo) if topo is not None and topo.cfg_groups == 2 else None
    if pairing is None and merge_cfg and shard is None and (topo is None or topo.world == 1):
        pairing = CfgPairing(None)
    get = _PackedParams(dict(model.named_parameters()))
    if topo is not None and topo.tp is not None:          # north_star's head / FFN-column partition (tensor_parallel.py)
        from .tensor_parallel import TPFusionEngine
        engine = TPFusionEngine(cfg, get, ops, topo.tp, heads_cfg=heads_config_from_model(model.vggt),
                                cache_step_invariants=cache_step_invariants, precision=precision, fp8_attention=fp8_attention)
    else:
        engine = FusionEngine(cfg, get, ops, heads_cfg=heads_config_from_model(model.vggt),
                              cache_step_invariants=cache_step_invariants, precision=precision, shard=shard,
                              fp8_attention=fp8_attention)
    if engine.shard is not None and engine.shard.world > 1:
        # the grouped q|k|v exchange is probed once on the side communicator, exactly as parallel.make_engine does for bench.py: a
        # multi-GPU drop-in run of the reference script gets the same first-run safety.  A bare SequenceShard without
        # a probe communicator is NOT probed on the forward's own communicator (a stalled probe would leave its collectives there,
        # which is the hang the probe exists to prevent): it runs one exchange per attention.
        engine.exchange_groups = engine.shard.negotiate_exchange_groups(engine.exchange_groups, getattr(ops, "device", None) or "cpu")
    if release_reference_weights:
        if engine.heads_cfg is not None:
            engine.geometry_heads()                              # pack now: their source tensors are about to go
        # EXACTLY what the engine (and its geometry heads) fetched: the track head, aggregator blocks beyond the IRG depth, the
        # aggregator's own patch embedding, the pose encoder (packed lazily by get_pose_fea) and anything a later reference version
        # adds under pipe.dit / vggt keep their storage and stay usable
        with torch.no_grad():
            for name in get.fetched:
                p = get.params[name]
                p.data = torch.empty(0, dtype=p.dtype, device=p.device)
                get.released.add(name)
        model._fw_released_weights = len(get.released)
    watch = _WeightWatch(model, names=get.fetched)
    engine.weight_watch = watch
    import weakref
    model_ref = weakref.ref(model)

    def joint_forward(self, x, timestep, context, clip_feature=None, y=None, use_gradient_checkpointing=True,
                      camera_token=None, plucker_fea=None, plucker_context_lens=None, uncond=False,
                      return_prediction=False, control_camera_latents_input=None, **kwargs):
        if not watch.unchanged():
            raise RuntimeError("the model's parameters changed after fantasy_world_amd.install() (load_state_dict, LoRA merge, "
                               f".to(); e.g. {watch.changed_names()}): the engine runs on a packed snapshot -- "
                               + ("build / load the model again (its reference weights were released)"
                                  if getattr(self, "_fw_released_weights", 0) else "call install(model) again"))
        def forward(ctx, want_prediction):
            return engine.joint_forward(x, timestep, ctx, clip_feature=clip_feature, y=y,
                                        plucker_fea=plucker_fea, plucker_context_lens=plucker_context_lens,
                                        uncond=uncond, return_prediction=want_prediction, camera_token=camera_token,
                                        control_camera_latents_input=control_camera_latents_input)
        def forward_pair(ctx_a, ctx_b, want_prediction):
            return engine.joint_forward_pair(x, timestep, ctx_a, ctx_b, clip_feature=clip_feature, y=y, plucker_fea=plucker_fea,
                                             plucker_context_lens=plucker_context_lens, uncond=uncond,
                                             return_prediction=want_prediction, camera_token=camera_token,
                                             control_camera_latents_input=control_camera_latents_input)
        if pairing is not None:
            out, outputs = pairing.run(forward, x, timestep, context, uncond, return_prediction, forward_pair,
                                       cond=dict(clip_feature=clip_feature, y=y, camera_token=camera_token, plucker_fea=plucker_fea,
                                                 plucker_context_lens=plucker_context_lens,
                                                 control_camera_latents_input=control_camera_latents_input))
        else:
            out, outputs = forward(context, return_prediction)
        watch.sync(self, get.fetched)                # whatever this call packed lazily (geometry heads) is watched from here on
        if not return_prediction:
            return out, None
        if engine.heads_cfg is not None:
            return out, outputs                      # the prediction dict, computed by fantasy_world_amd.heads
        n = cfg.n_irg
        output_list = [outputs.get(i) for i in range(n)]
        f, h, w = x.shape[2], x.shape[3] // 2, x.shape[4] // 2
        # dpt_head only reads the shape of `images` ([B,S,h,w,C], dpt_head.py:141,215)
        patch_token = torch.empty(1, f, h, w, cfg.vggt_dim, dtype=out.dtype, device=out.device)
        prediction = self.vggt._head_predction(patch_token, self.vggt.aggregator.patch_start_idx, output_list)
        return out, prediction

    def joint_forward22(self, x, timestep, context, y=None, use_gradient_checkpointing=True, camera_token=None,
                        control_camera_latents_input=None, uncond=False, return_prediction=False, **kwargs):
        """Wan2.2 signature (FantasyWorld/fusion/model_wan22.py:231-242)."""
        return joint_forward(self, x, timestep, context, y=y, camera_token=camera_token, uncond=uncond,
                             return_prediction=return_prediction,
                             control_camera_latents_input=control_camera_latents_input)

    # An instance attribute holding a BOUND method would tie model -> method -> model into a reference cycle (36 GB of packed weights
    # waiting for the cyclic collector after `del model`): the rebound entry reaches the model through a weak reference instead.
    impl = joint_forward22 if cfg.control_adapter else joint_forward

    def rebound(*args, **kwargs):
        return impl(model_ref(), *args, **kwargs)
    rebound.__name__, rebound.__doc__ = "joint_forward", impl.__doc__
    model._fw_reference_joint_forward = model.__dict__.get("joint_forward")      # None: the class's own method
    model.joint_forward = rebound
    model._fw_engine = engine
    engine.cfg_pairing = pairing

    # Wan2.1: the producer of plucker_fea (CameraConditionModel.get_pose_fea, camera_control.py:233-234; called once per
    # generation from generate_video, model_wan21.py:271) moves onto the same op set; packed on first use
    cam = getattr(model, "camera_condition", None)
    if cam is not None and getattr(cam, "pose_encoder", None) is not None and getattr(cam.pose_encoder, "pose_inject_method", "") == "adaln":
        state = {}

        def get_pose_fea(self, plucker):
            if plucker is None:
                return None
            if "enc" not in state:
                from .pose_encoder import PoseEncoder
                state["enc"] = PoseEncoder(get, ops)         # through the recording getter: released / watched like every packed name
                watch.sync(model_ref(), get.fetched)
            return state["enc"].encode(plucker)

        cam_ref = weakref.ref(cam)
        cam._fw_reference_get_pose_fea = cam.__dict__.get("get_pose_fea")
        cam.get_pose_fea = lambda plucker: get_pose_fea(cam_ref(), plucker)
    return engine


def install_vae(vae, ops=None, device=None):
    """Rebind `vae.model.decode(z, scale)` of a reference WanVideoVAE (diffsynth_wan21/models/wan_video_vae.py:552-575, the call
    `tiled_decode` / `single_decode` make per tile, :643-692, :752-755) to fantasy_world_amd.vae_decoder.  Tiling, blending and
    the final clamp stay the reference's own code.  Returns an `undo()` callable; weights are packed on first use."""
    if ops is None:
        from .hip_ops import HipOps
        ops = HipOps(device or "cuda")
    inner = vae.model
    state = {}

    def decode(self, z, scale):
        if "dec" not in state:
            from .vae_decoder import VaeDecoder
            dec = inner.decoder
            state["dec"] = VaeDecoder(dict(inner.named_parameters()).__getitem__, ops, dim=dec.dim, z_dim=dec.z_dim,
                                      dim_mult=tuple(dec.dim_mult), num_res_blocks=dec.num_res_blocks,
                                      temporal_upsample=tuple(dec.temperal_upsample))
        return state["dec"].decode(z.to(ops.device), scale)

    original = inner.decode
    inner.decode = types.MethodType(decode, inner)

    def undo():
        inner.decode = original
    return undo


def uninstall(model):
    if getattr(model, "_fw_released_weights", 0):
        raise RuntimeError("install(..., release_reference_weights=True) released the reference's own copy of the weights: the "
                           "reference forward cannot be restored -- build / load the model again")
    def restore(obj, attr, saved):
        prev = obj.__dict__.pop(saved)
        if prev is None:
            obj.__dict__.pop(attr, None)          # back to the class's own method
        else:
            setattr(obj, attr, prev)
    if "_fw_reference_joint_forward" in model.__dict__:
        restore(model, "joint_forward", "_fw_reference_joint_forward")
        model._fw_engine.invariants.clear()
        del model._fw_engine
    cam = getattr(model, "camera_condition", None)
    if cam is not None and "_fw_reference_get_pose_fea" in cam.__dict__:
        restore(cam, "get_pose_fea", "_fw_reference_get_pose_fea")


def install_flash_attention(modules, ops=None, device=None, name="flash_attention"):
    """Boundary B3: rebind the reference's op-level hook
    `flash_attention(q, k, v, num_heads, compatibility_mode=False)` (diffsynth_wan21/models/wan_video_dit.py:28-66, same in
    diffsynth_wan22) in the given reference modules, so an otherwise-reference forward runs its attention through
    fw_attention_bf16.  q/k/v are [b, s, heads*hd] tensors; the result has q's shape and dtype.  Returns an `undo()` callable.

        import FantasyWorld.diffsynth_wan21.models.wan_video_dit as dit
        undo = install_flash_attention([dit])
    """
    if ops is None:
        from .hip_ops import HipOps
        ops = HipOps(device or "cuda")

    def flash_attention(q, k, v, num_heads, compatibility_mode=False):
        b, lq, d = q.shape
        lk = k.shape[1]
        hd = d // num_heads
        to2 = lambda t, n: ops.to_act(t.reshape(b * n, d))
        # softmax_scale * log2(e) folded into a COPY of q (fw_qk_prep, one bf16 rounding): the log2-domain kernels take it from there
        qs = ops.qk_prep(to2(q, lq).clone(), num_heads, hd, out_scale=ops.q_scale(hd))
        out = ops.attention(qs, to2(k, lk), to2(v, lk), num_heads, hd, batch=b, q_prescaled=True)
        return out.view(b, lq, d).to(q.dtype)

    saved = [(m, getattr(m, name)) for m in modules]
    for m, _ in saved:
        setattr(m, name, flash_attention)

    def undo():
        for m, f in saved:
            setattr(m, name, f)
    return undo
