"""Host side of the denoising loop (stays PyTorch, as in the reference): flow-matching Euler scheduler and the CFG
denoise step around joint_forward.

  FlowMatchScheduler   FantasyWorld/diffsynth_wan21/schedulers/flow_match.py:18-53 (shift 5, sigma_min 0,
                       extra_one_step=True as diffsynth's Wan pipeline configures it)
  denoise_step         FantasyWorld/fusion/model_wan21.py:289-322 (pos pass, neg pass, neg + s*(pos-neg), Euler update)
"""
import torch


class FlowMatchScheduler:
    def __init__(self, shift=5.0, sigma_min=0.0, sigma_max=1.0, num_train_timesteps=1000, extra_one_step=True):
        self.shift, self.sigma_min, self.sigma_max = shift, sigma_min, sigma_max
        self.num_train_timesteps = num_train_timesteps
        self.extra_one_step = extra_one_step
        self.set_timesteps(50)

    def set_timesteps(self, num_inference_steps):
        if self.extra_one_step:
            s = torch.linspace(self.sigma_max, self.sigma_min, num_inference_steps + 1)[:-1]
        else:
            s = torch.linspace(self.sigma_max, self.sigma_min, num_inference_steps)
        self.sigmas = self.shift * s / (1 + (self.shift - 1) * s)
        self.timesteps = self.sigmas * self.num_train_timesteps
        self._dev = {}

    def dsigma(self, step_id):
        """sigma_{i+1} - sigma_i as the reference forms it (fp32 tensor arithmetic, flow_match.py:47-52; 0 after the last step)."""
        nxt = self.sigmas[step_id + 1] if step_id + 1 < len(self.sigmas) else torch.zeros((), dtype=self.sigmas.dtype)
        return float(nxt - self.sigmas[step_id])

    def step(self, model_output, step_id, sample):
        return sample + model_output * self.dsigma(step_id)

    def timestep_on(self, step_id, device, dtype):
        """The step's timestep as a 1-element DEVICE tensor in the latents' dtype (model_wan21.py:292-293: cast to bf16 first),
        sliced from a table uploaded once per (device, dtype): no host-to-device copy inside the sampling loop."""
        key = (str(device), dtype, self.timesteps.data_ptr(), len(self.timesteps))
        tab = self._dev.get(key)
        if tab is None:
            self._dev = {key: self.timesteps.to(device=device, dtype=dtype)}
            tab = self._dev[key]
        return tab[step_id:step_id + 1]


_SIDE_STREAMS = {}


def _side_stream(device):
    dev = torch.device(device)
    key = (dev.type, dev.index)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(dev)
    return _SIDE_STREAMS[key]


@torch.no_grad()
def denoise_step(engine, scheduler, step_id, latents, ctx_pos, ctx_neg, cond, cfg_scale=5.0, return_prediction=False,
                 topo=None, merge_cfg=False, cfg_streams=False):
    """One sampling step = 2 joint_forward calls (CFG) + combine + scheduler update (M21:289-322).
    topo (fantasy_world_amd.parallel.Topology) with two CFG groups: this rank runs only its group's forward and the two
    noise predictions are exchanged with one all-gather; the geometry prediction lives on the positive-prompt group.
    merge_cfg (single GPU): both forwards in one pass over 2L rows (FusionEngine.joint_forward_pair; bit-identical results).
    cfg_streams (single GPU, round 4): the two forwards -- independent until the combine -- on TWO HIP streams, so the HBM-bound row
    passes and the grid tails of one forward (7.5 % of a step's kernel time is non-MFMA work, serialised on one stream) run beside
    the other forward's matrix kernels; same kernels, same arithmetic per forward: bit-identical results.  Needs an engine whose
    step-invariant cache is off or already warm (its entries are produced on the stream that first asks for them)."""
    t = scheduler.timestep_on(step_id, latents.device, latents.dtype)
    if cfg_streams and (topo is None or topo.world == 1) and latents.is_cuda:
        # The engine fills shared state lazily on whichever stream asks first (rotary tables per grid, the all-zero verdict of the
        # Pluecker features, lazily packed heads, step-invariant cache entries): the first step on a grid therefore runs on ONE stream
        # and only later steps fork (a cold two-stream step could read a table the other stream was still writing).
        # keyed on the grid AND on the step-invariant cache's generation: after invariants.clear() (or a cache switched on later) the
        # next step fills the cache again and must not fork
        inv = getattr(engine, "invariants", None)
        key = (tuple(latents.shape), getattr(inv, "generation", 0), bool(getattr(inv, "enabled", False)))
        if getattr(engine, "_cfg_streams_warm", None) != key:
            engine._cfg_streams_warm = key
            cfg_streams = False
    if cfg_streams and (topo is None or topo.world == 1) and latents.is_cuda:
        cur, side = torch.cuda.current_stream(latents.device), _side_stream(latents.device)
        side.wait_stream(cur)                                 # inputs (latents, timestep, conditioning) were produced on `cur`
        with torch.cuda.stream(side):
            neg, _ = engine.joint_forward(latents, t, ctx_neg, **cond)
        pos, pred = engine.joint_forward(latents, t, ctx_pos, return_prediction=return_prediction, **cond)
        cur.wait_stream(side)
        neg.record_stream(cur)                                # allocated on the side stream's pool, consumed (and freed) on `cur`
    elif merge_cfg and (topo is None or topo.world == 1):
        pos, neg, pred = engine.joint_forward_pair(latents, t, ctx_pos, ctx_neg, return_prediction=return_prediction, **cond)
    elif topo is not None and topo.cfg_groups == 2:
        mine = ctx_pos if topo.cfg_rank == 0 else ctx_neg
        out, pred = engine.joint_forward(latents, t, mine, return_prediction=return_prediction and topo.cfg_rank == 0, **cond)
        pos, neg = topo.gather_cfg(out)
    else:
        pos, pred = engine.joint_forward(latents, t, ctx_pos, return_prediction=return_prediction, **cond)
        neg, _ = engine.joint_forward(latents, t, ctx_neg, **cond)
    return combine_and_step(getattr(engine, "ops", None), scheduler, step_id, latents, pos, neg, cfg_scale), pred


def combine_and_step(ops, scheduler, step_id, latents, pos, neg, cfg_scale=5.0, dev_params=None, out=None):
    """CFG combine + Euler update (model_wan21.py:318-321): one fw_cfg_euler_step launch on op sets that have it (bit-identical to
    the reference's tensor ops), the tensor ops themselves otherwise."""
    if hasattr(ops, "cfg_euler_step"):
        return ops.cfg_euler_step(pos, neg, latents, cfg_scale, scheduler.dsigma(step_id), out=out, dev_params=dev_params)
    return scheduler.step(neg + cfg_scale * (pos - neg), step_id, latents)


def select_expert(scheduler, step_id, engine_high, engine_low, timestep_boundary):
    """Wan2.2 dual-expert rule (inference_wan22.py:229-240): the high-noise expert while the step's timestep is above the
    boundary (compared on the host, like the reference's `t.item()`), the low-noise expert afterwards.  Both engines stay
    resident (2 x 36 GB of packed weights in 288 GB of HBM), so switching costs nothing."""
    return engine_high if float(scheduler.timesteps[step_id]) > float(timestep_boundary) else engine_low


def denoise_step_dual(engine_high, engine_low, timestep_boundary, scheduler, step_id, latents, ctx_pos, ctx_neg, cond,
                      cfg_scale=5.0, return_prediction=False, topo=None, merge_cfg=False):
    """One step of `generate_video_with_dual_models` (inference_wan22.py:227-277): pick the expert, then an ordinary step."""
    engine = select_expert(scheduler, step_id, engine_high, engine_low, timestep_boundary)
    return denoise_step(engine, scheduler, step_id, latents, ctx_pos, ctx_neg, cond, cfg_scale=cfg_scale,
                        return_prediction=return_prediction, topo=topo, merge_cfg=merge_cfg)


class GraphedDenoiseStep:
    """one WHOLE sampling step -- two joint_forward passes (or the merged pass), the CFG combine and the
    Euler update, ~8400 kernel launches -- captured once in a HIP graph and replayed per step.  Everything that changes from step
    to step lives in device memory the graph reads: the latents (static buffer), the timestep (1-element tensor sliced from a
    device table) and (cfg_scale, sigma_{i+1} - sigma_i) (device float[2], read by fw_cfg_euler_step), so a replay needs three
    small device-to-device copies and no host-to-device traffic.  Same kernels, same order, same arithmetic as the eager step:
    bit-identical results.  Single GPU (the collectives of a sequence shard are not captured)."""

    def __init__(self, engine, scheduler, latents, ctx_pos, ctx_neg, cond, cfg_scale=5.0, merge_cfg=False):
        assert engine.shard is None, "GraphedDenoiseStep is the single-GPU form"
        assert hasattr(engine.ops, "cfg_euler_step"), "needs the HIP op set"
        dev, n = latents.device, len(scheduler.timesteps)
        self.engine = engine
        self.lat = latents.clone()
        self.t_table = scheduler.timesteps.to(device=dev, dtype=latents.dtype)
        self.p_table = torch.tensor([[cfg_scale, scheduler.dsigma(i)] for i in range(n)], dtype=torch.float32, device=dev)
        self.t = self.t_table[:1].clone()
        self.params = self.p_table[0].clone()

        def body():
            if merge_cfg:
                pos, neg, _ = engine.joint_forward_pair(self.lat, self.t, ctx_pos, ctx_neg, **cond)
            else:
                pos, _ = engine.joint_forward(self.lat, self.t, ctx_pos, **cond)
                neg, _ = engine.joint_forward(self.lat, self.t, ctx_neg, **cond)
            return engine.ops.cfg_euler_step(pos, neg, self.lat, cfg_scale, 0.0, dev_params=self.params)

        # warm-up on a side stream, outside the capture: rotary tables, the adapter's all-zero verdict (one host sync), caches
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            body()
        cur.wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = body()

    def step(self, step_id, latents):
        """-> the next latents (a buffer owned by the graph: consumed / copied by the caller before the next replay)."""
        self.lat.copy_(latents)
        self.t.copy_(self.t_table[step_id:step_id + 1])
        self.params.copy_(self.p_table[step_id])
        self.graph.replay()
        return self.out
