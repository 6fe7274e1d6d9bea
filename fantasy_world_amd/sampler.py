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

    def step(self, model_output, step_id, sample):
        sigma = float(self.sigmas[step_id])
        sigma_next = float(self.sigmas[step_id + 1]) if step_id + 1 < len(self.sigmas) else 0.0
        return sample + model_output * (sigma_next - sigma)


@torch.no_grad()
def denoise_step(engine, scheduler, step_id, latents, ctx_pos, ctx_neg, cond, cfg_scale=5.0, return_prediction=False,
                 topo=None, merge_cfg=False):
    """One sampling step = 2 joint_forward calls (CFG) + combine + scheduler update (M21:289-322).
    topo (fantasy_world_amd.parallel.Topology) with two CFG groups: this rank runs only its group's forward and the two
    noise predictions are exchanged with one all-gather; the geometry prediction lives on the positive-prompt group.
    merge_cfg (single GPU): both forwards in one pass over 2L rows (FusionEngine.joint_forward_pair; bit-identical results)."""
    t = scheduler.timesteps[step_id].reshape(1).to(device=latents.device, dtype=latents.dtype)
    if merge_cfg and (topo is None or topo.world == 1):
        pos, neg, pred = engine.joint_forward_pair(latents, t, ctx_pos, ctx_neg, return_prediction=return_prediction, **cond)
    elif topo is not None and topo.cfg_groups == 2:
        mine = ctx_pos if topo.cfg_rank == 0 else ctx_neg
        out, pred = engine.joint_forward(latents, t, mine, return_prediction=return_prediction and topo.cfg_rank == 0, **cond)
        pos, neg = topo.gather_cfg(out)
    else:
        pos, pred = engine.joint_forward(latents, t, ctx_pos, return_prediction=return_prediction, **cond)
        neg, _ = engine.joint_forward(latents, t, ctx_neg, **cond)
    noise_pred = neg + cfg_scale * (pos - neg)
    return scheduler.step(noise_pred, step_id, latents), pred


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
