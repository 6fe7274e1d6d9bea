"""Host side of the loop (fantasy_world_amd.sampler): flow-match scheduler against the reference's, the CFG step algebra,
and the Wan2.2 dual-expert selection rule (inference_wan22.py:229-240)."""
import os

import pytest
import torch

from fantasy_world_amd.sampler import FlowMatchScheduler, denoise_step, denoise_step_dual, select_expert


class _StubEngine:
    """joint_forward stand-in: returns gain * latents + mean(context), and remembers the timesteps it was called with."""

    def __init__(self, gain):
        self.gain, self.calls = gain, []

    def joint_forward(self, latents, t, ctx, return_prediction=False, **cond):
        self.calls.append(float(t))
        return self.gain * latents + ctx.mean(), ({"tag": self.gain} if return_prediction else None)


def test_scheduler_known_values():
    s = FlowMatchScheduler()          # shift 5, 50 steps, extra_one_step (flow_match.py:18-41)
    assert len(s.sigmas) == 50 and float(s.sigmas[0]) == 1.0
    sig1 = 5 * 0.98 / (1 + 4 * 0.98)
    assert abs(float(s.sigmas[1]) - sig1) < 1e-6 and abs(float(s.timesteps[1]) - 1000 * sig1) < 1e-3
    x, v = torch.ones(3), torch.full((3,), 2.0)
    assert torch.allclose(s.step(v, 0, x), x + v * (sig1 - 1.0), atol=1e-6)
    assert torch.allclose(s.step(v, 49, x), x + v * (0.0 - float(s.sigmas[49])), atol=1e-6)      # last step goes to sigma 0


@pytest.mark.skipif(not __import__("oracle.ref_locate", fromlist=["x"]).available(), reason="reference not mounted / staged")
@pytest.mark.parametrize("steps", [1, 10, 50])
def test_scheduler_matches_reference(steps):
    from oracle import ref_harness
    ref_harness.install_stubs()
    from FantasyWorld.diffsynth_wan21.schedulers.flow_match import FlowMatchScheduler as Ref
    ref = Ref(shift=5, sigma_min=0.0, extra_one_step=True)
    ref.set_timesteps(steps)
    mine = FlowMatchScheduler()
    mine.set_timesteps(steps)
    assert torch.allclose(mine.sigmas, ref.sigmas, atol=1e-7) and torch.allclose(mine.timesteps, ref.timesteps, atol=1e-4)
    x, v = torch.randn(2, 5), torch.randn(2, 5)
    for i in (0, steps - 1):
        assert torch.allclose(mine.step(v, i, x), ref.step(v, ref.timesteps[i], x), atol=1e-6)


def test_cfg_step_algebra():
    s = FlowMatchScheduler()
    eng = _StubEngine(0.5)
    x = torch.randn(1, 4, 2, 2)
    pos, neg = torch.full((3,), 2.0), torch.full((3,), -1.0)
    out, pred = denoise_step(eng, s, 3, x, pos, neg, {}, cfg_scale=5.0, return_prediction=True)
    p, n = 0.5 * x + 2.0, 0.5 * x - 1.0
    want = x + (n + 5.0 * (p - n)) * (float(s.sigmas[4]) - float(s.sigmas[3]))
    assert torch.allclose(out, want, atol=1e-6) and pred == {"tag": 0.5}
    assert len(eng.calls) == 2 and abs(eng.calls[0] - float(s.timesteps[3])) < 1e-3


def test_dual_expert_selection():
    s = FlowMatchScheduler()
    hi, lo = _StubEngine(1.0), _StubEngine(2.0)
    boundary = 900.0                                    # inference_wan22.py: high-noise expert for t > boundary
    first_low = next(i for i in range(50) if float(s.timesteps[i]) <= boundary)
    assert 0 < first_low < 50
    assert select_expert(s, 0, hi, lo, boundary) is hi and select_expert(s, first_low - 1, hi, lo, boundary) is hi
    assert select_expert(s, first_low, hi, lo, boundary) is lo and select_expert(s, 49, hi, lo, boundary) is lo
    x, c = torch.zeros(2), torch.zeros(2)
    for i in range(50):
        denoise_step_dual(hi, lo, boundary, s, i, x, c, c, {})
    assert len(hi.calls) == 2 * first_low and len(lo.calls) == 2 * (50 - first_low)
