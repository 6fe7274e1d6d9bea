import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def load_golden(name):
    return torch.load(os.path.join(GOLDEN_DIR, name + ".pt"), map_location="cpu", weights_only=False)


class Case:
    """One golden case: config, synthetic weights and inputs regenerated from seeds, reference outputs from disk."""

    def __init__(self, name):
        from fantasy_world_amd import config as fwc, synth
        self.name = name
        self.golden = load_golden(name)
        meta = self.golden["meta"]
        self.cfg = fwc.plumbing(**meta["cfg"])
        f, h2, w2 = meta["grid"]
        self.grid = (f, h2, w2)
        self.uncond = meta["uncond"]
        self.weights = synth.make_weights(self.cfg, seed=meta["seed_weights"])
        self.inputs = synth.make_inputs(self.cfg, f, h2, w2, seed=meta["seed_inputs"], timestep=meta["timestep"],
                                        text_len=meta["text_len"])


@pytest.fixture(scope="session")
def case_l2():
    return Case("wan21_l2_f3_8x8")


@pytest.fixture(scope="session")
def case_l3():
    return Case("wan21_l3_f2_12x8")
