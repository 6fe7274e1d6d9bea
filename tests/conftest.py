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
    # the CPU-side checkers run many small fp32 ops: more than a handful of threads only spins (and the hosts differ widely)
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a machine without a GPU: the gpu-marked tests skip instead of erroring in their fixtures."""
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (torch.cuda.is_available() is False)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def load_golden(name):
    return torch.load(os.path.join(GOLDEN_DIR, name + ".pt"), map_location="cpu", weights_only=False)


class BF16Weights(dict):
    """name -> tensor, stored in bf16 (the synthetic weights are bf16-representable by construction, so this is exact) and handed
    out as fp32.  Halves what the session-scoped cases keep resident: several 1 B-parameter dictionaries plus the spawned gloo
    ranks would not fit the build container's memory otherwise."""

    def __init__(self, src=()):
        super().__init__()
        self.update(src)

    def __setitem__(self, k, v):
        super().__setitem__(k, v.detach().to(torch.bfloat16) if v.dtype == torch.float32 else v)

    def update(self, other=()):
        for k, v in (other.items() if hasattr(other, "items") else other):
            self[k] = v

    def __getitem__(self, k):
        v = super().__getitem__(k)
        return v.to(torch.float32) if v.dtype == torch.bfloat16 else v

    def get(self, k, default=None):
        return self[k] if k in self else default

    def items(self):
        return ((k, self[k]) for k in self.keys())

    def values(self):
        return (self[k] for k in self.keys())


class Case:
    """One golden case: config, synthetic weights and inputs regenerated from seeds, reference outputs from disk."""

    def __init__(self, name):
        from fantasy_world_amd import config as fwc, synth
        self.name = name
        self.golden = load_golden(name)
        meta = self.golden["meta"]
        self.cfg = (fwc.plumbing22 if meta.get("flavour") == "wan22" else fwc.plumbing)(**meta["cfg"])
        f, h2, w2 = meta["grid"]
        self.grid = (f, h2, w2)
        self.uncond = meta["uncond"]
        self.weights = BF16Weights(synth.make_weights(self.cfg, seed=meta["seed_weights"]))
        self.inputs = synth.make_inputs(self.cfg, f, h2, w2, seed=meta["seed_inputs"], timestep=meta["timestep"],
                                        text_len=meta["text_len"])


@pytest.fixture(scope="session")
def case_l2():
    return Case("wan21_l2_f3_8x8")


@pytest.fixture(scope="session")
def case_l3():
    return Case("wan21_l3_f2_12x8")


@pytest.fixture(scope="session")
def case_w22():
    """Wan2.2-Fun-A14B-Control-Camera flavour: control adapter in patchify, text-only context (model_wan22.py)."""
    return Case("wan22_l2_f2_8x12")


@pytest.fixture(scope="session")
def case_camtok():
    """joint_forward(camera_token=...) : per-frame camera tokens from CamTokenProjector (aggregator.py:265-266)."""
    return Case("wan21_camtok_l2_f3_8x8")


@pytest.fixture(scope="session")
def case_cfg1():
    """BASELINE.json configs[0]: 2-block model on latents [1,16,9,64,64] (L = 9216, L2 = 9261); noise_pred in full, the streams
    as 64 sampled rows (golden["rows_dit"], golden["rows_agg"])."""
    return Case("wan21_cfg1_l2_f9_64x64")


@pytest.fixture(scope="session")
def case_depth():
    """4 PCB + 4 IRG blocks on 96 tokens with the streams after EVERY block (24 sampled rows each)."""
    return Case("wan21_depth_l8_s4_f2_12x16")


class ParityLog:
    """Measured errors of the run, written to gpurun_out/parity_<gpu|cpu>.json at session end (copied to profiles/rNN/parity.json):
    every assert that goes through check() records (measured, bound), so a regression inside the bound is still visible."""

    def __init__(self):
        self.rows = {}
        # Tight bounds = 1.5 x (2.5 x where the checker runs on PyTorch-ROCm kernels) the value measured on MI355X (tools/make_parity_bounds.py from profiles/rNN/parity.json; the
        # kernels are deterministic, so a measured value reproduces bit for bit on any gfx950).  The bound written in the test
        # is the PHYSICAL one (what the arithmetic may cost at most); the tight one turns a silent regression inside it -- say
        # 1.2e-3 -> 3.9e-3 under a 4e-3 bound -- into a failure.  Only applied on a GPU run.
        self.tight = {}
        path = os.path.join(GOLDEN_DIR, "parity_bounds_gpu.json")
        # FW_PARITY_TIGHT=0: physical bounds only -- the run that REGENERATES the tight bounds after a deliberate bit-moving change
        # (compiler flags, a kernel edit; tests/test_hip_ops.py::test_production_kernel_outputs_match_committed_digests names those)
        if os.path.exists(path) and torch.cuda.is_available() and os.environ.get("FW_PARITY_TIGHT", "1") != "0":
            import json
            self.tight = json.load(open(path))["bounds"]

    def check(self, name, value, bound):
        tight = self.tight.get(name)
        eff = float(bound) if tight is None else min(float(bound), float(tight))
        self.rows[name] = {"measured": float(value), "bound": float(bound), "enforced": eff}
        assert value < eff, f"{name}: {value:.3e} >= {eff:.3e} (physical bound {bound:.1e}" + (
            "" if tight is None else f", tight bound from measurements {tight:.3e}") + ")"
        return value

    def note(self, name, value):
        self.rows[name] = value


PARITY = ParityLog()


@pytest.fixture(scope="session")
def parity():
    return PARITY


def pytest_sessionfinish(session, exitstatus):
    if not PARITY.rows:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    kind = "gpu" if torch.cuda.is_available() else "cpu"
    path = os.path.join(out, f"parity_{kind}.json")
    old = {}
    if os.path.exists(path):
        try:
            old = json.load(open(path))
        except Exception:
            old = {}
    old.update(PARITY.rows)
    with open(path, "w") as f:
        json.dump(old, f, indent=1, sort_keys=True)


def forward_kwargs(case, dev=None):
    """joint_forward keyword arguments of a case (both flavours), optionally moved to a device."""
    ins = case.inputs
    mv = (lambda v: v.to(dev) if torch.is_tensor(v) and dev is not None else v)
    kw = dict(clip_feature=mv(ins["clip_feature"]), y=mv(ins["y"]), plucker_fea=mv(ins["plucker_fea"]),
              plucker_context_lens=mv(ins["plucker_context_lens"]), uncond=case.uncond)
    if ins.get("control_camera_latents_input") is not None:
        kw["control_camera_latents_input"] = mv(ins["control_camera_latents_input"])
    if case.golden["meta"].get("camera_token"):
        kw["camera_token"] = mv(ins["camera_token"])
    return kw


class HeadsCase:
    """Geometry-head golden case: reduced-width heads, weights and tokens regenerated from seeds, the
    REAL reference's prediction dict from disk (oracle/make_golden.py)."""

    def __init__(self, name):
        from fantasy_world_amd import config as fwc, synth
        self.name = name
        self.golden = load_golden(name)
        meta = self.golden["meta"]
        self.hc = fwc.HeadsConfig() if meta.get("heads") == "full" else fwc.HeadsConfig.small()
        self.S, self.ph, self.pw = meta["grid"]
        self.weights = BF16Weights(synth.make_heads_weights(self.hc, seed=meta["seed_weights"]))
        self.output_list = synth.make_output_list(self.hc, self.S, self.ph, self.pw, seed=meta["seed_tokens"])


@pytest.fixture(scope="session", params=["heads_small_s3_4x6", "heads_small_s2_5x3"])
def heads_case(request):
    return HeadsCase(request.param)


@pytest.fixture(scope="session")
def heads_case_full():
    """The reference's real head widths (623 M parameters) on a 2x3 token grid."""
    return HeadsCase("heads_full_s2_2x3")


PRED_KEYS = ("pose_enc", "depth", "depth_conf", "world_points", "world_points_conf")


class PredCase(Case):
    """End-to-end case with return_prediction=True: the fusion model of a Case plus narrow geometry heads
    (HeadsConfig.e2e_small()); golden = noise_pred and the prediction dict of the REAL reference."""

    def __init__(self, name):
        from fantasy_world_amd import config as fwc, synth
        super().__init__(name)
        self.hc = fwc.HeadsConfig.e2e_small()
        self.weights.update(synth.make_heads_weights(self.hc, seed=self.golden["meta"]["seed_weights"]))


@pytest.fixture(scope="session")
def case_pred():
    return PredCase("wan21_pred_l3_f2_8x12")


class PoseCase:
    """CameraPoseEncoder golden: real widths, weights and Pluecker embedding regenerated from seeds, the
    reference module's plucker_fea from disk."""

    def __init__(self, name):
        from fantasy_world_amd import synth
        self.name = name
        self.golden = load_golden(name)
        meta = self.golden["meta"]
        self.grid = meta["grid"]
        self.weights = synth.make_pose_encoder_weights(seed=meta["seed_weights"])
        self.plucker = synth.make_plucker(*self.grid, seed=meta["seed_plucker"])


@pytest.fixture(scope="session", params=["pose_full_f9_32x48", "pose_full_f13_48x16"])
def pose_case(request):
    return PoseCase(request.param)


class VaeCase:
    """Wan VAE decoder golden: real widths, weights and latents regenerated from seeds, the reference's
    VideoVAE_.decode output from disk."""

    def __init__(self, name):
        from fantasy_world_amd import synth
        self.name = name
        self.golden = load_golden(name)
        meta = self.golden["meta"]
        self.grid = meta["grid"]
        self.weights = synth.make_vae_decoder_weights(seed=meta["seed_weights"])
        self.latents = synth.make_latents(*self.grid, seed=meta["seed_latents"])


@pytest.fixture(scope="session", params=["vae_full_t3_4x6", "vae_full_t2_3x5"])
def vae_case(request):
    return VaeCase(request.param)


@pytest.fixture(autouse=True)
def _collect_cycles_after_each_test():
    """install() ties model -> rebound method -> closure -> engine -> modules into a reference cycle, and the cyclic collector
    triggers on object COUNTS, not bytes: a few multi-GB tensors in a cycle stay resident for many tests.  Without this the CPU suite's
    resident set ratchets 11 -> 56 GB across the install / engine tests (profiles/r03/pytest_cpu_rss_trace_before_gc_fixture.tsv) in a
    62 GB container and ran into the OOM killer once."""
    yield
    import gc
    gc.collect()


# FW_TEST_RSS_LOG=<file>: append "nodeid <tab> RSS now (GB) <tab> peak RSS so far (GB)" after every test -- where the CPU suite's memory
# goes (the build container has 62 GB; the suite once ran into the OOM killer).  Off by default.
def pytest_runtest_teardown(item, nextitem):
    path = os.environ.get("FW_TEST_RSS_LOG")
    if not path:
        return
    import resource
    now = 0.0
    try:
        with open("/proc/self/statm") as f:
            now = int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 1e9
    except OSError:
        pass
    peak = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6
    with open(path, "a") as f:
        f.write(f"{item.nodeid}\t{now:.1f}\t{peak:.1f}\n")
