"""The CPU oracle (oracle/fw_oracle.py) must reproduce the golden fixtures, which are outputs of the REAL reference
(oracle/make_golden.py, run in the build container where /root/reference is mounted)."""
import pytest

from conftest import rel_l2
from oracle import fw_oracle


@pytest.mark.parametrize("case_name", ["case_l2", "case_l3", "case_w22"])
def test_oracle_matches_reference_golden(case_name, request):
    case = request.getfixturevalue(case_name)
    ins = case.inputs
    col = {}
    out = fw_oracle.joint_forward(case.weights, case.cfg, ins["x"], ins["timestep"], ins["context"],
                                  ins["clip_feature"], ins["y"], ins["plucker_fea"], ins["plucker_context_lens"],
                                  uncond=case.uncond, collect=col,
                                  control_camera_latents_input=ins.get("control_camera_latents_input"))
    col["noise_pred"] = out
    for k in ("x_after_pcb", "x_final", "tokens_final", "noise_pred"):
        err = rel_l2(col[k], case.golden[k])
        assert err < 5e-6, f"{case.name}:{k} oracle deviates from the reference golden: rel-L2 {err:.3e}"
