"""Op-by-op trace: first op whose sample-1 half differs between the merged CFG pass and the separate negative pass."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fantasy_world_amd import config as fwc, synth
from fantasy_world_amd.engine import FusionEngine
from fantasy_world_amd.hip_ops import HipOps
cfg = fwc.plumbing(2, 1)
ops = HipOps("cuda:0")
spec = synth.weight_spec(cfg)
eng = FusionEngine(cfg, lambda n: synth.make_param(n, spec[n][0], spec[n][1], device="cuda:0"), ops)
ins = synth.make_inputs(cfg, 9, 64, 64, seed=1, device="cuda:0", dtype=torch.float32)
kw = dict(clip_feature=ins["clip_feature"], y=ins["y"], plucker_fea=ins["plucker_fea"], plucker_context_lens=ins["plucker_context_lens"])
trace = []
MODE = {"halves": False}
def h(t):
    t = t.contiguous()
    return int(t.view(torch.int16 if t.element_size() == 2 else torch.int32).to(torch.int64).sum())
def wrap(name):
    fn = getattr(ops, name)
    def w(*a, **k):
        out = fn(*a, **k)
        o = out[0] if isinstance(out, tuple) else out
        if torch.is_tensor(o) and o.dim() == 2:
            rows = o.shape[0]
            if MODE["halves"] and rows in (2 * 9216, 2 * 9261, 1024, 514):
                trace.append((name, rows // 2, tuple(o.shape[1:]), str(o.dtype), h(o[rows // 2:]), h(o[:rows // 2])))
            else:
                trace.append((name, rows, tuple(o.shape[1:]), str(o.dtype), h(o), None))
        return out
    setattr(ops, name, w)
for n in ["linear", "attention", "layernorm", "qk_prep", "cast_act", "assemble_tokens", "rmsnorm", "to_act"]:
    if hasattr(ops, n):
        wrap(n)
eng.joint_forward(ins["x"], ins["timestep"], ins["context_neg"], **kw)
torch.cuda.synchronize()
single = list(trace); trace.clear()
MODE["halves"] = True
eng._forward(ins["x"], ins["timestep"], [ins["context"], ins["context_neg"]], ins["clip_feature"], ins["y"], ins["plucker_fea"],
             ins["plucker_context_lens"], False, False, None, None, None)
torch.cuda.synchronize()
pair = list(trace)
print(len(single), len(pair))
# align greedily: walk `pair`, match to the next `single` entry with the same (name, rows, cols, dtype)
j = 0
shown = 0
for i, p in enumerate(pair):
    k = j
    while k < len(single) and single[k][:4] != p[:4]:
        k += 1
    if k == len(single):
        continue
    j = k + 1
    ok = single[k][4] == p[4]
    if not ok or os.environ.get('ALL'):
        print(f"pair#{i} single#{k} {p[:4]} sample1-equal={ok}")
        shown += 1
        if shown > 12 and not os.environ.get('ALL'):
            break
print("done")
