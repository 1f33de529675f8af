#!/usr/bin/env python3
"""Which torch-dispatched kernels does one training step launch, and from which line of the product?
Runs a few steps of the bench model under a TorchDispatchMode that records every aten op touching a GPU tensor
(views excluded) with the innermost frame inside tec-mollm_amd/ -- the tecm_* launches go through ctypes and do not
appear here at all.  PRECISION=bf16|fp32, BATCH=8."""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tec-mollm_amd"))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from src.model.tec_mollm import TEC_MoLLM
from tecmollm.synthetic import grid_graph, synthetic_batch
from tecmollm.train import TrainStep
import bench

VIEW = ("view", "slice", "select", "expand", "unsqueeze", "squeeze", "t.default", "transpose", "permute", "detach", "alias",
        "as_strided", "reshape", "_unsafe_view", "unbind", "split", "narrow", "empty", "_local_scalar", "is_", "size",
        "stride", "sym_", "numel", "dim", "lift_fresh", "new_empty", "resize_", "set_", "_reshape_alias", "unflatten", "flatten")


class Rec(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.hits = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        out = func(*args, **(kwargs or {}))
        short = name.replace("aten.", "")
        if any(short.startswith(v) for v in VIEW):
            return out
        flat = [a for a in list(args) + list((kwargs or {}).values()) if isinstance(a, torch.Tensor)]
        if isinstance(out, torch.Tensor):
            flat.append(out)
        if not any(t.is_cuda for t in flat):
            return out
        where = "?"
        for fr in reversed(traceback.extract_stack()):
            if "tec-mollm_amd" in fr.filename or fr.filename.endswith("bench.py"):
                where = f"{os.path.basename(fr.filename)}:{fr.lineno}"
                break
        self.hits[(short, where)] += 1
        return out


prec = os.environ.get("PRECISION", "bf16")
args = bench.parse.__globals__["argparse"].Namespace(L_in=48, L_out=12, c_in=10, batch=int(os.environ.get("BATCH", 8)))
cfg = bench.make_config(args)
dev = torch.device("cuda")
torch.manual_seed(0)
model = TEC_MoLLM(dict(cfg, gat_graphs="per_timestep", include_wte=False, load_pretrained_gpt2=False, precision=prec)).to(dev).train()
B = args.batch
x, tf, y = synthetic_batch(B, 48, 2911, 10, 12, seed=1)
x, y = x.to(dev), y.to(dev)
tf = tf[:, :, 0, :].contiguous().to(dev).unsqueeze(-2).expand(B, 48, 2911, 4)
ei, ew = grid_graph()
ei, ew = ei.to(dev), ew.to(dev)
ts = TrainStep(model, world_size=1)
for _ in range(2):
    ts.step(x, tf, ei, ew, y)
torch.cuda.synchronize()
STEPS = 2
with Rec() as rec:
    for _ in range(STEPS):
        ts.step(x, tf, ei, ew, y)
torch.cuda.synchronize()
tot = 0
for (op, where), n in sorted(rec.hits.items(), key=lambda kv: (-kv[1], kv[0])):
    print(f"{n / STEPS:6.1f}/step  {op:40s} {where}")
    tot += n
print(f"total {tot / STEPS:.1f} torch-dispatched device ops per step ({prec}, B={B})")
