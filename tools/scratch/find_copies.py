"""Which Python lines of one bf16 training step cause torch-dispatched device copies / fills (the ~18 tiny copyBuffer launches)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tec-mollm_amd")):
    sys.path.insert(0, p)
import torch
sys.argv = [sys.argv[0]]
import tools.scratch.graph_try as G     # reuses make() (runs nothing: which = none)
from torch.profiler import profile, ProfilerActivity
m, x, tf, y = G.make(8, "bf16")
ts = G.TrainStep(m, world_size=1)
for _ in range(3): ts.step(x, tf, G.ei, G.ew, y)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    ts.step(x, tf, G.ei, G.ew, y)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::fill_", "aten::zero_", "aten::add_", "aten::_foreach_add_", "aten::mul", "aten::sum", "aten::to", "aten::_to_copy"):
        st = [f for f in (ev.stack or []) if "tec-mollm_amd" in f or "tecmollm" in f or "bench" in f]
        shapes = str(ev.input_shapes)[:60]
        cnt[(ev.name, st[0] if st else (ev.stack[0] if ev.stack else "?"), shapes)] += 1
for (name, where, shapes), n in sorted(cnt.items(), key=lambda kv: -kv[1])[:60]:
    print(f"{n:3d} {name:22s} {shapes:60s} {where}")
