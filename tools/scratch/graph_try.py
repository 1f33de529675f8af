"""step_graphed vs step: does it record, is it equivalent with dropout off, how fast is it (B = 1, 2, 8; bf16 / fp32)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tec-mollm_amd")):
    sys.path.insert(0, p)
import torch
from src.model.tec_mollm import TEC_MoLLM
from tecmollm.synthetic import grid_graph, synthetic_batch
from tecmollm.train import TrainStep
import bench

dev = torch.device("cuda")
ei, ew = grid_graph(); ei, ew = ei.to(dev), ew.to(dev)
args = bench.parse() if False else None

def make(B, mode, train=True, seed=0):
    import argparse
    a = argparse.Namespace(L_in=48, L_out=12, c_in=10, llm_layers=3, gat="per_timestep")
    cfg = bench.make_config(a)
    mc = dict(cfg, gat_graphs="per_timestep", include_wte=False, load_pretrained_gpt2=False, precision=mode)
    torch.manual_seed(seed)
    m = TEC_MoLLM(mc)
    with torch.no_grad():
        for blk in m.llm_backbone.trunk.h:
            blk.attn.c_attn.lora_B.default.weight.normal_(std=0.02)
    m = m.to(dev); m.train(train)
    x, tf, y = synthetic_batch(B, 48, 2911, 10, 12, seed=1234)
    x, y = x.to(dev), y.to(dev)
    tf = tf[:, :, 0, :].contiguous().to(dev).unsqueeze(-2).expand(B, 48, 2911, 4)
    return m, x, tf, y

def timeit(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "equiv"):
    # equivalence with dropout off: 4 steps eager vs 4 steps graphed from the same init
    outs = []
    for graphed in (False, True):
        m, x, tf, y = make(2, "fp32", train=False)
        ts = TrainStep(m, world_size=1)
        f = ts.step_graphed if graphed else ts.step
        losses = [float(f(x, tf, ei, ew, y)) for _ in range(4)]
        outs.append((losses, ts.optimizer.flat_param.clone()))
        print("graphed" if graphed else "eager", losses, flush=True)
    d = (outs[0][1] - outs[1][1]).abs().max().item()
    print("max param diff after 4 steps:", d, flush=True)
if which in ("all", "time"):
    for mode in ("bf16", "fp32"):
        for B in (1, 2, 8):
            m, x, tf, y = make(B, mode)
            ts = TrainStep(m, world_size=1)
            for _ in range(3): ts.step(x, tf, ei, ew, y)
            n = 20 if mode == "bf16" else 8
            te = timeit(lambda: ts.step(x, tf, ei, ew, y), n)
            for _ in range(3): ts.step_graphed(x, tf, ei, ew, y)
            tg = timeit(lambda: ts.step_graphed(x, tf, ei, ew, y), n)
            print(f"{mode} B={B}: eager {te:.3f} ms  graphed {tg:.3f} ms  ({te / tg:.3f}x)", flush=True)
            del ts, m; torch.cuda.empty_cache()
