"""Parity helpers shared by the GPU tests, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg.
Lives under tests/ because it imports the oracle (test infrastructure); the product package never does."""
from __future__ import annotations

import os
import sys
from typing import Dict, Optional, Tuple

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tec-mollm_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from oracle import ref_cpu as R  # noqa: E402


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max|a-b| / max|b|  (b = oracle)."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def build_model(cfg: dict, params: Dict[str, torch.Tensor], device, gat_graphs: str = "reference"):
    """TEC_MoLLM (HIP path) carrying exactly the oracle's parameters (strict state-dict load)."""
    from src.model.tec_mollm import TEC_MoLLM
    mc = dict(cfg)
    mc.update(gat_graphs=gat_graphs, include_wte=False, load_pretrained_gpt2=False)
    model = TEC_MoLLM(mc)
    missing, unexpected = model.load_state_dict(params, strict=True), None
    del missing, unexpected
    return model.to(device)


def oracle_step(cfg, params, x, tf, ei, y, graphs_with_edges) -> Tuple[torch.Tensor, torch.Tensor, Dict[str, torch.Tensor]]:
    """CPU oracle forward + Huber loss + autograd backward.  Returns (out, loss, grads of trainable params)."""
    p = {k: v.clone().requires_grad_(R.is_trainable(k)) for k, v in params.items()}
    out = R.forward(x, tf, ei, p, cfg, graphs_with_edges)
    loss = R.huber(out, y)
    names = [k for k, v in p.items() if v.requires_grad]
    grads = torch.autograd.grad(loss, [p[k] for k in names], allow_unused=True)
    return out.detach(), loss.detach(), {k: (g if g is not None else torch.zeros_like(p[k])) for k, g in zip(names, grads)}


def compare_forward_backward(cfg: dict, B: int, grid: Tuple[int, int], threshold_km: float = 150.0,
                             gat_graphs: str = "reference", seed: int = 0, use_fused_huber: bool = True,
                             device: Optional[str] = None) -> dict:
    """Run the same seeded step through the CPU oracle and the HIP model (eval mode: dropout off) and
    report max relative errors of the forward output, the loss and every trainable gradient."""
    from tecmollm import functions as F_
    device = device or "cuda"
    N = grid[0] * grid[1]
    assert N == cfg["num_nodes"]
    params = R.init_params(cfg, seed=seed)
    x, tf, y = R.synthetic_batch(B, cfg["temporal_seq_len"], N, cfg["spatial_in_channels_base"],
                                 cfg["prediction_horizon"], seed=seed + 100)
    ei, ew = R.grid_graph(grid[0], grid[1], threshold_km=threshold_km)
    gwe = 1 if gat_graphs == "reference" else None
    out_ref, loss_ref, grads_ref = oracle_step(cfg, params, x, tf, ei, y, gwe)

    model = build_model(cfg, params, device, gat_graphs).eval()
    xd, yd = x.to(device), y.to(device)
    tfd = tf[:, :, 0, :].contiguous().to(device).unsqueeze(-2).expand(B, cfg["temporal_seq_len"], N, 4)
    eid, ewd = ei.to(device), ew.to(device)
    out = model(xd, tfd, eid, ewd)
    loss = F_.HuberFn.apply(out, yd, 1.0) if use_fused_huber else torch.nn.functional.huber_loss(out, yd, delta=1.0)
    loss.backward()
    torch.cuda.synchronize()
    res = {"fwd_rel": rel_err(out, out_ref), "loss_rel": abs(loss.item() - loss_ref.item()) / abs(loss_ref.item())}
    worst, worst_name = 0.0, ""
    per = {}
    named = dict(model.named_parameters())
    for k, gref in grads_ref.items():
        g = named[k].grad
        assert g is not None, f"no gradient for trainable parameter {k}"
        e = rel_err(g, gref) if gref.abs().max() > 0 else float(g.abs().max())
        per[k] = e
        if e > worst:
            worst, worst_name = e, k
    frozen_with_grad = [k for k, p in named.items() if not R.is_trainable(k) and p.grad is not None]
    res.update(grad_rel_max=worst, grad_worst=worst_name, n_grads=len(per), frozen_with_grad=frozen_with_grad)
    res["per_param"] = per
    return res
