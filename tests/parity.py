"""Parity helpers shared by the GPU tests, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg.
Lives under tests/ because it imports the oracle (test infrastructure); the product package never does."""
from __future__ import annotations

import os
import sys
from typing import Dict, Optional, Tuple

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tec-mollm_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from oracle import ref_cpu as R  # noqa: E402

RTOL = 1e-3          # north star: "within 1e-3 rel fp32"
# Absolute slack of the element-wise bar, as a fraction of the oracle tensor's RMS.  Not tighter than 5e-3: GATv2's
# LeakyReLU has a kink -- when a sum x_l[j,c] + x_r[i,c] lands within rounding distance of 0 the two sides can see
# opposite signs, lrelu' jumps between 0.2 and 1, and ONE term of a gradient sum changes (observed at N = 2911, 48
# graphs: two neighbouring rows of the node-table gradient off by 1.8e-3 of its RMS, every other row below 2e-5).
ATOL_RMS = 5e-3


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max|a-b| / max|b|  (b = oracle): the global, max-norm relative error."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def elem_err(a: torch.Tensor, b: torch.Tensor, rtol: float = RTOL, atol_rms: float = ATOL_RMS) -> float:
    """Element-wise bar: max over elements of |a-b| / (rtol*|b| + atol), atol = atol_rms * rms(b).
    A value < 1 means EVERY element satisfies |a-b| <= rtol*|b| + atol (the torch.allclose form with the absolute
    term tied to the tensor's own scale), so a small entry that is wrong by 100 % fails even when the tensor's
    largest entry hides it from `rel_err`.  b = oracle."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    rms = float(b.pow(2).mean().sqrt())
    if rms == 0.0:
        return 0.0 if float((a - b).abs().max()) == 0.0 else float("inf")
    return float(((a - b).abs() / (rtol * b.abs() + atol_rms * rms)).max())


def build_model(cfg: dict, params: Dict[str, torch.Tensor], device, gat_graphs: str = "reference"):
    """TEC_MoLLM (HIP path) carrying exactly the oracle's parameters (strict state-dict load)."""
    from src.model.tec_mollm import TEC_MoLLM
    mc = dict(cfg)
    mc.update(gat_graphs=gat_graphs, include_wte=False, load_pretrained_gpt2=False)
    model = TEC_MoLLM(mc)
    missing, unexpected = model.load_state_dict(params, strict=True), None
    del missing, unexpected
    return model.to(device)


def oracle_step(cfg, params, x, tf, ei, y, graphs_with_edges,
                masks=None) -> Tuple[torch.Tensor, torch.Tensor, Dict[str, torch.Tensor]]:
    """CPU oracle forward + Huber loss + autograd backward.  Returns (out, loss, grads of trainable params)."""
    p = {k: v.clone().requires_grad_(R.is_trainable(k)) for k, v in params.items()}
    out = R.forward(x, tf, ei, p, cfg, graphs_with_edges, masks=masks)
    loss = R.huber(out, y)
    names = [k for k, v in p.items() if v.requires_grad]
    grads = torch.autograd.grad(loss, [p[k] for k in names], allow_unused=True)
    return out.detach(), loss.detach(), {k: (g if g is not None else torch.zeros_like(p[k])) for k, g in zip(names, grads)}


# ------------------------------------------------------------------------------------ mirrored dropout masks
def _tm_to_seq(m: np.ndarray, B: int, T: int, N: int) -> torch.Tensor:
    """(B*T*N, D) time-major rows m = (b*T + t)*N + n  ->  the oracle's (S = B*N, T, D) with s = b*N + n."""
    D = m.shape[1]
    return torch.from_numpy(np.ascontiguousarray(m.reshape(B, T, N, D).transpose(0, 2, 1, 3).reshape(B * N, T, D)))


def device_masks(cfg: dict, B: int, ei: torch.Tensor, base_seed: int, gat_graphs: str, p: float = 0.1) -> dict:
    """The dropout multipliers (0 | 1/(1-p)) the HIP path applies in ONE training-mode forward whose plan has
    `base_seed`, at every site of the reference, laid out the way oracle.ref_cpu.forward(masks=...) wants them.
    Built from tecmollm/rng.py (the bit-for-bit NumPy mirror of csrc/common.h:tecm_hash24) and the site / index
    conventions of tecmollm/functions.py:
        seed(site) = splitmix64(base_seed * 1000003 + site);   idx = logical_row * ld + col
    GATv2 alpha: row = (t*B + b)*N + i, idx = (row*H + head)*(max_deg + 1) + slot, slot = position of the edge in the
    target's in-edge list (self loops removed, given order kept), the implicit self loop last."""
    from tecmollm import functions as F_
    from tecmollm import graph as G_
    from tecmollm import ops, rng
    N, L = cfg["num_nodes"], cfg["temporal_seq_len"]
    H = cfg["spatial_heads"]
    D = cfg["d_llm"]
    T = (L // (cfg["temporal_strides"][0] * cfg["temporal_strides"][1])) // cfg["patch_len"]
    S = B * N
    Hd = (D * T) // 4
    n_layers = cfg["llm_layers"]

    def seed(site):
        return ops.splitmix64(base_seed * 1000003 + site)

    def rows(site, nrows, ncols, ld=None):
        ld = ncols if ld is None else ld
        idx = (np.arange(nrows, dtype=np.uint64)[:, None] * np.uint64(ld) + np.arange(ncols, dtype=np.uint64)[None, :])
        return rng.keep_mult(seed(site), idx, p)

    masks = {}
    # ---- GATv2 attention coefficients (modules.py:333)
    ein = ei.numpy()
    rowptr, colidx = G_.csr_by_target(ein, N)
    deg = np.diff(rowptr).astype(np.int64)
    ld = int(deg.max() if deg.size else 0) + 1
    Gtot = L * B
    Gedges = 1 if gat_graphs == "reference" else Gtot
    src, dst = ein[0].astype(np.int64), ein[1].astype(np.int64)
    nonself = src != dst
    # slot of every given non-self edge inside its target's list (stable order)
    order = np.argsort(dst[nonself], kind="stable")
    slot_sorted = np.arange(order.size, dtype=np.int64) - np.repeat(rowptr[:-1].astype(np.int64), deg)
    slot = np.empty(order.size, dtype=np.int64)
    slot[order] = slot_sorted
    d_ns = dst[nonself]
    # oracle edge order: batched_edge_index -> (graph g, edge e) at g*E + e, self loops of the INPUT removed, then one
    # self loop per row 0..M-1
    g_ids = np.arange(Gedges, dtype=np.int64)
    row_e = (g_ids[:, None] * N + d_ns[None, :]).reshape(-1)
    slot_e = np.tile(slot, Gedges)
    M = Gtot * N
    row_l = np.arange(M, dtype=np.int64)
    node_l = row_l % N
    slot_l = np.where(row_l // N < Gedges, deg[node_l], 0)
    row_all = np.concatenate([row_e, row_l]).astype(np.uint64)
    slot_all = np.concatenate([slot_e, slot_l]).astype(np.uint64)
    hh = np.arange(H, dtype=np.uint64)[None, :]
    idx = (row_all[:, None] * np.uint64(H) + hh) * np.uint64(ld) + slot_all[:, None]
    masks["gat"] = torch.from_numpy(rng.keep_mult(seed(F_.SITE_GAT), idx, p))
    # ---- GPT-2 front end and blocks (time-major rows)
    Mtok = B * T * N
    masks["embd"] = _tm_to_seq(rows(F_.SITE_EMBD, Mtok, D), B, T, N)
    KE = D + F_.LORA_R
    for i in range(n_layers):
        masks[f"lora{i}"] = _tm_to_seq(rows(F_.site_lora(i), Mtok, D, ld=KE), B, T, N)
        aidx = np.arange(S * F_.GPT_HEADS * T * T, dtype=np.uint64)
        masks[f"attn{i}"] = torch.from_numpy(rng.keep_mult(seed(F_.site_attn(i)), aidx, p)).view(S, F_.GPT_HEADS, T, T)
        masks[f"res1_{i}"] = _tm_to_seq(rows(F_.site_res1(i), Mtok, D), B, T, N)
        masks[f"res2_{i}"] = _tm_to_seq(rows(F_.site_res2(i), Mtok, D), B, T, N)
    masks["post"] = _tm_to_seq(rows(F_.SITE_POST, Mtok, D), B, T, N)
    masks["head"] = torch.from_numpy(rows(F_.SITE_HEAD, S, Hd))
    return masks


def compare_forward_backward(cfg: dict, B: int, grid: Tuple[int, int], threshold_km: float = 150.0,
                             gat_graphs: str = "reference", seed: int = 0, use_fused_huber: bool = True,
                             device: Optional[str] = None, train: bool = False) -> dict:
    """Run the same seeded step through the CPU oracle and the HIP model and report the errors of the forward output,
    the loss and every trainable gradient: `*_rel` = max-norm relative error, `*_elem` = the element-wise bar
    (`elem_err`, < 1 passes).  train=False: eval mode (dropout off).  train=True: training mode, every dropout site of
    the reference active (p = 0.1); the oracle receives the NumPy mirror of the device's counter-based masks
    (`device_masks`), so both sides drop exactly the same elements."""
    from src.model import modules as M_
    from tecmollm import functions as F_
    device = device or "cuda"
    N = grid[0] * grid[1]
    assert N == cfg["num_nodes"]
    params = R.init_params(cfg, seed=seed)
    x, tf, y = R.synthetic_batch(B, cfg["temporal_seq_len"], N, cfg["spatial_in_channels_base"],
                                 cfg["prediction_horizon"], seed=seed + 100)
    ei, ew = R.grid_graph(grid[0], grid[1], threshold_km=threshold_km)
    gwe = 1 if gat_graphs == "reference" else None
    masks = None
    if train:
        torch.manual_seed(4242 + seed)
        M_._seed_counter[0] = 17                                  # make_plan: base = initial_seed + 7919 * (count + 1)
        base_seed = torch.initial_seed() + 7919 * 18
        masks = device_masks(cfg, B, ei, base_seed, gat_graphs)
    out_ref, loss_ref, grads_ref = oracle_step(cfg, params, x, tf, ei, y, gwe, masks)

    model = build_model(cfg, params, device, gat_graphs)
    model.train(train)
    xd, yd = x.to(device), y.to(device)
    tfd = tf[:, :, 0, :].contiguous().to(device).unsqueeze(-2).expand(B, cfg["temporal_seq_len"], N, 4)
    eid, ewd = ei.to(device), ew.to(device)
    out = model(xd, tfd, eid, ewd)
    loss = F_.HuberFn.apply(out, yd, 1.0) if use_fused_huber else torch.nn.functional.huber_loss(out, yd, delta=1.0)
    loss.backward()
    torch.cuda.synchronize()
    res = {"fwd_rel": rel_err(out, out_ref), "fwd_elem": elem_err(out, out_ref),
           "loss_rel": abs(loss.item() - loss_ref.item()) / abs(loss_ref.item())}
    worst, worst_name = 0.0, ""
    worst_e, worst_e_name = 0.0, ""
    per = {}
    named = dict(model.named_parameters())
    for k, gref in grads_ref.items():
        g = named[k].grad
        assert g is not None, f"no gradient for trainable parameter {k}"
        nz = gref.abs().max() > 0
        e = rel_err(g, gref) if nz else float(g.abs().max())
        ee = elem_err(g, gref) if nz else float(g.abs().max())
        per[k] = (e, ee)
        if e > worst:
            worst, worst_name = e, k
        if ee > worst_e:
            worst_e, worst_e_name = ee, k
    frozen_with_grad = [k for k, p in named.items() if not R.is_trainable(k) and p.grad is not None]
    res.update(grad_rel_max=worst, grad_worst=worst_name, grad_elem_max=worst_e, grad_elem_worst=worst_e_name,
               n_grads=len(per), frozen_with_grad=frozen_with_grad)
    res["per_param"] = per
    return res


def assert_parity(res: dict, tol: float = RTOL) -> None:
    """The bar of every full-step test: forward, loss and all gradients within 1e-3 in the max norm AND element-wise."""
    brief = {k: v for k, v in res.items() if k != "per_param"}
    assert res["fwd_rel"] < tol and res["loss_rel"] < tol, brief
    assert res["grad_rel_max"] < tol, brief
    assert res["fwd_elem"] < 1.0, brief
    assert res["grad_elem_max"] < 1.0, brief
    assert res["frozen_with_grad"] == [], brief
