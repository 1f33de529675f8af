"""CPU oracle for the shell around the model step.

TEST INFRASTRUCTURE ONLY -- imported by tests/ (and nothing in the product path).

  * `evaluate_horizons` / `evaluate_metrics` restate `/root/reference/src/evaluation/metrics.py:10-89,
    :119-183` in plain numpy (no sklearn / scipy): pinned by tests/golden/shell_metrics*.npz, which
    oracle/make_golden_shell.py produced by calling the reference's own functions.
  * `SlidingWindows` restates `SlidingWindowSamplerDataset` (`/root/reference/src/data/dataset.py:10-99`)
    on in-memory arrays plus the harness reshapes of train.py:62-65,:76; pinned by
    tests/golden/shell_windows.npz (items drawn from the reference class itself).
  * `reference_optimizer_steps` is not a restatement: it RUNS what train.py:94-109,:358-366 calls --
    torch.nn.utils.clip_grad_norm_, torch.optim.AdamW, CosineAnnealingWarmRestarts -- on CPU tensors.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch


# ---------------------------------------------------------------------------- metrics.py
def _inverse_transform_f32(y: np.ndarray, mean: float, scale: float) -> np.ndarray:
    """sklearn StandardScaler.inverse_transform on float32 input (metrics.py:36-37): `X *= scale_; X += mean_`
    in place on a float32 copy -> each step computed in float64 and rounded to float32."""
    out = np.array(y, dtype=np.float32, copy=True)
    out *= np.float64(scale)
    out += np.float64(mean)
    return out


def _nan_to_num(a: np.ndarray) -> np.ndarray:
    if not np.all(np.isfinite(a)):                                   # metrics.py:40-46
        a = np.nan_to_num(a, nan=0.0, posinf=100.0, neginf=0.0)
    return a


def _metrics_from_unscaled(t: np.ndarray, p: np.ndarray) -> Dict[str, float]:
    """MAE / RMSE / R^2 (sklearn semantics incl. force_finite) / Pearson r on the flattened arrays
    (metrics.py:53-78; with one output column the per-column and the flat forms coincide)."""
    t = t.reshape(-1).astype(np.float64)
    p = p.reshape(-1).astype(np.float64)
    d = t - p
    mae = float(np.mean(np.abs(d)))
    rmse = float(np.sqrt(np.mean(d * d)))
    ss_res = float(np.sum(d * d))
    ss_tot = float(np.sum((t - t.mean()) ** 2))
    if ss_tot != 0.0:
        r2 = 1.0 - ss_res / ss_tot
    else:
        r2 = 1.0 if ss_res == 0.0 else 0.0                           # sklearn r2_score(force_finite=True)
    if np.std(t) > 0 and np.std(p) > 0:
        tc, pc = t - t.mean(), p - p.mean()
        pear = float(np.sum(tc * pc) / np.sqrt(np.sum(tc * tc) * np.sum(pc * pc)))
        pear = max(-1.0, min(1.0, pear))                             # scipy clamps to [-1, 1]
    else:
        pear = 0.0
    return {"mae": mae, "rmse": rmse, "r2_score": r2, "pearson_r": pear}


def evaluate_metrics(y_true_scaled: np.ndarray, y_pred_scaled: np.ndarray, mean: float, scale: float) -> Dict[str, float]:
    """metrics.py:10-87 with the fitted StandardScaler given as (mean_[0], scale_[0])."""
    t = _nan_to_num(_inverse_transform_f32(y_true_scaled, mean, scale))
    p = _nan_to_num(_inverse_transform_f32(y_pred_scaled, mean, scale))
    p = np.clip(p, 0, 200)                                            # metrics.py:50-51
    return _metrics_from_unscaled(t, p)


def evaluate_horizons(y_true: np.ndarray, y_pred: np.ndarray, mean: Optional[float] = None,
                      scale: Optional[float] = None) -> Dict[str, object]:
    """metrics.py:119-183.  y_* are (S, L_out, ...) scaled arrays; (mean, scale) None = the no-scaler
    fallback (:89-117), which neither inverse-transforms nor clips."""
    if not np.all(np.isfinite(y_pred)):                              # :139-145
        y_pred = np.nan_to_num(y_pred, nan=0.0, posinf=0.0, neginf=0.0)
    per: List[Dict[str, float]] = []
    for h in range(y_true.shape[1]):
        if mean is not None:
            per.append(evaluate_metrics(y_true[:, h], y_pred[:, h], mean, scale))
        else:
            per.append(_metrics_from_unscaled(np.asarray(y_true[:, h]), np.asarray(y_pred[:, h])))
    return {
        "mae_avg": float(np.mean([m["mae"] for m in per])),
        "rmse_avg": float(np.mean([m["rmse"] for m in per])),
        "r2_score_avg": float(np.mean([m["r2_score"] for m in per])),
        "pearson_r_avg": float(np.mean([m["pearson_r"] for m in per])),
        "mae_by_horizon": [m["mae"] for m in per],
        "rmse_by_horizon": [m["rmse"] for m in per],
        "r2_by_horizon": [m["r2_score"] for m in per],
        "pearson_by_horizon": [m["pearson_r"] for m in per],
    }


# ---------------------------------------------------------------------------- dataset.py
class SlidingWindows:
    """SlidingWindowSamplerDataset (dataset.py:10-99) over in-memory arrays X (T,H,W,C), Y (T,H,W,L_out),
    time_features (T,F)."""

    def __init__(self, X, Y, time_features, L_in: int = 336, L_out: int = 12, stride: int = 1):
        self.X, self.Y, self.tf = X, Y, time_features
        self.L_in, self.L_out, self.stride = L_in, L_out, stride
        max_start = len(X) - L_in - L_out + 1                         # dataset.py:47
        self.sample_indices = list(range(0, max_start, stride)) if max_start > 0 else []

    def __len__(self) -> int:
        return len(self.sample_indices)

    def __getitem__(self, idx: int):
        if idx >= len(self):
            raise IndexError(idx)
        a = self.sample_indices[idx]
        return {"x": self.X[a:a + self.L_in], "y": self.Y[a + self.L_in - 1],   # dataset.py:80-92
                "x_time_features": self.tf[a:a + self.L_in]}

    def batch(self, indices: Sequence[int]):
        """Collate + the harness reshapes: x (B,L,N,C), time features (B,L,N,F) expanded, target
        (B,L_out,N,1) (train.py:62-65, :76)."""
        items = [self[i] for i in indices]
        x = np.stack([np.asarray(it["x"]) for it in items])
        y = np.stack([np.asarray(it["y"]) for it in items])
        tf = np.stack([np.asarray(it["x_time_features"]) for it in items])
        B, L, H, W, C = x.shape
        x = x.reshape(B, L, H * W, C)
        tf = np.broadcast_to(tf[:, :, None, :], (B, L, H * W, tf.shape[-1]))
        y = np.transpose(y, (0, 3, 1, 2)).reshape(B, -1, H * W, 1)
        return x, tf, y


# ---------------------------------------------------------------------------- train.py:94-109, :358-366
def reference_optimizer_steps(params: List[torch.Tensor], grads_per_step: List[List[torch.Tensor]], lr: float = 1e-4,
                              weight_decay: float = 1e-2, max_norm: float = 1.0):
    """Run clip_grad_norm_ -> AdamW.step -> zero_grad -> CosineAnnealingWarmRestarts.step for each entry of
    grads_per_step on CPU copies of `params`.  Returns (final params, [total_norm per step], [lr used per step])."""
    ps = [torch.nn.Parameter(p.detach().clone().float()) for p in params]
    opt = torch.optim.AdamW(ps, lr=lr, weight_decay=weight_decay)
    sched = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, T_0=10, T_mult=2, eta_min=1e-7)
    norms, lrs = [], []
    for grads in grads_per_step:
        for p, g in zip(ps, grads):
            p.grad = g.detach().clone().float()
        norms.append(float(torch.nn.utils.clip_grad_norm_(ps, max_norm=max_norm)))
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        opt.zero_grad()
        sched.step()
    return [p.detach() for p in ps], norms, lrs
