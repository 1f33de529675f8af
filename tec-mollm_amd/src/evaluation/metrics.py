"""Device-side mirror of the reference's `src/evaluation/metrics.py` (`/root/reference/src/evaluation/metrics.py`).

`evaluate_horizons` (:119-183) is what `validate()` calls once per epoch after collecting every batch's
predictions on the host (`train.py:153-166`, one `.cpu().numpy()` sync per batch).  Here the per-horizon
sufficient statistics are accumulated on the GPU batch by batch (`HorizonMetrics.update`, one kernel,
no sync, reads the model's permuted output view in place) and the 4 metrics x L_out horizons are derived
from 8*L_out doubles in `compute()` -- the only device->host copy of the whole evaluation.

Same numbers as the reference: inverse StandardScaler transform, non-finite guards, clip of predictions
to [0, 200] TECU, MAE / RMSE / R^2 (sklearn semantics) / Pearson r per horizon and their averages.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional, Tuple, Union

import numpy as np
import torch

from tecmollm import _lib
from tecmollm._lib import TecmError, TecmMetrics, check, lib, stream_ptr

TEC_MIN, TEC_MAX = 0.0, 200.0           # metrics.py:50


def _scaler_params(scaler) -> Optional[Tuple[float, float]]:
    """Accepts None, a (mean, scale) pair, a fitted sklearn StandardScaler, or a joblib path to one (:148-150)."""
    if scaler is None:
        return None
    if isinstance(scaler, str):
        import joblib
        scaler = joblib.load(scaler)
    if isinstance(scaler, (tuple, list)):
        return float(scaler[0]), float(scaler[1])
    mean = np.asarray(scaler.mean_).reshape(-1)
    scale = np.asarray(scaler.scale_).reshape(-1)
    if mean.size != 1 or scale.size != 1:
        raise ValueError("the target scaler must be fitted on a single feature (metrics.py:26-37)")
    return float(mean[0]), float(scale[0])


def _as_shi(t: torch.Tensor, name: str):
    """(S, H, ...) tensor -> element strides (s, h, i) with the trailing dims flattened to one index."""
    _lib.require_gpu_tensor(t, name)
    if t.dim() < 2:
        raise ValueError(f"{name} must be (samples, horizons, ...)")
    S, H = t.shape[0], t.shape[1]
    rest = t.shape[2:]
    I = int(np.prod(rest)) if rest else 1
    # the trailing dims must collapse to a single stride (true for contiguous tensors and for the
    # (B, L_out, N, 1) permuted view the model returns)
    stride_i, expect = None, None
    for size, st in zip(reversed(rest), reversed(t.stride()[2:])):
        if size == 1:
            continue
        if stride_i is None:
            stride_i, expect = st, st * size
        elif st != expect:
            return _as_shi(t.contiguous(), name)
        else:
            expect = st * size
    return t, S, H, I, t.stride(0), t.stride(1), (stride_i if stride_i is not None else 1)


class HorizonMetrics:
    """Streaming `evaluate_horizons`: update(pred, true) per batch on the device, compute() at the end."""

    def __init__(self, num_horizons: int, scaler=None, device: Union[str, torch.device] = "cuda"):
        self.H = int(num_horizons)
        self.scaler = _scaler_params(scaler)
        self.stats = torch.zeros(self.H, _lib.TECM_METRIC_STATS, device=device, dtype=torch.float64)

    def reset(self) -> None:
        self.stats.zero_()

    def update(self, y_pred_scaled: torch.Tensor, y_true_scaled: torch.Tensor) -> None:
        p, S, H, I, ps, ph, pi = _as_shi(y_pred_scaled, "y_pred")
        t, S2, H2, I2, ts, th, ti = _as_shi(y_true_scaled, "y_true")
        if (S, H, I) != (S2, H2, I2) or H != self.H:
            raise ValueError(f"shape mismatch: pred {tuple(y_pred_scaled.shape)}, true {tuple(y_true_scaled.shape)}, "
                             f"horizons {self.H}")
        mean, scale = self.scaler if self.scaler is not None else (0.0, 1.0)
        m = TecmMetrics(pred=p.data_ptr(), p_stride_s=ps, p_stride_h=ph, p_stride_i=pi,
                        target=t.data_ptr(), t_stride_s=ts, t_stride_h=th, t_stride_i=ti,
                        S=S, H=H, I=I, mean=mean, scale=scale, clip_lo=TEC_MIN, clip_hi=TEC_MAX,
                        clip=1 if self.scaler is not None else 0, stats=self.stats.data_ptr())
        check(lib().tecm_metrics_accumulate(C.byref(m), stream_ptr()), "tecm_metrics_accumulate")

    def compute(self) -> Dict[str, object]:
        st = self.stats.cpu().numpy()
        per = []
        for n, st_, sp, stt, spp, stp, sabs, ssq in st:
            if n == 0:
                raise ValueError("HorizonMetrics.compute() before any update()")
            mae, rmse = sabs / n, math.sqrt(ssq / n)
            ss_tot = stt - st_ * st_ / n
            ss_p = spp - sp * sp / n
            # a constant series must read as exactly zero variance (np.std(...) > 0 test, metrics.py:73):
            # the one-pass form leaves rounding noise of order eps * sum(t^2)
            var_floor = 64 * np.finfo(np.float64).eps
            if ss_tot <= var_floor * max(stt, 1e-300):
                ss_tot = 0.0
            if ss_p <= var_floor * max(spp, 1e-300):
                ss_p = 0.0
            if ss_tot != 0.0:
                r2 = 1.0 - ssq / ss_tot
            else:
                r2 = 1.0 if ssq == 0.0 else 0.0
            if ss_tot > 0 and ss_p > 0:
                pear = (stp - st_ * sp / n) / math.sqrt(ss_tot * ss_p)
                pear = max(-1.0, min(1.0, pear))
            else:
                pear = 0.0
            per.append({"mae": mae, "rmse": rmse, "r2_score": r2, "pearson_r": pear})
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


def evaluate_horizons(y_true_horizons_scaled, y_pred_horizons_scaled, target_scaler_path: Optional[str] = None,
                      device: Union[str, torch.device] = "cuda") -> Dict[str, object]:
    """Same signature and result dict as the reference (:119-183); arrays (S, L_out, ...) may be CUDA tensors
    (used in place) or numpy arrays / CPU tensors (uploaded first -- the arithmetic always runs on the GPU)."""
    def dev(a):
        t = torch.as_tensor(a)
        return t.to(device=device, dtype=torch.float32)
    t, p = dev(y_true_horizons_scaled), dev(y_pred_horizons_scaled)
    hm = HorizonMetrics(t.shape[1], target_scaler_path, device=t.device)
    hm.update(p, t)
    return hm.compute()
