"""Golden vectors for the shell rows (metrics, sliding windows) from the REFERENCE's own code.

Run in the authoring container only (needs /root/reference):   python oracle/make_golden_shell.py

Imports `/root/reference/src/evaluation/metrics.py` (numpy / sklearn / scipy / joblib -- all installed),
`/root/reference/src/data/dataset.py` (torch only) and the numpy/scipy/sklearn functions of
`/root/reference/src/graph/graph_constructor.py` unmodified, feeds seeded inputs, stores inputs and
the functions' outputs under tests/golden/shell_*.npz.
"""
from __future__ import annotations

import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

KEYS = ("mae_avg", "rmse_avg", "r2_score_avg", "pearson_r_avg", "mae_by_horizon", "rmse_by_horizon", "r2_by_horizon",
        "pearson_by_horizon")


def _pack(d):
    return {f"out_{k}": np.asarray(d[k], dtype=np.float64) for k in KEYS}


def main():
    import logging
    sys.path.insert(0, REF)
    from src.evaluation import metrics as RM
    from src.data.dataset import SlidingWindowSamplerDataset
    logging.disable(logging.CRITICAL)
    import joblib
    from sklearn.preprocessing import StandardScaler
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(7)

    # ------------------------------------------------------------ metrics with a fitted StandardScaler
    S, H, N = 6, 12, 37
    tec = rng.gamma(2.0, 12.0, size=(4000, 1)).astype(np.float64)           # TECU-like positive skewed values
    scaler = StandardScaler().fit(tec)
    mean, scale = float(scaler.mean_[0]), float(scaler.scale_[0])
    y_true = rng.standard_normal((S, H, N, 1)).astype(np.float32)
    y_pred = (y_true + 0.3 * rng.standard_normal((S, H, N, 1)) * (1 + np.arange(H)[None, :, None, None] / 6)).astype(np.float32)
    y_pred[0, 0, :5, 0] = -9.0                                              # un-scales below 0 -> clipped to 0
    y_pred[1, 3, :4, 0] = 40.0                                              # un-scales above 200 -> clipped
    y_pred[2, 5, 0, 0] = np.nan
    y_pred[2, 5, 1, 0] = np.inf
    y_pred[2, 6, 2, 0] = -np.inf
    with tempfile.TemporaryDirectory() as td:
        sp = os.path.join(td, "target_scaler.joblib")
        joblib.dump(scaler, sp)
        out = RM.evaluate_horizons(y_true.copy(), y_pred.copy(), sp)
    np.savez_compressed(os.path.join(OUT, "shell_metrics_scaled.npz"), y_true=y_true, y_pred=y_pred,
                        mean=np.float64(mean), scale=np.float64(scale), **_pack(out))

    # ------------------------------------------------------------ no scaler (fallback :89-117), incl. degenerate horizons
    y_true2 = (20 + 8 * rng.standard_normal((S, 5, N, 1))).astype(np.float32)
    y_pred2 = (y_true2 + rng.standard_normal((S, 5, N, 1))).astype(np.float32)
    y_true2[:, 1] = 3.5                                                     # constant target: pearson 0, r2 force_finite
    y_pred2[:, 2] = 1.25                                                    # constant prediction: pearson 0
    y_true2[:, 3] = 2.0
    y_pred2[:, 3] = 2.0                                                     # both constant and equal: r2 = 1
    out2 = RM.evaluate_horizons(y_true2.copy(), y_pred2.copy(), None)
    np.savez_compressed(os.path.join(OUT, "shell_metrics_unscaled.npz"), y_true=y_true2, y_pred=y_pred2, **_pack(out2))

    # ------------------------------------------------------------ sliding windows from the reference Dataset
    T, Hh, Ww, C, L_in, L_out, stride = 40, 3, 5, 6, 7, 4, 3
    X = torch.from_numpy(rng.standard_normal((T, Hh, Ww, C)).astype(np.float32))
    Y = torch.from_numpy(rng.standard_normal((T, Hh, Ww, L_out)).astype(np.float32))
    TF = torch.from_numpy(np.stack([rng.integers(0, 12, T), rng.integers(0, 366, T), rng.integers(0, 13, T),
                                    rng.integers(0, 4, T)], axis=1).astype(np.float32))
    with tempfile.TemporaryDirectory() as td:
        torch.save({"X": X, "Y": Y, "time_features": TF}, os.path.join(td, "val_set.pt"))
        ds = SlidingWindowSamplerDataset(td, "val", L_in=L_in, L_out=L_out, stride=stride)
        n = len(ds)
        pick = [0, 1, n // 2, n - 1]
        items = [ds[i] for i in pick]
        try:
            ds[n]
            raised = False
        except IndexError:
            raised = True
        ds_short = SlidingWindowSamplerDataset(td, "val", L_in=38, L_out=4, stride=1)      # too short: zero samples
    np.savez_compressed(os.path.join(OUT, "shell_windows.npz"), X=X.numpy(), Y=Y.numpy(), TF=TF.numpy(),
                        L_in=L_in, L_out=L_out, stride=stride, length=n, pick=np.asarray(pick),
                        x=np.stack([it["x"].numpy() for it in items]), y=np.stack([it["y"].numpy() for it in items]),
                        tf=np.stack([it["x_time_features"].numpy() for it in items]),
                        index_error_past_end=raised, length_when_too_short=len(ds_short))
    # ------------------------------------------------------------ grid graph from the reference's graph_constructor
    # src/graph/graph_constructor.py imports src.data.data_loader (h5py, not installed) only for its HDF5 helper; the
    # name is pre-seeded as an empty placeholder so that the pure numpy/scipy/sklearn functions can be called.
    import types
    ph = types.ModuleType("src.data.data_loader")
    ph.load_and_split_data = None
    sys.modules["src.data.data_loader"] = ph
    from src.graph import graph_constructor as GC
    graphs = {}
    for tag, (lat, lon, thr) in {"full": (15.0 + np.arange(41), 70.0 + np.arange(71), 150.0),
                                 "small": (30.0 + 0.5 * np.arange(5), 100.0 + 0.5 * np.arange(7), 120.0)}.items():
        dist = GC.calculate_haversine_distance_matrix(lat, lon)
        adj = GC.construct_binary_adjacency(dist, thr)
        norm = GC.symmetrically_normalize_adjacency(adj)
        graphs[f"{tag}_lat"], graphs[f"{tag}_lon"], graphs[f"{tag}_thr"] = lat, lon, np.float64(thr)
        graphs[f"{tag}_edge_index"] = np.vstack((norm.row, norm.col)).astype(np.int32)
        graphs[f"{tag}_edge_weight"] = norm.data.astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "shell_graph.npz"), **graphs)
    print("wrote", sorted(f for f in os.listdir(OUT) if f.startswith("shell_")))


if __name__ == "__main__":
    main()
