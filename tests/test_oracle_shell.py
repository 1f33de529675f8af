"""CPU: the shell oracle (oracle/shell_cpu.py) against vectors produced by the reference's own
metrics.py / dataset.py (oracle/make_golden_shell.py), and the host-side schedule against torch's."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import shell_cpu as S

KEYS = ("mae_avg", "rmse_avg", "r2_score_avg", "pearson_r_avg", "mae_by_horizon", "rmse_by_horizon", "r2_by_horizon",
        "pearson_by_horizon")
# the reference computes on float32 arrays (sklearn/numpy pairwise float32 sums); the oracle sums in float64
TOL = dict(rtol=2e-5, atol=2e-6)


def test_metrics_with_scaler_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "shell_metrics_scaled.npz"))
    out = S.evaluate_horizons(g["y_true"], g["y_pred"], float(g["mean"]), float(g["scale"]))
    for k in KEYS:
        np.testing.assert_allclose(np.asarray(out[k]), g[f"out_{k}"], err_msg=k, **TOL)


def test_metrics_fallback_and_degenerate_horizons_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "shell_metrics_unscaled.npz"))
    out = S.evaluate_horizons(g["y_true"], g["y_pred"])
    for k in KEYS:
        np.testing.assert_allclose(np.asarray(out[k]), g[f"out_{k}"], err_msg=k, **TOL)
    assert out["pearson_by_horizon"][1] == 0.0 and out["pearson_by_horizon"][2] == 0.0
    assert out["r2_by_horizon"][3] == 1.0


def test_sliding_windows_match_reference_dataset(golden_dir):
    g = np.load(os.path.join(golden_dir, "shell_windows.npz"))
    ds = S.SlidingWindows(g["X"], g["Y"], g["TF"], int(g["L_in"]), int(g["L_out"]), int(g["stride"]))
    assert len(ds) == int(g["length"])
    for j, i in enumerate(g["pick"]):
        it = ds[int(i)]
        assert np.array_equal(it["x"], g["x"][j]) and np.array_equal(it["y"], g["y"][j])
        assert np.array_equal(it["x_time_features"], g["tf"][j])
    with pytest.raises(IndexError):
        ds[len(ds)]
    assert bool(g["index_error_past_end"])
    assert len(S.SlidingWindows(g["X"], g["Y"], g["TF"], 38, 4, 1)) == int(g["length_when_too_short"]) == 0
    x, tf, y = ds.batch([0, 2])
    N = g["X"].shape[1] * g["X"].shape[2]
    assert x.shape == (2, int(g["L_in"]), N, g["X"].shape[3]) and tf.shape == (2, int(g["L_in"]), N, 4)
    assert y.shape == (2, int(g["L_out"]), N, 1)
