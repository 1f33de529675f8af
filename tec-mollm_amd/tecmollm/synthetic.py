"""Synthetic inputs with the reference's shapes (bench.py, tools/): no data set ships with the repo.
x ~ N(0,1) like the z-scored features (feature_engineering.py:186), integer time features (tod < 12, doy < 366,
year < 13, season < 4) as float32 expanded stride-0 over the nodes exactly as train.py:65 does, y ~ N(0,1)."""
from __future__ import annotations

import torch


def synthetic_batch(B: int, L_in: int, N: int, c_in: int, L_out: int, seed: int = 1234):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, L_in, N, c_in, generator=g)
    tf = torch.stack([torch.randint(0, hi, (B, L_in), generator=g) for hi in (12, 366, 13, 4)], -1).float()
    tf = tf.unsqueeze(-2).expand(B, L_in, N, 4)
    y = torch.randn(B, L_out, N, 1, generator=g)
    return x, tf, y


def grid_graph(n_lat: int = 41, n_lon: int = 71, threshold_km: float = 150.0):
    """(edge_index, edge_weight) of the reference's 41 x 71 grid."""
    from src.graph.graph_constructor import build_grid_graph, china_grid
    lat, lon = china_grid(n_lat, n_lon)
    return build_grid_graph(lat, lon, threshold_km)
