"""The reference's epoch loops (train.py:52-128 `train_one_epoch`, :130-168 `validate`) over the device-resident
pieces of this package: `SlidingWindowSamplerDataset.batch` feeds `TrainStep.step`, `HorizonMetrics` replaces the
per-batch `.cpu().numpy()` + end-of-epoch `evaluate_horizons`.  No host synchronisation inside an epoch except the
loss read-out at its end."""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import torch

from .train import TrainStep


def _batches(n: int, batch_size: int, order: Optional[Sequence[int]] = None, drop_last: bool = False):
    idx = list(range(n)) if order is None else list(order)
    for a in range(0, len(idx), batch_size):
        chunk = idx[a:a + batch_size]
        if drop_last and len(chunk) < batch_size:
            return
        yield chunk


def train_one_epoch(ts: TrainStep, dataset, edge_index: torch.Tensor, batch_size: int,
                    order: Optional[Sequence[int]] = None, edge_weight: Optional[torch.Tensor] = None) -> float:
    """train.py:52-128: one pass over `dataset` (sample order `order`, e.g. a DistributedSampler's indices),
    optimizer every `ts.accumulation_steps` batches, trailing partial accumulation flushed (train.py:117-126).
    Returns the mean batch loss."""
    ts.model.train()
    total = torch.zeros((), device=edge_index.device)
    nb = 0
    for chunk in _batches(len(dataset), batch_size, order):
        x, tf, y = dataset.batch(chunk)
        total += ts.step(x, tf, edge_index, edge_weight, y)
        nb += 1
    ts.finish_accumulation()                       # trailing partial cycle (train.py:117-126); a no-op when none is open
    from .devcheck import check_device_errors
    mean = float(total) / max(nb, 1)               # the epoch's one host synchronisation
    check_device_errors(edge_index.device, sync=True)
    return mean


@torch.no_grad()
def validate(model: torch.nn.Module, dataset, edge_index: torch.Tensor, batch_size: int, scaler=None,
             edge_weight: Optional[torch.Tensor] = None) -> Tuple[float, Dict[str, object]]:
    """train.py:130-168: mean HuberLoss(delta=1) over the batches + the `evaluate_horizons` dict."""
    from src.evaluation.metrics import HorizonMetrics
    from .functions import HuberFn
    model.eval()
    hm = None
    total = torch.zeros((), device=edge_index.device)
    nb = 0
    for chunk in _batches(len(dataset), batch_size):
        x, tf, y = dataset.batch(chunk)
        out = model(x, tf, edge_index, edge_weight)
        total += HuberFn.apply(out, y, 1.0)
        if hm is None:
            hm = HorizonMetrics(out.shape[1], scaler, device=out.device)
        hm.update(out, y)
        nb += 1
    if hm is None:
        raise ValueError("validate() on an empty dataset")
    return float(total) / nb, hm.compute()
