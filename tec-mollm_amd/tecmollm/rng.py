"""NumPy mirror of the device dropout mask (csrc/common.h: tecm_hash24 / tecm_drop_mult) so that tests can
feed the *same* masks to an independent implementation."""
from __future__ import annotations

import numpy as np

_M1 = np.uint32(0x7FEB352D)
_M2 = np.uint32(0x846CA68B)


def hash24(seed: int, idx: np.ndarray) -> np.ndarray:
    idx = idx.astype(np.uint64)
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    with np.errstate(over="ignore"):
        x = (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        x = x + ((idx >> np.uint64(32)) & np.uint64(0xFFFFFF)).astype(np.uint32) * np.uint32(0x9E3779)
        x = x + np.uint32(((seed & 0xFFFFFFFF) + (seed >> 32) * 0x9E3779B1) & 0xFFFFFFFF)
        x = x ^ (x >> np.uint32(16))
        x = x * _M1
        x = x ^ (x >> np.uint32(15))
        x = x * _M2
        x = x ^ (x >> np.uint32(16))
    return (x >> np.uint32(8)).astype(np.uint32)


def keep_mult(seed: int, idx: np.ndarray, p: float) -> np.ndarray:
    """Multiplier applied by the kernels: 0 where dropped, 1/(1-p) where kept (fp32 like the device)."""
    thresh = np.uint32(np.float32(p) * np.float32(16777216.0))
    inv = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    return np.where(hash24(seed, idx) >= thresh, inv, np.float32(0.0)).astype(np.float32)
