"""Device-resident mirror of the reference's `SlidingWindowSamplerDataset`
(`/root/reference/src/data/dataset.py:10-99`).

The reference keeps a whole split on the host and slices one window per `__getitem__`; the DataLoader
collates B of them and the harness copies the batch to the GPU and reshapes it (train.py:58-65, :76).
An MI355X has 288 GB of HBM and a full split is a few GB, so here the split lives on the device once
and a batch is assembled by one streaming kernel (`tecm_window_batch`): x (B, L_in, N, C), the target
already as (B, L_out, N, 1), time features (B, L_in, 4) returned as the same stride-0 expanded
(B, L_in, N, 4) view train.py:65 builds.  Same constructor, `__len__`, `__getitem__` and sample indexing.
"""
from __future__ import annotations

import ctypes as C
import logging
import os
from typing import Dict, Sequence, Tuple, Union

import torch

from tecmollm import _lib
from tecmollm._lib import TecmWindowBatch, check, lib, stream_ptr

log = logging.getLogger(__name__)


class SlidingWindowSamplerDataset(torch.utils.data.Dataset):
    def __init__(self, data_path: str, mode: str, L_in: int = 336, L_out: int = 12, stride: int = 1,
                 device: Union[str, torch.device] = "cuda", tensors: Dict[str, torch.Tensor] = None):
        super().__init__()
        assert mode in ["train", "val", "test"], "Mode must be one of 'train', 'val', or 'test'"
        self.L_in, self.L_out, self.stride = L_in, L_out, stride
        if tensors is None:
            file_path = os.path.join(data_path, f"{mode}_set.pt")          # dataset.py:33
            try:
                tensors = torch.load(file_path, map_location="cpu")
            except FileNotFoundError:
                log.error("FATAL: Pre-processed data file not found at %s. Please run the preprocessing script first.",
                          file_path)
                raise
        dev = torch.device(device)
        self.X = tensors["X"].to(device=dev, dtype=torch.float32).contiguous()                         # (T, H, W, C)
        self.Y = tensors["Y"].to(device=dev, dtype=torch.float32).contiguous()                         # (T, H, W, L_out)
        self.time_features = tensors["time_features"].to(device=dev, dtype=torch.float32).contiguous()  # (T, F)
        for name in ("X", "Y", "time_features"):
            _lib.require_gpu_tensor(getattr(self, name), name)
        if self.X.dim() != 4 or self.Y.dim() != 4 or self.Y.shape[:3] != self.X.shape[:3]:
            raise ValueError("X must be (T, H, W, C) and Y (T, H, W, L_out) with matching T, H, W")
        if self.Y.shape[3] != L_out:
            raise ValueError(f"Y holds {self.Y.shape[3]} horizons, L_out = {L_out}")
        self._check_time_feature_ranges()
        max_start_idx = len(self.X) - self.L_in - self.L_out + 1          # dataset.py:47-55
        self.sample_indices = list(range(0, max_start_idx, self.stride)) if max_start_idx > 0 else []
        self.num_samples = len(self.sample_indices)
        if self.num_samples <= 0:
            log.warning("Insufficient data for windowing: Total length=%d, L_in=%d, L_out=%d, stride=%d",
                        len(self.X), L_in, L_out, stride)

    # table sizes of SpatioTemporalEmbedding (modules.py:219-225); num_years is a model setting, so the year column is
    # only required to be non-negative here and is checked against the table by the kernel (device error word)
    TIME_FEATURE_ROWS = (12, 366, None, 4)

    def _check_time_feature_ranges(self) -> None:
        """One device reduction at construction: the reference's nn.Embedding raises IndexError on an out-of-range
        time feature the first time a batch hits it (modules.py:255-258); a resident split can say so up front."""
        tf = self.time_features
        if tf.dim() != 2 or tf.shape[1] < 4 or tf.shape[0] == 0:
            return
        lo, hi = torch.aminmax(tf[:, :4].long(), dim=0)
        lo, hi = lo.tolist(), hi.tolist()
        names = ("time-of-day", "day-of-year", "year", "season")
        for c, rows in enumerate(self.TIME_FEATURE_ROWS):
            if lo[c] < 0 or (rows is not None and hi[c] >= rows):
                raise IndexError(f"time_features[:, {c}] ({names[c]}) spans [{lo[c]}, {hi[c]}], outside its embedding "
                                 f"table [0, {rows if rows is not None else 'num_years'})")

    @classmethod
    def from_tensors(cls, X, Y, time_features, L_in: int, L_out: int, stride: int = 1, device="cuda", mode: str = "train"):
        return cls("", mode, L_in, L_out, stride, device, {"X": X, "Y": Y, "time_features": time_features})

    def __len__(self) -> int:
        return self.num_samples

    def __getitem__(self, idx: int) -> dict:
        """One sample as device views (dataset.py:65-99) -- for code that iterates the dataset directly."""
        if idx >= self.num_samples:
            raise IndexError(f"Index {idx} is out of bounds for a dataset of size {self.num_samples}")
        a = self.sample_indices[idx]
        return {"x": self.X[a:a + self.L_in], "y": self.Y[a + self.L_in - 1],
                "x_time_features": self.time_features[a:a + self.L_in]}

    def batch(self, indices: Sequence[int]) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """Sample indices -> (x (B,L_in,N,C), time_features (B,L_in,N,F) expanded view, y (B,L_out,N,1))."""
        idx = [int(i) for i in indices]
        for i in idx:
            if not 0 <= i < self.num_samples:
                raise IndexError(f"Index {i} is out of bounds for a dataset of size {self.num_samples}")
        B = len(idx)
        T, H, W, Cc = self.X.shape
        N, F = H * W, self.time_features.shape[1]
        host = torch.tensor([self.sample_indices[i] for i in idx], dtype=torch.int64)
        starts = host.to(self.X.device, non_blocking=True)
        x = torch.empty(B, self.L_in, N, Cc, device=self.X.device, dtype=torch.float32)
        tf = torch.empty(B, self.L_in, F, device=self.X.device, dtype=torch.float32)
        y = torch.empty(B, self.L_out, N, 1, device=self.X.device, dtype=torch.float32)
        w = TecmWindowBatch(X=self.X.data_ptr(), TF=self.time_features.data_ptr(), Y=self.Y.data_ptr(),
                            starts=starts.data_ptr(), starts_host_check=host.data_ptr(), T=T, row=N * Cc, N=N,
                            L_in=self.L_in, L_out=self.L_out, F_t=F, B=B, x_out=x.data_ptr(), tf_out=tf.data_ptr(),
                            y_out=y.data_ptr())
        check(lib().tecm_window_batch(C.byref(w), stream_ptr()), "tecm_window_batch")
        return x, tf.unsqueeze(-2).expand(B, self.L_in, N, F), y
