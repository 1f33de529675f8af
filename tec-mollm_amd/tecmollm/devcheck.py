"""Deferred device-side error reporting.

The reference's `nn.Embedding` lookups raise on an out-of-range time index (modules.py:255-258; on a CUDA device
that is an asynchronous device-side assert which surfaces at the next synchronisation).  The fused spatial kernel
does the same thing the MI355X way: it ORs a bit into a per-device int32 error word and turns every value built from
the bad index into NaN (so nothing downstream can look valid), and the host reads the word back WITHOUT adding a
synchronisation to the step:

  * after every launch that can set the word, `post()` queues an async copy of it into pinned host memory and records
    an event on the launch stream;
  * `poll()` -- called at the start of the next forward, by the epoch loops, by bench.py after its timed region and by
    anything that already synchronises -- raises `IndexError` once that copy has landed and the word is non-zero;
    `poll(sync=True)` waits for the copy first.

Bit 16 is used by the data-parallel rank-divergence check (tecmollm/train.py).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

BAD_TOD, BAD_DOY, BAD_YEAR, BAD_SEASON = 1, 2, 4, 8
RANKS_DIVERGED = 1 << 16

_NAMES = {BAD_TOD: "time-of-day index outside [0, 12)", BAD_DOY: "day-of-year index outside [0, 366)",
          BAD_YEAR: "year index outside [0, num_years)", BAD_SEASON: "season index outside [0, 4)",
          RANKS_DIVERGED: "data-parallel ranks no longer hold identical parameters"}


class DeviceErrorWord:
    def __init__(self, device: torch.device):
        self.device = device
        self.word = torch.zeros(1, device=device, dtype=torch.int32)
        self.host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.event: Optional[torch.cuda.Event] = None

    def ptr(self) -> int:
        return self.word.data_ptr()

    def post(self) -> None:
        """Queue the read-back of the word behind everything launched so far on the current stream.  While a stream is
        being captured (TrainStep.step_graphed) nothing is queued: events recorded inside a capture cannot be queried from
        outside it -- the step posts once after every replay instead."""
        if torch.cuda.is_current_stream_capturing():
            return
        self.host.copy_(self.word, non_blocking=True)
        if self.event is None:
            self.event = torch.cuda.Event()
        self.event.record()

    def poll(self, sync: bool = False) -> None:
        if self.event is None or torch.cuda.is_current_stream_capturing():
            return
        if sync:
            self.event.synchronize()
        elif not self.event.query():
            return
        code = int(self.host[0])
        if code:
            self.word.zero_()
            self.host.zero_()
            self.event = None
            what = "; ".join(text for bit, text in _NAMES.items() if code & bit)
            if code & RANKS_DIVERGED and not (code & 15):
                raise RuntimeError(f"device error word = {code:#x}: {what}")
            raise IndexError(f"index out of range in self (device error word = {code:#x}): {what} -- the reference's "
                             "nn.Embedding raises here (modules.py:255-258); every output built from the bad index is NaN")


_words: Dict[str, DeviceErrorWord] = {}


def error_word(device: torch.device) -> DeviceErrorWord:
    key = str(torch.device(device))
    if key == "cuda":
        key = f"cuda:{torch.cuda.current_device()}"
    w = _words.get(key)
    if w is None:
        w = _words[key] = DeviceErrorWord(torch.device(key))
    return w


def check_device_errors(device=None, sync: bool = True) -> None:
    """Raise if any kernel launched so far reported an error (waits for the read-back when sync=True)."""
    if device is not None:
        if torch.device(device).type == "cuda":
            error_word(device).poll(sync)
        return
    for w in list(_words.values()):
        w.poll(sync)
