"""Checkpoint I/O with the reference's conventions (train.py:440-447 save, test.py:175-190 load).

The reference saves `model.state_dict()` (unwrapped from DDP) with torch.save and, on load, strips the
`module.` (DDP) and `_orig_mod.` (torch.compile) prefixes before a strict load_state_dict.  The mirror
modules expose the reference's parameter names, so the same files go both ways.
"""
from __future__ import annotations

from typing import Dict

import torch

PREFIXES = ("module.", "_orig_mod.")


def strip_wrapper_prefixes(state: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """test.py:178-187: drop one leading `module.` and then one leading `_orig_mod.` from every key."""
    out = {}
    for k, v in state.items():
        for pre in PREFIXES:
            if k.startswith(pre):
                k = k[len(pre):]
        out[k] = v
    return out


def save_model(model: torch.nn.Module, path: str) -> None:
    """train.py:444-445: the bare state_dict of the unwrapped model."""
    target = model.module if hasattr(model, "module") else model
    torch.save(target.state_dict(), path)


def load_model(model: torch.nn.Module, path: str, map_location=None, strict: bool = True):
    """test.py:175-190.  Parameters keep their storage (load_state_dict copies in place), so flat
    optimizer views (tecmollm.optim.FlatAdamW) stay valid."""
    state = torch.load(path, map_location=map_location or "cpu")
    if isinstance(state, dict) and "state_dict" in state and not any(k.endswith(".weight") for k in state):
        state = state["state_dict"]
    target = model.module if hasattr(model, "module") else model
    return target.load_state_dict(strip_wrapper_prefixes(state), strict=strict)
