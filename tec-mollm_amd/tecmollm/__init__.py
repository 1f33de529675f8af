"""tecmollm -- MI355X (gfx950) implementation of the TEC-MoLLM forward/backward hot path.

Package layout (inside `tec-mollm_amd/`, which must be on sys.path):
  csrc/            hand-written HIP kernels + the C ABI (include/tecmollm.h at the repo root)
  tecmollm/        ctypes binding, autograd stage functions, graph preparation, training-step helpers
  src/model/       mirror of the reference's module API (`from src.model.tec_mollm import TEC_MoLLM`)
"""
from ._lib import LIB_PATH, TecmError, lib  # noqa: F401
from .devcheck import check_device_errors  # noqa: F401

__all__ = ["LIB_PATH", "TecmError", "lib", "check_device_errors"]
