"""One full training step (forward, loss, all gradients) against the CPU oracle over model configurations the reference's
ctor accepts and the timed configurations do not touch: sequence lengths 8..96, channel lists, strides, patch lengths,
feature widths, layer counts -- in fp32 and bf16 mode, eval and train (tools/config_sweep.py as a test: a regression fails
here instead of landing in a profile file).  The problems are tiny (12 nodes = 24 sequences) so that the CPU oracle
finishes in seconds; fp32 takes the standard bars, bf16 the stated 24-sequence bars of tests/parity.py."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tec-mollm_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import ref_cpu as R  # noqa: E402
from tests.parity import assert_parity, compare_forward_backward  # noqa: E402

pytestmark = pytest.mark.gpu

# (tools/config_sweep.py walks the full matrix -- eight sequence lengths, four (precision, mode) pairs: profiles/r05_config_sweep.txt;
#  the test keeps the lengths that take different kernel routes and the two train-mode pairs -- eval is the same kernels
#  with the mask multiply off -- to stay well inside a minute of GPU time)
CASES = [(f"L{L_in}_{L_out}", dict(L_in=L_in, L_out=L_out), {}) for L_in, L_out in
         ((16, 4), (40, 12), (80, 12), (96, 24), (8, 2))] + [
    ("ch128_256", dict(), {"temporal_channel_list": [128, 256]}),
    ("ch64_64", dict(), {"temporal_channel_list": [64, 64]}),
    # shapes whose fp32 sequence tiles do NOT fit the LDS (tecm_conv_*_supported says no): the window-GEMM fallback serves them
    ("ch64_256_L96", dict(L_in=96, L_out=24), {"temporal_channel_list": [64, 256]}),
    ("ch128_128_L96", dict(L_in=96, L_out=24), {"temporal_channel_list": [128, 128]}),
    ("strides1_2", dict(), {"temporal_strides": [1, 2], "patch_len": 4}),
    ("patch2", dict(), {"patch_len": 2}),
    ("F10", dict(c_in=10, d_emb=12), {}),
    ("layers1", dict(llm_layers=1), {}),
]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda")


@pytest.mark.parametrize("prec,train", [("fp32", True), ("bf16", True)], ids=["fp32-train", "bf16-train"])
@pytest.mark.parametrize("name,kw,over", CASES, ids=[c[0] for c in CASES])
def test_full_step_over_model_configurations(dev, name, kw, over, prec, train):
    cfg = R.default_config(num_nodes=12, **kw)
    cfg.update(over)
    res = compare_forward_backward(cfg, B=2, grid=(3, 4), threshold_km=170.0, gat_graphs="per_timestep", seed=5, train=train,
                                   precision=prec)
    if prec == "fp32":
        assert_parity(res)
    else:
        assert_parity(res, small24=True)


def test_unsupported_feature_width_fails_loudly(dev):
    """C = C_in + d_emb != 22 is the one configuration the kernels are not built for: it must raise, not mis-compute."""
    from tecmollm import TecmError
    cfg = R.default_config(num_nodes=12, c_in=4, d_emb=8)
    with pytest.raises((TecmError, ValueError, RuntimeError)):
        compare_forward_backward(cfg, B=2, grid=(3, 4), threshold_km=170.0, gat_graphs="per_timestep", seed=5, precision="fp32")
