"""CPU tests of the host side: the C-ABI library loads and exports every symbol declared in
include/tecmollm.h, the graph preparation (CSR + LDS windows), the dropout-hash mirror, the module
API mirror (parameter names / freeze rule) and the data-parallel step on 2 gloo ranks."""
import ctypes
import os
import re
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_cpu as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tec-mollm_amd", "tecmollm", "libtecmollm_hip.so")


@pytest.fixture(scope="session")
def built_lib():
    if not os.path.exists(LIB):
        import __graft_entry__ as g
        g.build()
    return LIB


def test_library_exports_every_declared_symbol(built_lib):
    header = open(os.path.join(ROOT, "include", "tecmollm.h")).read()
    declared = set(re.findall(r"\b(tecm_[a-z0-9_]+)\s*\(", header))
    assert {"tecm_gemm_f32", "tecm_spatial_fwd", "tecm_spatial_bwd", "tecm_attention_bwd"} <= declared
    handle = ctypes.CDLL(built_lib)
    for name in sorted(declared):
        assert hasattr(handle, name), f"{name} declared in tecmollm.h but not exported"
    handle.tecm_abi_version.restype = ctypes.c_int
    from tecmollm import _lib
    in_header = int(re.search(r"#define\s+TECM_ABI_VERSION\s+(\d+)", header).group(1))
    assert handle.tecm_abi_version() == _lib.ABI_VERSION == in_header
    handle.tecm_last_error.restype = ctypes.c_char_p
    assert isinstance(handle.tecm_last_error(), bytes)


def test_binding_matches_header(built_lib):
    from tecmollm import _lib
    header = open(os.path.join(ROOT, "include", "tecmollm.h")).read()
    declared = set(re.findall(r"\b(tecm_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert ctypes.sizeof(_lib.TecmWin) == 32 and ctypes.sizeof(_lib.TecmDrop) == 32
    _lib.lib()                                              # loads, sets prototypes, checks the ABI version


def test_conv_sequence_kernels_report_what_they_serve_including_the_lds_budget(built_lib):
    """tecm_conv_{fwd,dx}_supported are pure host arithmetic (no GPU call): the predicates ConvBlockFn routes on must say no
    exactly where the launchers would refuse with TECM_E_LDS, so that such shapes fall back to the window-view GEMMs
    (Multi_Scale_Conv_Block accepts any channel list, modules.py:19-41).  Cases: the timed shapes; out_channels = 256 with an
    fp32 d-input image (3088-B rows x 4 nodes x 14 steps > 160 KiB); ld_in = 128 forward in fp32 (528-B rows x 4 x 39 > 64 KiB)."""
    from tecmollm import ops
    assert ops.conv_fwd_seq_ok(48, 64, 24, f32=True) and ops.conv_fwd_seq_ok(48, 64, 24, f32=False)
    assert ops.conv_fwd_seq_ok(24, 128, 64, f32=True) and ops.conv_dx_seq_ok(24, 128, 64, f32=True)
    assert ops.conv_dx_seq_ok(48, 64, 24, f32=True) and ops.conv_dx_seq_ok(48, 64, 24, f32=False)
    assert not ops.conv_dx_seq_ok(16, 256, 64, f32=True)          # [64, 256]: the fp32 image does not fit one CU's LDS
    assert ops.conv_dx_seq_ok(16, 256, 64, f32=False)             # ... the bf16 image does
    assert not ops.conv_fwd_seq_ok(48, 128, 128, f32=True)        # [128, 128] at L_in = 96: fp32 rows of 528 B
    assert ops.conv_fwd_seq_ok(48, 128, 128, f32=False)
    assert not ops.conv_fwd_seq_ok(48, 256, 64) and not ops.conv_dx_seq_ok(20, 64, 24) and not ops.conv_dx_seq_ok(48, 64, 128)


def test_eight_phase_gemm_tile_height_is_chosen_by_tile_count(built_lib, monkeypatch):
    """tecm_p8_rows is pure host arithmetic (no launch; without a GPU it assumes 256 CUs): 224-row tiles only for results at
    most three tiles wide or at most 32 768 rows tall whose tile count then fills the last round of CUs better, 256 rows
    otherwise; TECM_P8_ROWS pins a height (csrc/gemm_bf16_p8.hip)."""
    from tecmollm import _lib
    h = _lib.lib()
    monkeypatch.delenv("TECM_P8_ROWS", raising=False)
    assert h.tecm_p8_rows(69864, 768) == 112           # 819 tiles = 3.2 rounds -> 936 tiles of 7/8 the work
    assert h.tecm_p8_rows(69864, 3072) == 128          # 12.8 rounds: nothing to win
    assert h.tecm_p8_rows(69864, 2304) == 128          # wider than three tiles: measured slower on short tiles
    assert h.tecm_p8_rows(23288, 2304) == 112          # few m-tiles (the head's d-input)
    assert h.tecm_p8_rows(65536, 768) == 128           # 768 tiles = exactly three rounds
    for rows in (128, 112, 96):
        monkeypatch.setenv("TECM_P8_ROWS", str(rows))
        assert h.tecm_p8_rows(69864, 3072) == rows
    monkeypatch.setenv("TECM_P8_ROWS", "100")
    assert h.tecm_p8_rows(69864, 3072) == 128          # not a height the kernel is built for: ignored


def test_product_path_refuses_cpu_tensors(built_lib):
    """No CPU fallback: the model raises instead of silently computing somewhere else."""
    from tests.parity import build_model
    from tecmollm import TecmError
    cfg = R.default_config(L_in=16, num_nodes=12)
    model = build_model(cfg, R.init_params(cfg, 0), "cpu")
    x, tf, _ = R.synthetic_batch(1, 16, 12, 6, 12)
    ei, _ = R.grid_graph(3, 4, threshold_km=170.0)
    with pytest.raises(TecmError):
        model(x, tf, ei)


def test_state_dict_keys_and_freeze_rule_match_reference_names():
    """SURVEY.md section 8a parameter names; freeze rule modules.py:195-203; census 3 081 996 trainable."""
    from src.model.tec_mollm import TEC_MoLLM
    cfg = R.default_config()
    cfg.update(load_pretrained_gpt2=False, include_wte=True)
    model = TEC_MoLLM(cfg)
    oracle_names = set(R.init_params(R.default_config(num_nodes=8), 0, include_wte=True))
    assert set(model.state_dict().keys()) == oracle_names
    for n, p in model.named_parameters():
        assert p.requires_grad == R.is_trainable(n), n
    assert sum(p.numel() for p in model.parameters() if p.requires_grad) == 3_081_996
    assert model.state_dict()["llm_backbone.model.base_model.model.wte.weight"].shape == (50257, 768)
    model.llm_backbone.model.gradient_checkpointing_enable()


def test_gpt2_trunk_keys_equal_transformers_gpt2model_keys():
    """test.py:175-190 loads checkpoints with strict=True, so the trunk must expose exactly the parameter names of the
    transformers GPT2Model the reference wraps (modules.py:165-170), modulo peft's c_attn -> c_attn.base_layer rename
    and the added lora_A/lora_B; persistent buffers included."""
    from transformers import GPT2Config, GPT2Model
    from src.model.modules import LLMBackbone, hf_key_of
    with torch.device("meta"):
        hf = GPT2Model(GPT2Config())
    hf_keys = {k for k in hf.state_dict() if not k.startswith("h.") or int(k.split(".")[1]) < 3}
    bb = LLMBackbone(3, include_wte=True, load_pretrained=False)
    own = {k for k in bb.trunk.state_dict()}
    lora = {k for k in own if ".lora_A." in k or ".lora_B." in k}
    assert lora == {f"h.{i}.attn.c_attn.lora_{ab}.default.weight" for i in range(3) for ab in "AB"}
    assert {hf_key_of(k) for k in own - lora} == hf_keys
    shapes_hf = {k: tuple(v.shape) for k, v in hf.state_dict().items()}
    for k, v in bb.trunk.state_dict().items():
        if k not in lora:
            assert tuple(v.shape) == shapes_hf[hf_key_of(k)], k


def test_pretrained_gpt2_copy_is_complete_or_raises():
    """A partial copy of the GPT-2 checkpoint must raise (the reference's from_pretrained either loads everything or
    fails, modules.py:165): every non-LoRA trunk tensor is copied, a missing or mis-shaped key is an error, and the
    default constructor does not fall back to random weights when the checkpoint cannot be had."""
    from transformers import GPT2Config, GPT2Model
    from src.model.modules import LLMBackbone, copy_hf_gpt2_weights
    torch.manual_seed(0)
    hf = GPT2Model(GPT2Config(n_layer=3, vocab_size=64))
    bb = LLMBackbone(3, include_wte=False, load_pretrained=False)
    sd = hf.state_dict()
    with torch.no_grad():
        n = copy_hf_gpt2_weights(bb.trunk, sd)
    assert n == len([k for k in bb.trunk.state_dict() if "lora_" not in k])
    assert torch.equal(bb.trunk.h[2].attn.c_attn.base_layer.weight, sd["h.2.attn.c_attn.weight"])
    assert torch.equal(bb.trunk.wpe.weight, sd["wpe.weight"])
    broken = {k: v for k, v in sd.items() if k != "h.1.mlp.c_fc.bias"}
    with pytest.raises(RuntimeError, match="h.1.mlp.c_fc.bias"):
        copy_hf_gpt2_weights(bb.trunk, broken)
    broken = dict(sd)
    broken["ln_f.weight"] = torch.zeros(5)
    with pytest.raises(RuntimeError, match="ln_f.weight"):
        copy_hf_gpt2_weights(bb.trunk, broken)
    try:
        full = LLMBackbone(3, include_wte=False)             # load_pretrained defaults to True, like the reference
    except Exception as e:                                   # offline box: loud, never a silent random backbone
        assert "gpt2" in str(e).lower() or "offline" in str(e).lower() or "connect" in str(e).lower(), e
    else:
        assert float(full.trunk.wpe.weight.abs().max()) > 0.2     # the real checkpoint (config init is N(0, 0.02))


def test_graph_csr_and_windows():
    from tecmollm import graph
    ei, _ = R.grid_graph()
    rowptr, col = graph.csr_by_target(ei.numpy(), 2911)
    assert rowptr[-1] == 20924 and (np.diff(rowptr) >= 2).all() and (np.diff(rowptr) <= 10).all()
    for i in (0, 70, 1500, 2910):
        assert sorted(col[rowptr[i]:rowptr[i + 1]].tolist()) == sorted(ei[0][ei[1] == i].tolist())
    lo, hi = graph.tile_windows(rowptr, col, 2911, 128)
    assert lo.shape == (23,) and (hi - lo).max() <= 128 + 2 * 73
    for k in range(23):
        n0, n1 = k * 128, min(2911, (k + 1) * 128)
        c = col[rowptr[n0]:rowptr[n1]]
        assert lo[k] <= min(n0, c.min()) and hi[k] >= max(n1, c.max() + 1)
    meta = graph.build(ei, 2911, torch.device("cpu"))
    assert meta.tile_nodes == 112 and meta.win_max <= 256 and meta.max_deg == 10 and meta.num_edges == 20924
    assert graph.lds_bytes_bwd(meta.win_max, meta.tile_nodes) <= 160 * 1024


def test_graph_self_loops_dropped_duplicates_kept_and_bad_ids_rejected():
    from tecmollm import graph
    ei = np.array([[0, 1, 1, 2, 2, 3], [1, 1, 0, 1, 1, 9]])
    with pytest.raises(ValueError):
        graph.csr_by_target(ei, 4)
    rowptr, col = graph.csr_by_target(ei[:, :5], 4)
    assert rowptr.tolist() == [0, 1, 4, 4, 4] and col.tolist() == [1, 0, 2, 2]
    far = torch.tensor([[0, 49999], [49999, 0]])
    with pytest.raises(ValueError, match="renumber"):
        graph.build(far, 50000, torch.device("cpu"))
    empty = graph.build(torch.zeros(2, 0, dtype=torch.long), 10, torch.device("cpu"))
    assert empty.num_edges == 0 and empty.max_deg == 0


def test_rng_mirror_statistics_and_determinism():
    from tecmollm import rng
    idx = np.arange(1 << 18, dtype=np.uint64)
    a = rng.keep_mult(12345, idx, 0.1)
    assert np.array_equal(a, rng.keep_mult(12345, idx, 0.1))
    assert abs((a == 0).mean() - 0.1) < 0.005
    assert np.allclose(a[a != 0], 1 / 0.9)
    b = rng.keep_mult(12346, idx, 0.1)
    assert abs(((a == 0) & (b == 0)).mean() - 0.01) < 0.003          # independent across seeds
    assert rng.hash24(0, np.array([0], np.uint64))[0] == 0           # the mixer maps 0 to 0

    # known answers from an independent pure-Python statement of the mixer (csrc/common.h:tecm_hash24)
    def ref(seed, idx):
        m = 0xFFFFFFFF
        x = ((idx & m) + ((idx >> 32) & 0xFFFFFF) * 0x9E3779 + (seed & m) + (seed >> 32) * 0x9E3779B1) & m
        x ^= x >> 16
        x = (x * 0x7FEB352D) & m
        x ^= x >> 15
        x = (x * 0x846CA68B) & m
        x ^= x >> 16
        return x >> 8
    for seed, i in ((0, 1), (12345, 0), (0x9E3779B97F4A7C15, 7), (2 ** 64 - 1, 2 ** 32 + 5), (77, 2 ** 40 + 123456789)):
        assert int(rng.hash24(seed, np.array([i], np.uint64))[0]) == ref(seed, i)
    # the high words matter: an index 2^32 further on, or a seed differing in its high word only, is another stream
    lo = rng.keep_mult(12345, idx, 0.1)
    assert not np.array_equal(lo, rng.keep_mult(12345, idx + np.uint64(1 << 32), 0.1))
    c = rng.keep_mult(12345 + (1 << 32), idx, 0.1)
    assert not np.array_equal(lo, c) and abs(((lo == 0) & (c == 0)).mean() - 0.01) < 0.003
    # no serial structure a dropout mask would care about: neighbours along a row and across rows of 768 / 3072 columns
    for lag in (1, 2, 4, 768, 3072):
        assert abs(((a[:-lag] == 0) & (a[lag:] == 0)).mean() - 0.01) < 0.003, lag


def test_clip_matches_torch_clip_grad_norm():
    from tecmollm.train import clip_flat_, flatten_grads
    lin = torch.nn.Linear(7, 5)
    flat = flatten_grads(lin.parameters())
    flat.copy_(torch.randn(flat.numel(), generator=torch.Generator().manual_seed(0)) * 3)
    ref = [p.grad.clone() for p in lin.parameters()]
    total = torch.nn.utils.clip_grad_norm_([torch.nn.Parameter(torch.zeros_like(g)) for g in ref], 1.0)
    lin2 = torch.nn.Linear(7, 5)
    for p, g in zip(lin2.parameters(), ref):
        p.grad = g.clone()
    tn = torch.nn.utils.clip_grad_norm_(lin2.parameters(), 1.0)
    got = clip_flat_(flat, 1.0)
    assert torch.allclose(got, tn) and total == 0
    for p, q in zip(lin.parameters(), lin2.parameters()):
        assert torch.allclose(p.grad, q.grad, rtol=1e-6, atol=1e-7)


class _Toy(torch.nn.Module):
    """CPU stand-in with the TEC_MoLLM call signature, for the data-parallel plumbing test."""

    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(6, 4)
        self.b = torch.nn.Linear(4, 12)
        self.frozen = torch.nn.Parameter(torch.ones(3), requires_grad=False)

    def forward(self, x, tf, ei, ew=None):
        h = torch.tanh(self.a(x)).mean(1)                  # (B, N, 4)
        return self.b(h).permute(0, 2, 1).unsqueeze(-1)    # (B, 12, N, 1)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tecmollm.train import TrainStep
        torch.manual_seed(0)
        model = _Toy()
        ts = TrainStep(model, world_size=world, fused_huber=False, optimizer="torch")
        g = torch.Generator().manual_seed(100)
        X = torch.randn(4, 5, 9, 6, generator=g)
        Y = torch.randn(4, 12, 9, 1, generator=g)
        sl = slice(rank * 2, rank * 2 + 2)                 # each rank takes its shard of the global batch
        for _ in range(3):
            ts.step(X[sl], None, None, None, Y[sl])
        out[rank] = torch.cat([p.detach().flatten() for p in model.parameters()])
    finally:
        dist.destroy_process_group()


def test_data_parallel_step_two_ranks_equals_single_rank_on_global_batch():
    """world_size=2 over gloo: one all-reduce of the flat gradient per step reproduces the single-process
    step on the concatenated batch (mean reduction => mean of per-rank gradients)."""
    from tecmollm.train import TrainStep
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_dp_worker, args=(2, port, out), nprocs=2, join=True)
    assert torch.allclose(out[0], out[1], rtol=0, atol=0)          # ranks stay bit-identical
    torch.manual_seed(0)
    model = _Toy()
    ts = TrainStep(model, world_size=1, fused_huber=False, optimizer="torch")
    g = torch.Generator().manual_seed(100)
    X = torch.randn(4, 5, 9, 6, generator=g)
    Y = torch.randn(4, 12, 9, 1, generator=g)
    for _ in range(3):
        ts.step(X, None, None, None, Y)
    single = torch.cat([p.detach().flatten() for p in model.parameters()])
    assert torch.allclose(out[0], single, rtol=1e-5, atol=1e-6)


def _dp8_worker(rank, world, port, out, odd_rank):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tecmollm.train import TrainStep
        torch.manual_seed(0 if rank != odd_rank else 999)  # odd_rank >= 0: that one rank starts from other weights
        model = _Toy()
        ts = TrainStep(model, world_size=world, fused_huber=False, optimizer="torch", broadcast_init=odd_rank < 0)
        assert ts.flat_grad_ext.numel() - ts.flat_grad.numel() == 2 * world      # the checksum tail: 2 floats per rank
        g = torch.Generator().manual_seed(100)
        X = torch.randn(world, 5, 9, 6, generator=g)
        Y = torch.randn(world, 12, 9, 1, generator=g)
        try:
            for _ in range(3):
                ts.step(X[rank:rank + 1], None, None, None, Y[rank:rank + 1])
            out[rank] = ("ok", torch.cat([p.detach().flatten() for p in model.parameters()]))
        except RuntimeError as e:
            out[rank] = ("diverged", str(e))
    finally:
        dist.destroy_process_group()


def test_data_parallel_step_eight_ranks_one_sample_each():
    """The reference's widest configuration (BASELINE configs[3]: 8 ranks): world_size 8 over gloo, one sample per rank --
    the ONE all-reduce per step carries the flat gradient plus a 16-float checksum tail; the ranks stay bit-identical and
    reproduce the single-process step on the 8-sample batch; a single rank (5) that starts from other weights is caught
    by every rank on the first step."""
    from tecmollm.train import TrainStep
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_dp8_worker, args=(8, _free_port(), out, -1), nprocs=8, join=True)
    assert all(out[r][0] == "ok" for r in range(8))
    for r in range(1, 8):
        assert torch.equal(out[r][1], out[0][1])
    torch.manual_seed(0)
    model = _Toy()
    ts = TrainStep(model, world_size=1, fused_huber=False, optimizer="torch")
    g = torch.Generator().manual_seed(100)
    X, Y = torch.randn(8, 5, 9, 6, generator=g), torch.randn(8, 12, 9, 1, generator=g)
    for _ in range(3):
        ts.step(X, None, None, None, Y)
    single = torch.cat([p.detach().flatten() for p in model.parameters()])
    assert torch.allclose(out[0][1], single, rtol=1e-5, atol=1e-6)
    out2 = mgr.dict()
    mp.spawn(_dp8_worker, args=(8, _free_port(), out2, 5), nprocs=8, join=True)
    assert all(out2[r][0] == "diverged" for r in range(8))


def _sync_worker(rank, world, port, out, broadcast_init):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tecmollm.train import TrainStep
        torch.manual_seed(100 + rank)                      # the ranks start from DIFFERENT weights
        model = _Toy()
        ts = TrainStep(model, world_size=world, fused_huber=False, optimizer="torch", broadcast_init=broadcast_init)
        g = torch.Generator().manual_seed(7)
        X, Y = torch.randn(4, 5, 9, 6, generator=g), torch.randn(4, 12, 9, 1, generator=g)
        sl = slice(rank * 2, rank * 2 + 2)
        try:
            for _ in range(2):
                ts.step(X[sl], None, None, None, Y[sl])
            out[rank] = ("ok", torch.cat([p.detach().flatten() for p in model.parameters()]))
        except RuntimeError as e:
            out[rank] = ("diverged", str(e))
    finally:
        dist.destroy_process_group()


def test_rank_divergence_is_detected_and_initial_broadcast_prevents_it():
    """train.py:354: the DDP constructor broadcasts rank 0's parameters.  TrainStep does the same (broadcast_init), and
    every optimizer step carries the per-rank parameter checksums in the tail of the ONE gradient all-reduce: ranks
    that started from different weights are caught on the first step when the broadcast is switched off."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_sync_worker, args=(2, _free_port(), out, True), nprocs=2, join=True)
    assert out[0][0] == "ok" and out[1][0] == "ok" and torch.equal(out[0][1], out[1][1])
    out2 = mgr.dict()
    mp.spawn(_sync_worker, args=(2, _free_port(), out2, False), nprocs=2, join=True)
    assert out2[0][0] == "diverged" and out2[1][0] == "diverged"
    assert "identical parameters" in out2[0][1]


def test_accumulation_boundaries_restart_every_epoch_like_the_reference_loop():
    """train.py:92-126 counts (i + 1) % accumulation_steps PER EPOCH and flushes a trailing partial cycle: with 10
    batches and accumulation 4 every epoch makes 2 full updates + 1 flush = 3 optimizer and scheduler steps, and
    epoch 2 starts a fresh cycle (it must not fire after 2 batches because epoch 1 left 2 over)."""
    from tecmollm.loop import train_one_epoch
    from tecmollm.train import TrainStep

    class _DS:
        def __init__(self, n):
            g = torch.Generator().manual_seed(3)
            self.X, self.Y = torch.randn(n, 5, 9, 6, generator=g), torch.randn(n, 12, 9, 1, generator=g)

        def __len__(self):
            return len(self.X)

        def batch(self, idx):
            return self.X[idx], None, self.Y[idx]

    torch.manual_seed(0)
    model = _Toy()
    ts = TrainStep(model, accumulation_steps=4, fused_huber=False, optimizer="torch")
    fired = []
    orig = ts.optimizer.step
    ts.optimizer.step = lambda *a, **k: (fired.append(ts._seen), orig(*a, **k))[1]
    ts._seen = 0
    step0 = ts.step

    def counting_step(*a):
        ts._seen += 1
        return step0(*a)
    ts.step = counting_step
    ds = _DS(10)
    ei = torch.zeros(2, 0, dtype=torch.int64)
    for _ in range(3):
        train_one_epoch(ts, ds, ei, batch_size=1)
    # optimizer steps after micro-batches 4, 8, 10 of every epoch (reference: i+1 in {4, 8} then the trailing flush)
    assert fired == [4, 8, 10, 14, 18, 20, 24, 28, 30]
    assert ts.scheduler.last_epoch == 9
    assert float(ts.flat_grad.abs().sum()) == 0.0
    ts.finish_accumulation()                               # nothing accumulated: no spurious AdamW / scheduler step
    assert len(fired) == 9 and ts.scheduler.last_epoch == 9


def test_accumulation_fires_optimizer_on_boundary_only():
    from tecmollm.train import TrainStep
    torch.manual_seed(0)
    model = _Toy()
    ts = TrainStep(model, accumulation_steps=3, fused_huber=False, optimizer="torch")
    before = model.a.weight.detach().clone()
    g = torch.Generator().manual_seed(1)
    X, Y = torch.randn(2, 5, 9, 6, generator=g), torch.randn(2, 12, 9, 1, generator=g)
    ts.step(X, None, None, None, Y)
    ts.step(X, None, None, None, Y)
    assert torch.equal(model.a.weight, before) and float(ts.flat_grad.abs().sum()) > 0
    ts.step(X, None, None, None, Y)
    assert not torch.equal(model.a.weight, before) and float(ts.flat_grad.abs().sum()) == 0


def test_cosine_warm_restarts_closed_form_equals_torch_scheduler():
    """CosineAnnealingWarmRestarts(T_0=10, T_mult=2, eta_min=1e-7) stepped once per update (train.py:366, :108)."""
    from tecmollm.optim import CosineWarmRestarts
    for base, T0, Tm in [(1e-4, 10, 2), (3e-3, 4, 1), (1e-4, 7, 3)]:
        opt = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=base)
        ref = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, T_0=T0, T_mult=Tm, eta_min=1e-7)
        mine = CosineWarmRestarts(base, T0, Tm, 1e-7)
        for _ in range(200):
            assert abs(mine.lr - opt.param_groups[0]["lr"]) <= 1e-18 + 1e-12 * base
            opt.step()
            ref.step()
            mine.step()
        st = mine.state_dict()
        again = CosineWarmRestarts(base, T0, Tm, 1e-7)
        again.load_state_dict(st)
        assert again.lr == mine.lr and again.step() == mine.step()


def test_checkpoint_prefix_stripping_follows_test_py():
    from tecmollm.checkpoint import strip_wrapper_prefixes
    sd = {"module._orig_mod.a.weight": 1, "module.b.bias": 2, "_orig_mod.c": 3, "d.module.e": 4}
    assert strip_wrapper_prefixes(sd) == {"a.weight": 1, "b.bias": 2, "c": 3, "d.module.e": 4}


def test_native_train_step_refuses_cpu_parameters():
    from tecmollm import TecmError
    from tecmollm.train import TrainStep
    with pytest.raises(TecmError):
        TrainStep(_Toy())                      # optimizer="native" is the default and is HIP-only


def test_by_source_lists_are_a_permutation_of_each_tile_segment():
    """Every by-target CSR entry of a tile appears exactly once in the tile's by-source lists, under its source row,
    carrying its (tile target, position in the tile's by-target segment) -- what the backward's gather pass relies on."""
    from tecmollm import graph as G
    ei, _ = R.grid_graph(7, 9, threshold_km=900.0)
    rowptr, col = G.csr_by_target(ei.numpy(), 63)
    for tn in (16, 5):
        lo, hi = G.tile_windows(rowptr, col, 63, tn)
        sp, sc, so = G.by_source_lists(rowptr, col, 63, tn, lo, hi)
        for k in range(lo.size):
            n0, n1 = k * tn, min(63, (k + 1) * tn)
            e0, e1 = int(rowptr[n0]), int(rowptr[n1])
            W = int(hi[k] - lo[k])
            ptr = sp[so[k]:so[k] + W + 1]
            assert ptr[0] == 0 and ptr[-1] == e1 - e0 and np.all(np.diff(ptr) >= 0)
            seen = set()
            for w in range(W):
                for q in range(ptr[w], ptr[w + 1]):
                    code = int(sc[e0 + q])
                    t_rel, pos = code >> 16, code & 0xFFFF
                    e = e0 + pos                                         # the by-target entry this refers to
                    assert e0 <= e < e1 and int(col[e]) == int(lo[k]) + w
                    assert rowptr[n0 + t_rel] <= e < rowptr[n0 + t_rel + 1]   # ... and it really enters that target
                    seen.add(e)
            assert seen == set(range(e0, e1))


def test_grid_graph_matches_reference_graph_constructor(golden_dir):
    """src/graph/graph_constructor.py (product, banded search) and the oracle's grid_graph against edges and weights
    produced by the reference's own calculate_haversine_distance_matrix / construct_binary_adjacency /
    symmetrically_normalize_adjacency (oracle/make_golden_shell.py), full 41 x 71 grid and a small half-degree grid."""
    from src.graph import graph_constructor as GC
    g = np.load(os.path.join(golden_dir, "shell_graph.npz"))
    for tag in ("full", "small"):
        lat, lon, thr = g[f"{tag}_lat"], g[f"{tag}_lon"], float(g[f"{tag}_thr"])
        ei, ew = GC.build_grid_graph(lat, lon, thr)
        assert np.array_equal(ei.numpy(), g[f"{tag}_edge_index"].astype(np.int64))          # same edges, same order
        np.testing.assert_allclose(ew.numpy(), g[f"{tag}_edge_weight"], rtol=1e-6)
    # the dense helpers of the mirrored API agree with the banded builder
    lat, lon, thr = g["small_lat"], g["small_lon"], float(g["small_thr"])
    row, col, data = GC.symmetrically_normalize_adjacency(
        GC.construct_binary_adjacency(GC.calculate_haversine_distance_matrix(lat, lon), thr))
    assert np.array_equal(np.vstack((row, col)), g["small_edge_index"])
    np.testing.assert_allclose(data, g["small_edge_weight"], rtol=1e-6)
    # the oracle's own builder (used by the parity tests) is the same graph
    ei_o, ew_o = R.grid_graph()
    assert np.array_equal(ei_o.numpy(), g["full_edge_index"].astype(np.int64))
    np.testing.assert_allclose(ew_o.numpy(), g["full_edge_weight"], rtol=1e-6)
    assert ei_o.shape[1] == 20924


def test_bench_multi_gpu_launcher_fails_loudly_without_gpus():
    """`python bench.py --gpus 2` starts its own ranks; on a box with fewer than 2 visible GPUs it must exit non-zero
    with a message instead of silently measuring one device (and without initialising a GPU in the launcher)."""
    import subprocess
    import sys
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TECM_DIST_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    # the launcher counts GPUs from sysfs; where the KFD topology is not readable (this container) the ranks refuse
    assert r.returncode != 0 and not r.stdout.strip()
    assert "visible GPUs" in r.stderr or "needs MI355X GPUs" in r.stderr
    # a readable topology with ONE GPU: the launcher itself refuses, before starting any rank
    import tempfile
    with tempfile.TemporaryDirectory() as topo:
        for i, simd in enumerate((0, 1024)):
            os.makedirs(os.path.join(topo, str(i)))
            open(os.path.join(topo, str(i), "properties"), "w").write(f"simd_count {simd}\n")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                           env=dict(env, TECM_KFD_TOPOLOGY=topo), capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "needs 2 visible GPUs" in r.stderr and not r.stdout.strip()
    # under torchrun-style env with a mismatching --gpus the rank refuses as well
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"],
                       env=dict(env, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_synthetic_inputs_do_not_come_from_the_oracle():
    """bench.py and tools/ draw their inputs from the product package; the oracle keeps an identical generator for
    the parity tests.  Outside tests/, only bench.py's cpu_baseline leg and smoke() may touch oracle/."""
    from tecmollm import synthetic
    a, b = synthetic.synthetic_batch(2, 4, 5, 3, 2, seed=9), R.synthetic_batch(2, 4, 5, 3, 2, seed=9)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def main():"):]
    assert "oracle" not in body.replace("cpu_baseline", "")       # main() reaches the oracle only through cpu_baseline()
    for root, _, files in os.walk(os.path.join(ROOT, "tec-mollm_amd")):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(root, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, os.path.join(root, f)


def test_bench_counts_gpus_from_sysfs_without_the_hip_runtime(tmp_path, monkeypatch):
    """bench.py's launcher parent must never touch the HIP runtime: it counts GPU nodes of the KFD topology in sysfs
    (CPU nodes have simd_count 0) and honours the *_VISIBLE_DEVICES variables."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for i, simd in enumerate((0, 0, 1024, 1024, 1024)):              # two CPU sockets, three GPUs
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {64 if simd == 0 else 0}\nsimd_count {simd}\nmem_banks_count 1\n")
    monkeypatch.setenv("TECM_KFD_TOPOLOGY", str(tmp_path))
    for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    assert bench.visible_gpus() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.visible_gpus() == 2
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "")
    assert bench.visible_gpus() == 0
    monkeypatch.setenv("TECM_KFD_TOPOLOGY", str(tmp_path / "missing"))
    assert bench.visible_gpus() is None
    src = open(os.path.join(ROOT, "bench.py")).read()
    launcher = src[src.index("def launch_ranks"):src.index("def make_config")]
    assert "torch.cuda" not in launcher                               # the parent never asks the runtime


def test_isa_audit_of_hand_issued_lds_reads_catches_early_uses():
    """__graft_entry__.audit_hand_issued_lds_reads (run by build() on gemm_bf16_tn.hip's ISA): a register a hand-issued
    ds_read_b64_tr_b16 writes may not be touched before the counted s_waitcnt that retires it."""
    import __graft_entry__ as ge
    ok = """
_Zkernel:
	ds_read_b64_tr_b16 v[10:11], v2 offset:64
	ds_read_b64_tr_b16 v[12:13], v2 offset:128
	v_add_u32_e32 v3, 1, v2
	s_waitcnt lgkmcnt(1)
	v_mfma_f32_32x32x16_bf16 a[0:15], v[10:11], v[20:21], a[0:15]
	s_waitcnt lgkmcnt(0)
	v_mov_b32_e32 v14, v12
	s_endpgm
"""
    assert ge.audit_hand_issued_lds_reads(ok, "ok") == 2
    early_use = ok.replace("v_add_u32_e32 v3, 1, v2", "v_mov_b32_e32 v3, v13")
    copied = ok.replace("s_waitcnt lgkmcnt(1)", "s_waitcnt lgkmcnt(2)")            # the first read is not retired yet
    smem = ok.replace("v_add_u32_e32 v3, 1, v2", "s_load_dwordx2 s[4:5], s[0:1], 0x0")
    across = ok.replace("s_waitcnt lgkmcnt(1)", ".LBB0_1:")
    for bad in (early_use, copied, smem, across):
        with pytest.raises(RuntimeError):
            ge.audit_hand_issued_lds_reads(bad, "bad")

def test_bench_line_stays_under_the_drivers_parse_budget():
    """The driver parses the LAST stdout line of bench.py out of ~8 KB of kept tail.  Round 4's line (committed as
    profiles/r04_bench_default.json: 24 KB with per-shape tables) was cut -> parsed: null.  compact_line() must turn that
    very line -- and an 8-rank one with per-rank lists -- into < 4 KB carrying every contract key, with the tables moved
    to the detail object."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_default.json")))
    assert len(json.dumps(full)) > 20000
    head, detail = bench.compact_line(full)
    text = json.dumps(head)
    assert len(text) < bench.LINE_BUDGET == 4096
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity", "configs_extra"):
        assert head[k] == full[k] or k in ("roofline", "cpu_baseline", "parity", "configs_extra", "config"), k
    rf = head["roofline"]
    assert rf["frac"] == full["roofline"]["frac"] and rf["traffic"] == full["roofline"]["traffic"]["hbm_bytes_per_launch"]
    assert rf["step"]["frac_of_peak"] == full["roofline"]["step"]["frac_of_peak"] and "shapes" not in rf and "non_gemm" not in rf
    assert head["cpu_baseline"]["cores"] == 16 and head["cpu_baseline"]["kind"] == "port"
    assert head["configs_extra"]["bf16"]["roofline"]["frac"] == full["configs_extra"]["bf16"]["roofline"]["frac"]
    assert detail["roofline"]["shapes"] == full["roofline"]["shapes"]
    assert detail["configs_extra"]["L96"]["roofline"]["non_gemm"] == full["configs_extra"]["L96"]["roofline"]["non_gemm"]
    # an 8-rank line: per-rank lists must not ride in the headline
    eight = {k: v for k, v in full.items() if k not in ("configs_extra", "cpu_baseline", "parity")}
    eight["n_gpus"] = 8
    eight["config"] = dict(full["config"], dist={
        "backend": "nccl", "world_size": 8, "allreduce_of_ones": 8.0, "launcher": "torchrun", "param_checksum_min_eq_max": True,
        "allreduce_ms": {"mean_over_ranks": 0.5, "max_over_ranks": 0.9, "per_rank_mean": [0.5] * 8, "bytes": 12327984,
                         "launches": 10, "note": "x" * 200},
        "step_ms_per_rank": {"mean": [64.0] * 8, "min_over_ranks": 63.9, "max_over_ranks": 64.2, "slowest_single_step": 64.9}})
    h8, d8 = bench.compact_line(eight)
    assert len(json.dumps(h8)) < 4096 and "per_rank_mean" not in json.dumps(h8)
    assert h8["config"]["dist"]["allreduce_ms"]["bytes"] == 12327984 and h8["config"]["dist"]["world_size"] == 8
    assert d8["dist"]["allreduce_ms"]["per_rank_mean"] == [0.5] * 8
