"""GPU parity of the shell around the step through the C ABI:
fused clip + AdamW vs torch's clip_grad_norm_/AdamW/CosineAnnealingWarmRestarts, device metrics vs the
reference's metrics.py (golden vectors + numpy oracle), window batches vs the reference Dataset."""
import os

import numpy as np
import pytest
import torch

from oracle import shell_cpu as S

pytestmark = pytest.mark.gpu

KEYS = ("mae_avg", "rmse_avg", "r2_score_avg", "pearson_r_avg", "mae_by_horizon", "rmse_by_horizon", "r2_by_horizon",
        "pearson_by_horizon")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda")


def _model(cfg, seed, dev, gat="per_timestep"):
    from tests.parity import build_model
    from oracle import ref_cpu as R
    return build_model(cfg, R.init_params(cfg, seed=seed), dev, gat)


def _inputs(cfg, B, grid, seed, dev):
    from oracle import ref_cpu as R
    N = grid[0] * grid[1]
    x, tf, y = R.synthetic_batch(B, cfg["temporal_seq_len"], N, cfg["spatial_in_channels_base"],
                                 cfg["prediction_horizon"], seed=seed)
    tfd = tf[:, :, 0, :].contiguous().to(dev).unsqueeze(-2).expand(B, cfg["temporal_seq_len"], N, 4)
    return x.to(dev), tfd, R.grid_graph(*grid)[0].to(dev), y.to(dev)


# ------------------------------------------------------------------------------------ optimizer
@pytest.mark.parametrize("shapes,scale", [([(7, 5), (5,), (3, 4, 2), (1,)], 5.0),      # 65 values: scalar tail path
                                          ([(257, 33), (33,), (1024,)], 0.01),          # below max_norm: no clipping
                                          ([(768, 96), (2304, 32), (13, 16)], 3.0)])
def test_fused_clip_adamw_matches_torch(dev, shapes, scale):
    from tecmollm.optim import CosineWarmRestarts, FlatAdamW
    g = torch.Generator().manual_seed(3)
    params0 = [torch.randn(*s, generator=g) for s in shapes]
    steps = 25                                             # crosses the first warm restart (T_0 = 10) and the second
    grads = [[torch.randn(*s, generator=g) * scale / (1 + 0.1 * k) for s in shapes] for k in range(steps)]
    want, norms, lrs = S.reference_optimizer_steps(params0, grads, lr=1e-3, weight_decay=1e-2, max_norm=1.0)

    ps = [torch.nn.Parameter(p.clone().to(dev)) for p in params0]
    opt = FlatAdamW(ps, lr=1e-3, weight_decay=1e-2)
    sched = CosineWarmRestarts(1e-3)
    for k in range(steps):
        for p, gk in zip(ps, grads[k]):
            p.grad.copy_(gk)                               # writes into the flat buffer through the view
        assert abs(sched.lr - lrs[k]) <= 1e-12 + 1e-9 * lrs[k]
        tn = opt.step(lr=sched.lr, max_norm=1.0)
        assert abs(float(tn) - norms[k]) <= 2e-6 * norms[k]
        assert float(opt.flat_grad.abs().sum()) == 0.0     # zero_grad fused
        sched.step()
    for p, w in zip(ps, want):
        torch.testing.assert_close(p.detach().cpu(), w, rtol=2e-5, atol=2e-7)


def test_fused_adamw_grad_scale_is_the_data_parallel_mean(dev):
    """grad_scale = 1/world on the SUM-reduced buffer equals clipping/updating the mean gradient."""
    from tecmollm.optim import FlatAdamW
    g = torch.Generator().manual_seed(5)
    p0 = torch.randn(1000, generator=g)
    g1, g2 = torch.randn(1000, generator=g), torch.randn(1000, generator=g)
    want, norms, _ = S.reference_optimizer_steps([p0], [[(g1 + g2) / 2]], lr=1e-4)
    p = torch.nn.Parameter(p0.clone().to(dev))
    opt = FlatAdamW([p])
    p.grad.copy_(g1 + g2)
    tn = opt.step(max_norm=1.0, grad_scale=0.5)
    assert abs(float(tn) - norms[0]) <= 2e-6 * norms[0]
    torch.testing.assert_close(p.detach().cpu(), want[0], rtol=1e-6, atol=1e-8)


def test_flat_adamw_state_dict_round_trips_through_torch_adamw(dev):
    from tecmollm.optim import FlatAdamW
    g = torch.Generator().manual_seed(9)
    shapes = [(6, 4), (4,)]
    ps = [torch.nn.Parameter(torch.randn(*s, generator=g).to(dev)) for s in shapes]
    opt = FlatAdamW(ps, lr=1e-3)
    for _ in range(3):
        for p in ps:
            p.grad.copy_(torch.randn(*p.shape, generator=g))
        opt.step(max_norm=0.0)
    sd = opt.state_dict()
    tps = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    topt = torch.optim.AdamW(tps, lr=1e-3, weight_decay=1e-2)
    topt.load_state_dict(sd)                                # torch accepts the exported state
    ps2 = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt2 = FlatAdamW(ps2, lr=1e-3)
    opt2.load_state_dict(topt.state_dict())                 # and the flat optimizer accepts torch's
    gk = [torch.randn(*s, generator=g) for s in shapes]
    for p, p2, tp, gg in zip(ps, ps2, tps, gk):
        p.grad.copy_(gg)
        p2.grad.copy_(gg)
        tp.grad = gg.to(dev)
    opt.step(max_norm=0.0)
    opt2.step(max_norm=0.0)
    topt.step()
    for p, p2, tp in zip(ps, ps2, tps):
        assert torch.equal(p, p2)
        torch.testing.assert_close(p.detach(), tp.detach(), rtol=2e-6, atol=1e-8)


def test_optimizer_rejects_cpu_parameters_and_absorbs_foreign_grads(dev):
    from tecmollm import TecmError
    from tecmollm.optim import FlatAdamW
    with pytest.raises(TecmError):
        FlatAdamW([torch.nn.Parameter(torch.zeros(4))])
    p = torch.nn.Parameter(torch.zeros(8, device=dev))
    opt = FlatAdamW([p])
    p.grad = torch.ones(8, device=dev)                      # a gradient stored outside the flat buffer is absorbed
    opt.step(max_norm=0.0, zero_grad=False)
    assert p.grad is None and float(opt.flat_grad.sum()) == 8.0


def test_train_step_native_equals_torch_optimizer_path(dev):
    """Whole step on the model: native fused optimizer vs torch AdamW on identical gradients."""
    from oracle import ref_cpu as R
    from tecmollm.train import TrainStep
    cfg = R.default_config(L_in=48, L_out=12, num_nodes=12, llm_layers=1)
    outs = []
    for kind in ("native", "torch"):
        model = _model(cfg, 2, dev).eval()                  # dropout off: both runs see identical gradients
        x, tf, ei, y = _inputs(cfg, 2, (3, 4), 4, dev)
        ts = TrainStep(model, lr=1e-3, optimizer=kind, accumulation_steps=2)
        losses = [float(ts.step(x, tf, ei, None, y)) for _ in range(6)]
        outs.append((losses, torch.cat([q.detach().flatten() for q in ts.params]).cpu()))
    assert outs[0][0][0] == outs[1][0][0]
    np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=2e-4)
    torch.testing.assert_close(outs[0][1], outs[1][1], rtol=1e-3, atol=2e-5)
    assert outs[0][0][2] != outs[0][0][0]                   # parameters moved after the first boundary


def test_recorded_step_equals_the_eager_step_with_dropout_off(dev):
    """TrainStep.step_graphed (the micro-batch recorded once as a hipGraph and replayed; the optimizer stays outside) against
    TrainStep.step: same losses and parameters over six micro-batches with accumulation and changing inputs."""
    from oracle import ref_cpu as R
    from tecmollm.train import TrainStep
    cfg = R.default_config(L_in=48, L_out=12, num_nodes=12, llm_layers=1)
    outs = []
    for graphed in (False, True):
        model = _model(cfg, 2, dev).eval()
        x, tf, ei, y = _inputs(cfg, 2, (3, 4), 4, dev)
        ts = TrainStep(model, lr=1e-3, accumulation_steps=2)
        f = ts.step_graphed if graphed else ts.step
        losses = []
        for i in range(6):
            xi = x + 0.01 * i                                # the recorded input tensors must be refilled every call
            losses.append(float(f(xi, tf, ei, None, y)))
        outs.append((losses, ts.optimizer.flat_param.detach().clone().cpu()))
        if graphed:
            assert len(ts._graphs) == 1 and all(v not in (None, False) for v in ts._graphs.values())
    np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=1e-5)
    torch.testing.assert_close(outs[0][1], outs[1][1], rtol=1e-4, atol=2e-6)
    assert len(set(outs[1][0])) == 6


def test_recorded_step_draws_fresh_masks_every_replay_and_the_same_ones_in_forward_and_backward(dev):
    """Training mode: the recorded kernels add the step's device word to their seeds (TecmDrop::seed_dev).  With lr = 0 the
    loss of a replay is a function of that word alone: it differs from replay to replay, and setting the word back
    reproduces loss and gradients bit for bit (the spatial stage's float atomics aside) -- and an EAGER micro-batch run
    with the same word and plan gives the same gradients, i.e. forward and backward of a replay read one value."""
    from oracle import ref_cpu as R
    from src.model import modules as M
    from tecmollm import ops
    from tecmollm.train import TrainStep
    cfg = R.default_config(L_in=48, L_out=12, num_nodes=12, llm_layers=1)
    model = _model(cfg, 2, dev).train()
    x, tf, ei, y = _inputs(cfg, 2, (3, 4), 4, dev)
    ts = TrainStep(model, lr=0.0, weight_decay=0.0, accumulation_steps=1 << 30)       # gradients stay in the flat buffer
    ts.step_graphed(x, tf, ei, None, y)                      # eager warm-up
    ts.flat_grad.zero_()
    l1 = float(ts.step_graphed(x, tf, ei, None, y))          # records + first replay
    plan_count = M._seed_counter[0]
    w1 = int(ts._seed_word.item())
    g1 = ts.flat_grad.clone()
    ts.flat_grad.zero_()
    l2 = float(ts.step_graphed(x, tf, ei, None, y))
    g2 = ts.flat_grad.clone()
    assert w1 != 0 and int(ts._seed_word.item()) != w1 and l2 != l1 and not torch.equal(g1, g2)
    # back to the first word: the next replay advances it to w1 again
    ts._seed_word.fill_(0)
    ts.flat_grad.zero_()
    l3 = float(ts.step_graphed(x, tf, ei, None, y))
    assert int(ts._seed_word.item()) == w1 and l3 == l1
    torch.testing.assert_close(ts.flat_grad, g1, rtol=1e-4, atol=1e-7)
    # the eager micro-batch with the same plan (base seed) and the same word
    ts.flat_grad.zero_()
    M._seed_counter[0] = plan_count - 1
    word = torch.tensor([w1], device=dev, dtype=torch.int64)
    ops.SEED_WORD = word
    try:
        l4 = float(ts._micro_batch(x, tf, ei, None, y))
    finally:
        ops.SEED_WORD = None
    assert l4 == l1
    torch.testing.assert_close(ts.flat_grad, g1, rtol=1e-4, atol=1e-7)


# ------------------------------------------------------------------------------------ metrics
def _check(out, want, rtol=2e-5, atol=2e-6):
    for k in KEYS:
        np.testing.assert_allclose(np.asarray(out[k]), np.asarray(want[k]), rtol=rtol, atol=atol, err_msg=k)


def test_device_metrics_match_reference_golden(dev, golden_dir):
    from src.evaluation.metrics import HorizonMetrics
    g = np.load(os.path.join(golden_dir, "shell_metrics_scaled.npz"))
    hm = HorizonMetrics(g["y_true"].shape[1], (float(g["mean"]), float(g["scale"])), device=dev)
    yt, yp = torch.from_numpy(g["y_true"]).to(dev), torch.from_numpy(g["y_pred"]).to(dev)
    for a in range(0, yt.shape[0], 2):                      # streamed in batches of 2 like validate()
        hm.update(yp[a:a + 2], yt[a:a + 2])
    _check(hm.compute(), {k: g[f"out_{k}"] for k in KEYS})


def test_device_metrics_fallback_and_degenerate_horizons(dev, golden_dir):
    from src.evaluation.metrics import evaluate_horizons
    g = np.load(os.path.join(golden_dir, "shell_metrics_unscaled.npz"))
    out = evaluate_horizons(g["y_true"], g["y_pred"], None, device=dev)
    _check(out, {k: g[f"out_{k}"] for k in KEYS})
    assert out["pearson_by_horizon"][1] == 0.0 and out["pearson_by_horizon"][2] == 0.0 and out["r2_by_horizon"][3] == 1.0


def test_device_metrics_read_the_permuted_model_output_view(dev):
    from src.evaluation.metrics import HorizonMetrics
    g = torch.Generator().manual_seed(1)
    B, N, H = 3, 2911, 12
    pred_bnl = torch.randn(B, N, H, generator=g).to(dev)
    out_view = pred_bnl.permute(0, 2, 1).unsqueeze(-1)      # what TEC_MoLLM.forward returns
    y = torch.randn(B, H, N, 1, generator=g).to(dev)
    assert not out_view.is_contiguous()
    hm = HorizonMetrics(H, (21.5, 9.25), device=dev)
    hm.update(out_view, y)
    want = S.evaluate_horizons(y.cpu().numpy(), out_view.cpu().numpy(), 21.5, 9.25)
    _check(hm.compute(), want, rtol=1e-9, atol=1e-10)       # oracle and kernel both sum in float64


# ------------------------------------------------------------------------------------ sliding windows
def test_window_batches_match_reference_dataset_items(dev, golden_dir):
    from src.data.dataset import SlidingWindowSamplerDataset
    g = np.load(os.path.join(golden_dir, "shell_windows.npz"))
    ds = SlidingWindowSamplerDataset.from_tensors(torch.from_numpy(g["X"]), torch.from_numpy(g["Y"]),
                                                  torch.from_numpy(g["TF"]), int(g["L_in"]), int(g["L_out"]),
                                                  int(g["stride"]), device=dev)
    assert len(ds) == int(g["length"])
    pick = [int(i) for i in g["pick"]]
    x, tf, y = ds.batch(pick)
    T, Hh, Ww, Cc = g["X"].shape
    N = Hh * Ww
    assert x.shape == (len(pick), int(g["L_in"]), N, Cc) and y.shape == (len(pick), int(g["L_out"]), N, 1)
    assert tf.shape == (len(pick), int(g["L_in"]), N, 4) and tf.stride(2) == 0
    assert np.array_equal(x.cpu().numpy(), g["x"].reshape(len(pick), -1, N, Cc))                     # bit-exact copies
    assert np.array_equal(y.cpu().numpy(), np.transpose(g["y"], (0, 3, 1, 2)).reshape(len(pick), -1, N, 1))
    assert np.array_equal(tf[:, :, 0, :].cpu().numpy(), g["tf"])
    for j, i in enumerate(pick):                                                                     # per-item API
        it = ds[i]
        assert np.array_equal(it["x"].cpu().numpy(), g["x"][j]) and np.array_equal(it["y"].cpu().numpy(), g["y"][j])
    with pytest.raises(IndexError):
        ds[len(ds)]
    with pytest.raises(IndexError):
        ds.batch([0, len(ds)])
    short = SlidingWindowSamplerDataset.from_tensors(torch.from_numpy(g["X"]), torch.from_numpy(g["Y"]),
                                                     torch.from_numpy(g["TF"]), 38, 4, 1, device=dev)
    assert len(short) == int(g["length_when_too_short"]) == 0


def test_window_batch_full_grid_and_odd_row_width(dev):
    """41x71 grid at C=10 (float4 path) and a row width that is not a multiple of 4 (scalar path) vs the oracle."""
    from src.data.dataset import SlidingWindowSamplerDataset
    rng = np.random.default_rng(0)
    for (Hh, Ww, Cc, L_in, L_out) in [(41, 71, 10, 48, 12), (3, 3, 3, 5, 2)]:
        T = L_in + L_out + 9
        X = rng.standard_normal((T, Hh, Ww, Cc)).astype(np.float32)
        Y = rng.standard_normal((T, Hh, Ww, L_out)).astype(np.float32)
        TF = np.stack([rng.integers(0, hi, T) for hi in (12, 366, 13, 4)], 1).astype(np.float32)
        ref = S.SlidingWindows(X, Y, TF, L_in, L_out, 2)
        ds = SlidingWindowSamplerDataset.from_tensors(torch.from_numpy(X), torch.from_numpy(Y), torch.from_numpy(TF),
                                                      L_in, L_out, 2, device=dev)
        assert len(ds) == len(ref)
        idx = [len(ref) - 1, 0, 2]
        x, tf, y = ds.batch(idx)
        wx, wtf, wy = ref.batch(idx)
        assert np.array_equal(x.cpu().numpy(), wx) and np.array_equal(y.cpu().numpy(), wy)
        assert np.array_equal(tf.cpu().numpy(), wtf)


def test_window_batch_feeds_the_model(dev):
    """dataset -> batch -> TEC_MoLLM forward/backward: the expanded time-feature view and the (B,L_out,N,1)
    target are consumed as they come."""
    from src.data.dataset import SlidingWindowSamplerDataset
    from oracle import ref_cpu as R
    from tecmollm.train import TrainStep
    cfg = R.default_config(L_in=48, L_out=12, num_nodes=12, llm_layers=1)
    rng = np.random.default_rng(2)
    T = 80
    X = torch.from_numpy(rng.standard_normal((T, 3, 4, cfg["spatial_in_channels_base"])).astype(np.float32))
    Y = torch.from_numpy(rng.standard_normal((T, 3, 4, 12)).astype(np.float32))
    TF = torch.from_numpy(np.stack([rng.integers(0, 12, T), rng.integers(0, 366, T), rng.integers(0, 13, T),
                                    rng.integers(0, 4, T)], 1).astype(np.float32))
    ds = SlidingWindowSamplerDataset.from_tensors(X, Y, TF, 48, 12, device=dev)
    model = _model(cfg, 1, dev)
    ei = R.grid_graph(3, 4)[0].to(dev)
    ts = TrainStep(model)
    x, tf, y = ds.batch([0, 5, 11])
    l0 = float(ts.step(x, tf, ei, None, y))
    assert np.isfinite(l0)


# ------------------------------------------------------------------------------------ checkpoints
def test_checkpoint_round_trip_with_wrapper_prefixes(dev, tmp_path):
    from oracle import ref_cpu as R
    from tecmollm import checkpoint as ck
    from tecmollm.optim import FlatAdamW
    cfg = R.default_config(L_in=48, L_out=12, num_nodes=12, llm_layers=1)
    model = _model(cfg, 1, dev)
    path = str(tmp_path / "best_model.pth")
    ck.save_model(model, path)
    saved = torch.load(path, map_location="cpu")
    assert set(saved) == set(model.state_dict())
    # a DDP + torch.compile style file, as test.py:178-187 expects to meet
    torch.save({"module._orig_mod." + k: v + 1.0 if v.dtype.is_floating_point else v for k, v in saved.items()}, path)
    model2 = _model(cfg, 3, dev)
    opt = FlatAdamW([p for p in model2.parameters() if p.requires_grad])
    before = opt.flat_param.data_ptr()
    res = ck.load_model(model2, path, map_location="cpu")
    assert not res.missing_keys and not res.unexpected_keys
    for k, v in model2.state_dict().items():
        if v.dtype.is_floating_point:
            torch.testing.assert_close(v.cpu(), saved[k] + 1.0)
    assert opt.flat_param.data_ptr() == before and opt.params[0].data_ptr() == before      # flat views survive a load


# ------------------------------------------------------------------------------------ the reference's DDP wrapper
def _ddp_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import ref_cpu as R
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        cfg = R.default_config(L_in=16, L_out=12, num_nodes=6, llm_layers=1)
        model = _model(cfg, 3, dev).eval()                       # eval: dropout off, ranks differ only by their data
        ddp = torch.nn.parallel.DistributedDataParallel(model)   # train.py:354, default flags
        x, tf, ei, y = _inputs(cfg, 4, (2, 3), 7, dev)
        sl = slice(2 * rank, 2 * rank + 2)
        out_ = ddp(x[sl], tf[sl], ei, None)
        torch.nn.functional.huber_loss(out_, y[sl]).backward()
        torch.cuda.synchronize()
        out[rank] = {k: p.grad.detach().cpu() for k, p in model.named_parameters() if p.requires_grad}
    finally:
        dist.destroy_process_group()


def test_model_trains_under_the_reference_ddp_wrapper(dev):
    """nn.parallel.DistributedDataParallel(model) as train.py:354 builds it (two ranks share this GPU over gloo):
    every trainable parameter receives a gradient through the autograd hooks and the averaged gradients equal the
    single-process gradients of the concatenated batch."""
    import socket
    import torch.multiprocessing as mp
    from oracle import ref_cpu as R
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_ddp_worker, args=(2, port, out), nprocs=2, join=True)
    cfg = R.default_config(L_in=16, L_out=12, num_nodes=6, llm_layers=1)
    model = _model(cfg, 3, dev).eval()
    x, tf, ei, y = _inputs(cfg, 4, (2, 3), 7, dev)
    torch.nn.functional.huber_loss(model(x, tf, ei, None), y).backward()
    want = {k: p.grad.detach().cpu() for k, p in model.named_parameters() if p.requires_grad}
    assert set(out[0]) == set(want) and len(want) > 20
    for k, g in want.items():
        assert torch.equal(out[0][k], out[1][k]), k                 # DDP left both ranks with the same averaged grads
        torch.testing.assert_close(out[0][k], g, rtol=2e-4, atol=1e-6, msg=k)


# ------------------------------------------------------------------------------------ the product's own DP step
def _native_dp_worker(rank, world, port, out, mismatch):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import tecmollm
        from oracle import ref_cpu as R
        from tecmollm.train import TrainStep
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        cfg = R.default_config(L_in=16, L_out=12, num_nodes=6, llm_layers=1)
        # rank 1 is BUILT from other weights: TrainStep's initial broadcast (train.py:354) must make the ranks equal
        model = _model(cfg, 3 + (rank if not mismatch else 10 * rank), dev).eval()
        ts = TrainStep(model, lr=1e-3, world_size=world, broadcast_init=not mismatch)
        x, tf, ei, y = _inputs(cfg, 4, (2, 3), 7, dev)
        sl = slice(2 * rank, 2 * rank + 2)
        try:
            for _ in range(3):
                ts.step(x[sl], tf[sl], ei, None, y[sl])
            tecmollm.check_device_errors(dev, sync=True)
            out[rank] = ("ok", ts.optimizer.flat_param.detach().cpu(), int(ts.optimizer.flat_grad_ext.numel() - ts.optimizer.n))
        except RuntimeError as e:
            out[rank] = ("diverged", str(e))
    finally:
        dist.destroy_process_group()


def test_native_train_step_two_ranks_equals_single_process_on_the_global_batch(dev):
    """The product's own data-parallel path on the real model (two ranks share this GPU over gloo; RCCL runs the same
    code with backend nccl): flat gradient buffer, ONE all-reduce per optimizer step with the parameter checksums in its
    tail, the 1/world mean folded into the fused clip + AdamW kernel.  Three steps on the two halves of a batch must
    equal three single-process steps on the whole batch (Huber mean => mean of the per-rank gradients)."""
    import socket
    import torch.multiprocessing as mp
    from oracle import ref_cpu as R
    from tecmollm.train import TrainStep
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_native_dp_worker, args=(2, port, out, False), nprocs=2, join=True)
    assert out[0][0] == "ok" and out[1][0] == "ok", (out[0], out[1])
    assert torch.equal(out[0][1], out[1][1])                       # ranks stay bit-identical
    assert out[0][2] == 4                                          # 2 checksum slots per rank ride in the all-reduce
    cfg = R.default_config(L_in=16, L_out=12, num_nodes=6, llm_layers=1)
    model = _model(cfg, 3, dev).eval()
    ts = TrainStep(model, lr=1e-3, world_size=1)
    x, tf, ei, y = _inputs(cfg, 4, (2, 3), 7, dev)
    for _ in range(3):
        ts.step(x, tf, ei, None, y)
    single = ts.optimizer.flat_param.detach().cpu()
    # three AdamW steps (lr 1e-3) move a parameter by ~3e-3; the two-rank mean adds the two half-batch gradients in a
    # different order than the single batch, and Adam's g / sqrt(v) amplifies that on near-zero gradients
    torch.testing.assert_close(out[0][1], single, rtol=1e-3, atol=2e-5)
    moved = float((single - torch.cat([p.detach().flatten().cpu() for p in _model(cfg, 3, dev).parameters()
                                       if p.requires_grad])).abs().max())
    assert moved > 1e-4                                            # the three steps really updated the parameters


def test_native_train_step_reports_diverged_ranks(dev):
    """Without the initial broadcast, ranks built from different weights are caught by the checksum tail of the first
    gradient all-reduce (device error word -> RuntimeError at the next check, no synchronisation inside the step)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_native_dp_worker, args=(2, port, out, True), nprocs=2, join=True)
    assert out[0][0] == "diverged" and out[1][0] == "diverged", (out[0], out[1])
    assert "identical parameters" in out[0][1]


def test_bench_py_starts_its_own_ranks(dev):
    """`python bench.py --gpus 2` with no torchrun around it must measure 2 ranks (here over gloo on the one GPU of the
    box; on an N-GPU node the same launcher runs RCCL): n_gpus = 2, the collective saw 2 ranks, global batch 2 x B."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TECM_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    import tempfile
    side = os.path.join(tempfile.mkdtemp(), "detail.json")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "1", "--no-kernel-timing", "--detail-json", side], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    last = r.stdout.strip().splitlines()[-1]
    assert len(last) < 4096                             # what the driver parses: the last stdout line, compact
    line = json.loads(last)
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 2 and line["config"]["parallelism"] == "dp2"
    d = line["config"]["dist"]
    assert d["world_size"] == 2 and d["allreduce_of_ones"] == 2.0 and d["launcher"] == "bench.py"
    assert d["param_checksum_min_eq_max"] is True and line["value"] > 0
    # the collective is event-timed on every rank and the per-rank step times ride in the same line, so that a scaling
    # record can separate communication from compute
    ar = d["allreduce_ms"]
    assert ar["launches"] == 2 and 0 < ar["mean_over_ranks"] <= ar["max_over_ranks"]
    assert ar["bytes"] >= 4 * 3_000_000                 # 3.07 M trainable values at F = 10 / d_emb = 12
    sm = d["step_ms_per_rank"]
    assert 0 < sm["min_over_ranks"] <= sm["max_over_ranks"] <= sm["slowest_single_step"] * 1.0001
    full = json.load(open(side))["dist"]                # per-rank lists live in the side file, not in the parsed line
    assert len(full["allreduce_ms"]["per_rank_mean"]) == 2 and len(full["step_ms_per_rank"]["mean"]) == 2


def test_bench_py_four_rank_rehearsal_on_one_card(dev):
    """The widest rank count this pool's process guard admits on one card (at most 6 of our processes may hold the GPU:
    the test runner + 4 ranks; the launcher parent never touches it): `bench.py --gpus 4 --batch 1` over gloo -- four
    children from the sysfs-counting launcher, an 8-float checksum tail, rank-0-only stdout.  (The 8-rank form of the same
    host logic runs on CPU: tests/test_host.py::test_data_parallel_step_eight_ranks_one_sample_each.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TECM_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    import tempfile
    side = os.path.join(tempfile.mkdtemp(), "detail.json")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1",
                        "--batch", "1", "--no-kernel-timing", "--precision", "bf16", "--detail-json", side], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                              # rank 0 alone prints
    line = json.loads(lines[0])
    assert line["n_gpus"] == 4 and line["config"]["global_batch"] == 4 and line["config"]["parallelism"] == "dp4"
    d = line["config"]["dist"]
    assert d["world_size"] == 4 and d["allreduce_of_ones"] == 4.0 and d["param_checksum_min_eq_max"] is True
    full = json.load(open(side))["dist"]
    assert len(full["allreduce_ms"]["per_rank_mean"]) == 4 and len(full["step_ms_per_rank"]["mean"]) == 4
    assert len(lines[0]) < 4096
    assert d["allreduce_ms"]["bytes"] == 4 * (3_081_996 - 52_896 + (2911 + 12 + 366 + 13 + 4) * 12 + 2 * 4)


def _bench_line(extra, env_extra=None, launcher=None):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    cmd = (launcher or [sys.executable]) + [os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-other-precisions",
                                            *extra]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_n1_through_the_launcher_environment_reproduces_the_plain_run(dev):
    """The driver starts N = 1 as plain `python bench.py` and N > 1 through torch.distributed.run; the one-rank case of the
    launcher path (RANK=0, WORLD_SIZE=1 in the environment) must be the same measurement: within 2 %, best of two runs
    each (box noise on this pool is about 1 %)."""
    flags = ["--steps", "10", "--warmup", "3"]
    tr = {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29517"}
    plain = max(_bench_line(flags)["value"] for _ in range(2))
    under = max(_bench_line(flags, tr)["value"] for _ in range(2))
    assert abs(under - plain) / plain < 0.02, (plain, under)


def test_bench_window_feed_costs_less_than_one_percent(dev):
    """--data window: every step's batch drawn by tecm_window_batch from a resident series (train.py:57-65): the sampler
    is event-timed inside the timed region and must cost < 1 % of the step; throughput within 2 % of the fixed batch."""
    flags = ["--steps", "10", "--warmup", "3"]
    win = _bench_line(flags + ["--data", "window"])
    assert win["config"]["data_feed"] == "window"
    assert 0 < win["config"]["data_feed_share_of_step"] < 0.01, win["config"]
    fixed = _bench_line(flags)
    assert win["value"] > 0.98 * fixed["value"], (win["value"], fixed["value"])


def test_rccl_process_group_runs_on_this_gpu(dev):
    """backend "nccl" IS RCCL on ROCm.  A one-GPU box cannot host two RCCL ranks, but it can run the exact calls the
    N-rank path makes -- init_process_group("nccl", device_id=...), broadcast of the parameters, all-reduce of the
    flat gradient buffer with its checksum tail -- in a 1-rank communicator, in a fresh process as bench.py's ranks are."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(ROOT, "tec-mollm_amd")); sys.path.insert(0, ROOT)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
ones = torch.ones(1, device=dev); dist.all_reduce(ones)
assert dist.get_backend() == "nccl" and float(ones.item()) == 1.0
from oracle import ref_cpu as R
from tests.parity import build_model
from tecmollm.train import TrainStep, broadcast_parameters
cfg = R.default_config(L_in=16, L_out=12, num_nodes=12, llm_layers=1)
model = build_model(cfg, R.init_params(cfg, 0), dev, "per_timestep").train()
assert broadcast_parameters(model, 0) > 0
ts = TrainStep(model, lr=1e-3, accumulation_steps=1, world_size=1)
x, tf, y = R.synthetic_batch(2, 16, 12, 6, 12)
ei, _ = R.grid_graph(3, 4, threshold_km=170.0)
ts.step(x.to(dev), tf.to(dev), ei.to(dev), None, y.to(dev))
g = ts.flat_grad_ext.clone()
before = g.clone(); dist.all_reduce(g); torch.cuda.synchronize()
assert torch.equal(g, before)                         # SUM over one rank
dist.destroy_process_group()
print("RCCL_OK")
""".replace("ROOT", repr(root))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


# ------------------------------------------------------------------------------------ epoch loops
def test_epoch_loops_train_and_validate_on_a_learnable_series(dev):
    """train_one_epoch / validate (train.py:52-168) over the device dataset: a smooth synthetic series is learnable,
    so a few epochs must lower the validation loss; metrics come back with the reference's keys."""
    from oracle import ref_cpu as R
    from src.data.dataset import SlidingWindowSamplerDataset
    from tecmollm.loop import train_one_epoch, validate
    from tecmollm.train import TrainStep
    cfg = R.default_config(L_in=16, L_out=12, num_nodes=12, llm_layers=1)
    T, H, W, C = 120, 3, 4, cfg["spatial_in_channels_base"]
    t = torch.arange(T, dtype=torch.float32)
    base = torch.sin(2 * torch.pi * t / 24)[:, None, None] * (1 + 0.1 * torch.arange(H * W).view(H, W))
    X = base[..., None].repeat(1, 1, 1, C) + 0.01 * torch.randn(T, H, W, C, generator=torch.Generator().manual_seed(0))
    Y = torch.stack([torch.roll(base, -k - 1, 0) for k in range(12)], -1)          # the next 12 steps
    TF = torch.stack([t % 12, t % 366, torch.zeros(T), (t // 30) % 4], 1)
    ds = SlidingWindowSamplerDataset.from_tensors(X, Y, TF, 16, 12, device=dev)
    model = _model(cfg, 5, dev)
    ei = R.grid_graph(3, 4)[0].to(dev)
    ts = TrainStep(model, lr=2e-3, accumulation_steps=2)
    v0, m0 = validate(model, ds, ei, 16, scaler=(20.0, 8.0))
    losses = [train_one_epoch(ts, ds, ei, 16, order=torch.randperm(len(ds), generator=torch.Generator().manual_seed(e)).tolist())
              for e in range(4)]
    v1, m1 = validate(model, ds, ei, 16, scaler=(20.0, 8.0))
    assert v1 < 0.7 * v0 and losses[-1] < losses[0]
    assert set(m1) == {"mae_avg", "rmse_avg", "r2_score_avg", "pearson_r_avg", "mae_by_horizon", "rmse_by_horizon",
                       "r2_by_horizon", "pearson_by_horizon"} and len(m1["rmse_by_horizon"]) == 12
    assert m1["rmse_avg"] < m0["rmse_avg"]


def test_reference_training_loop_body_runs_unchanged(dev):
    """The statements of train.py:57-112 / :358-372 verbatim around the drop-in model: autocast(bf16), the per-step
    gradient_checkpointing_enable() call, GradScaler scale/unscale_/step/update, clip_grad_norm_, torch AdamW over the
    trainable parameters, CosineAnnealingWarmRestarts, accumulation_steps = 2."""
    from oracle import ref_cpu as R
    from torch.optim.lr_scheduler import CosineAnnealingWarmRestarts
    cfg = R.default_config(L_in=16, L_out=12, num_nodes=12, llm_layers=1)
    model = _model(cfg, 6, dev).train()
    optimizer = torch.optim.AdamW(filter(lambda p: p.requires_grad, model.parameters()), lr=1e-4, weight_decay=1e-2)
    scheduler = CosineAnnealingWarmRestarts(optimizer, T_0=10, T_mult=2, eta_min=1e-7)
    loss_fn = torch.nn.HuberLoss(delta=1.0)
    scaler = torch.amp.GradScaler("cuda")
    accumulation_steps = 2
    edge_index = R.grid_graph(3, 4)[0].to(dev)
    edge_weight = torch.ones(edge_index.shape[1], device=dev)
    before = torch.cat([p.detach().flatten().clone() for p in model.parameters() if p.requires_grad])
    optimizer.zero_grad()
    total_loss = 0.0
    for i in range(4):
        x5 = torch.randn(2, 16, 3, 4, cfg["spatial_in_channels_base"], device=dev)        # (B, L, H, W, C) as the dataset yields
        y = torch.randn(2, 3, 4, 12, device=dev)
        time_features = torch.randint(0, 4, (2, 16, 4), device=dev).float()
        B, L, H, W, C = x5.shape
        x = x5.view(B, L, H * W, C)
        time_features = time_features.unsqueeze(-2).expand(B, L, H * W, -1)
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            model.llm_backbone.model.gradient_checkpointing_enable()
            output = model(x, time_features, edge_index, edge_weight)
            y_reshaped = y.permute(0, 3, 1, 2).reshape(B, -1, H * W, 1)
            loss = loss_fn(output, y_reshaped)
            loss = loss / accumulation_steps
        scaler.scale(loss).backward()
        if (i + 1) % accumulation_steps == 0:
            scaler.unscale_(optimizer)
            torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
            scaler.step(optimizer)
            scaler.update()
            optimizer.zero_grad()
            scheduler.step()
        total_loss += loss.item() * accumulation_steps
    after = torch.cat([p.detach().flatten() for p in model.parameters() if p.requires_grad])
    assert np.isfinite(total_loss) and torch.isfinite(after).all() and not torch.equal(before, after)

def test_bench_default_command_prints_a_compact_parseable_last_line(dev, tmp_path):
    """What the driver runs at round end: `python bench.py` with NO flags.  Its LAST stdout line must be one JSON object
    under 4 KB (the driver keeps ~8 KB of stdout tail; round 4's 24 KB line was cut and recorded as parsed: null) with
    the contract's keys, a scalar roofline summary, cpu_baseline, parity and both configs_extra legs; the per-shape tables
    are in the side file."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", TECM_BENCH_DETAIL=str(tmp_path / "detail.json"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    last = r.stdout.strip().splitlines()[-1]
    assert len(last) < 4096, len(last)
    assert len(r.stdout) < 8000                          # nothing else on stdout pushes the line out of the kept tail
    line = json.loads(last)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity", "configs_extra"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["dtype"] == "f32" and line["value"] > 50 and "workload" in line["config"]
    rf = line["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launches", "avg_launch_ms", "share_of_step", "step"):
        assert k in rf, k
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0.3 < rf["frac"] < 1.0
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and 0 < cb["value"] < line["value"]
    assert line["parity"]["max_rel_err"] < 1e-3
    for leg in ("bf16", "L96"):
        e = line["configs_extra"][leg]
        assert e["samples_per_s"] > 0 and 0 < e["roofline"]["frac"] < 1 and "step" in e["roofline"]
    detail = json.load(open(tmp_path / "detail.json"))
    assert len(detail["roofline"]["shapes"]) >= 8 and detail["configs_extra"]["bf16"]["roofline"]["shapes"]
