"""GPU parity tests of the individual HIP kernels through the C ABI (via tecmollm.ops).
References here are plain torch fp64 ops on the same inputs (floating-point kernels), the dropout
masks come from the NumPy mirror of the device hash.  Tolerance: 1e-4 relative to the reference's
max magnitude (fp32 accumulate vs fp64), tighter than the 1e-3 the north star asks end to end."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 2e-4


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda")


def _rand(*shape, dev, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev)


# ----------------------------------------------------------------------------- GEMM, plain views
@pytest.mark.parametrize("M,N,K", [(300, 200, 96), (129, 130, 64), (77, 33, 50), (64, 31, 51), (1000, 12, 576),
                                   (257, 32, 768), (513, 768, 32)])
def test_gemm_mk_nk(dev, M, N, K):
    from tecmollm import ops
    A, B = _rand(M, K, dev=dev, seed=1), _rand(N, K, dev=dev, seed=2)
    bias = _rand(N, dev=dev, seed=3)
    C = torch.full((M, N), float("nan"), device=dev)
    ops.gemm(M, N, K, A, K, B, K, C, N, bias=bias)
    ref = A.double() @ B.double().t() + bias.double()
    assert _rel(C, ref) < TOL


@pytest.mark.parametrize("M,N,K", [(300, 200, 96), (130, 2304, 800), (77, 22, 192), (65, 50, 33)])
def test_gemm_mk_kn(dev, M, N, K):
    from tecmollm import ops
    A, B = _rand(M, K, dev=dev, seed=1), _rand(K, N, dev=dev, seed=2)
    C = torch.full((M, N), float("nan"), device=dev)
    ops.gemm(M, N, K, A, K, B, N, C, N, b_layout=ops.B_KN, alpha=0.5)
    assert _rel(C, 0.5 * (A.double() @ B.double())) < TOL


@pytest.mark.parametrize("Mo,No,K,split", [(64, 154, 3000, 1), (64, 168, 5000, 7), (12, 576, 999, 3),
                                           (2304, 32, 2000, 4), (32, 768, 1501, 1), (100, 75, 333, 2)])
def test_gemm_km_kn_splitk(dev, Mo, No, K, split):
    from tecmollm import ops
    A, B = _rand(K, Mo, dev=dev, seed=1), _rand(K, No, dev=dev, seed=2)
    C = torch.full((Mo, No), float("nan"), device=dev)
    ops.gemm(Mo, No, K, A, Mo, B, No, C, No, a_layout=ops.A_KM, b_layout=ops.B_KN, split_k=split)
    assert _rel(C, A.double().t() @ B.double()) < TOL


def test_gemm_strided_slices_and_accumulate(dev):
    """column slices of wider buffers (lda/ldc + offsets), residual, accumulate."""
    from tecmollm import ops
    M, K, N = 200, 64, 96
    Abuf, B = _rand(M, 100, dev=dev, seed=1), _rand(N, K, dev=dev, seed=2)
    Cbuf = _rand(M, 300, dev=dev, seed=3)
    C0 = Cbuf.clone()
    R = _rand(M, N, dev=dev, seed=4)
    ops.gemm(M, N, K, Abuf, 100, B, K, Cbuf, 300, a_off=36, c_off=100, residual=(R, N), accumulate=True)
    ref = C0.double().clone()
    ref[:, 100:100 + N] += Abuf[:, 36:36 + K].double() @ B.double().t() + R.double()
    assert _rel(Cbuf, ref) < TOL
    assert torch.equal(Cbuf[:, :100], C0[:, :100]) and torch.equal(Cbuf[:, 196:], C0[:, 196:])


def test_gemm_activations_preact_dact(dev):
    from tecmollm import ops
    M, K, N = 150, 64, 80
    A, B = _rand(M, K, dev=dev, seed=1, scale=0.3), _rand(N, K, dev=dev, seed=2, scale=0.3)
    for act, fn in ((ops.ACT_GELU_ERF, lambda t: torch.nn.functional.gelu(t)),
                    (ops.ACT_GELU_TANH, lambda t: torch.nn.functional.gelu(t, approximate="tanh"))):
        C = torch.empty(M, N, device=dev)
        pre = torch.empty(M, N, device=dev)
        ops.gemm(M, N, K, A, K, B, K, C, N, act=act, preact=(pre, N))
        z = A.double() @ B.double().t()
        assert _rel(pre, z) < TOL and _rel(C, fn(z)) < TOL
        # backward through the activation: out = (A.B^T) * act'(src)
        src = _rand(M, N, dev=dev, seed=5)
        D = torch.empty(M, N, device=dev)
        ops.gemm(M, N, K, A, K, B, K, D, N, act=act, dact_src=(src, N))
        s = src.double().requires_grad_(True)
        (gsum,) = torch.autograd.grad(fn(s).sum(), s)
        assert _rel(D, z * gsum) < TOL


def test_gemm_dropout_prologue_epilogue_match_numpy_mirror(dev):
    from tecmollm import ops, rng
    M, K, N = 140, 96, 70
    A, B = _rand(M, K, dev=dev, seed=1), _rand(N, K, dev=dev, seed=2)
    p, seedA, seedO = 0.1, 0x1234567890ABCDEF, 987654321
    C = torch.empty(M, N, device=dev)
    ops.gemm(M, N, K, A, K, B, K, C, N, a_drop=ops.drop(p, seedA, 800), out_drop=ops.drop(p, seedO, N))
    ia = (np.arange(M)[:, None] * 800 + np.arange(K)[None, :]).astype(np.uint64)
    io = (np.arange(M)[:, None] * N + np.arange(N)[None, :]).astype(np.uint64)
    ma = torch.from_numpy(rng.keep_mult(seedA, ia, p)).double()
    mo = torch.from_numpy(rng.keep_mult(seedO, io, p)).double()
    ref = ((A.double().cpu() * ma) @ B.double().cpu().t()) * mo
    assert _rel(C, ref) < TOL
    frac = float((ma == 0).double().mean())
    assert 0.07 < frac < 0.13


# ----------------------------------------------------------------------------- GEMM, window views
def _tm(x_scl):
    """(S, C, L) reference conv layout -> time-major (B=S, L, N=1, C)."""
    S, Cc, L = x_scl.shape
    return x_scl.permute(0, 2, 1).contiguous().view(S, L, 1, Cc)


@pytest.mark.parametrize("Bn,L,N,Cin,Cout,k", [(2, 48, 5, 24, 64, 3), (2, 24, 3, 64, 128, 7), (1, 16, 7, 22, 64, 5),
                                               (3, 12, 4, 8, 32, 7)])
def test_gemm_window_conv_forward(dev, Bn, L, N, Cin, Cout, k):
    """A-window == Conv1d over time with zero padding (modules.py:27) on a (B, L, N, C) tensor."""
    from tecmollm import ops
    x = _rand(Bn, L, N, Cin, dev=dev, seed=1)
    w = _rand(Cout, Cin, k, dev=dev, seed=2, scale=0.2)
    b = _rand(Cout, dev=dev, seed=3)
    fp, bp = ops.conv_weight_pack(w)
    y = torch.empty(Bn, L, N, Cout, device=dev)
    ops.gemm(Bn * L * N, Cout, k * Cin, x, Cin, fp, k * Cin, y, Cout,
             a_win=ops.win(N, L, L, 1, k, Cin, (k - 1) // 2), bias=b)
    xs = x.permute(0, 2, 3, 1).reshape(Bn * N, Cin, L).double()
    ref = torch.nn.functional.conv1d(xs, w.double(), b.double(), padding=(k - 1) // 2)
    ref = ref.view(Bn, N, Cout, L).permute(0, 3, 1, 2)
    assert _rel(y, ref) < TOL
    # dX through the flipped pack, dW through the B-window
    dy = _rand(Bn, L, N, Cout, dev=dev, seed=4)
    dx = torch.empty(Bn, L, N, Cin, device=dev)
    ops.gemm(Bn * L * N, Cin, k * Cout, dy, Cout, bp, Cin, dx, Cin, b_layout=ops.B_KN,
             a_win=ops.win(N, L, L, 1, k, Cout, (k - 1) // 2))
    dpack = torch.empty(Cout, k * Cin, device=dev)
    ops.gemm(Cout, k * Cin, Bn * L * N, dy, Cout, x, Cin, dpack, k * Cin, a_layout=ops.A_KM, b_layout=ops.B_KN,
             b_win=ops.win(N, L, L, 1, k, Cin, (k - 1) // 2), split_k=3)
    dw = ops.conv_weight_unpack(dpack, Cout, Cin, k)
    xs.requires_grad_(True)
    wd = w.double().requires_grad_(True)
    out = torch.nn.functional.conv1d(xs, wd, None, padding=(k - 1) // 2)
    gy = dy.permute(0, 2, 3, 1).reshape(Bn * N, Cout, L).double()
    gx, gw = torch.autograd.grad(out, (xs, wd), gy)
    assert _rel(dx, gx.view(Bn, N, Cin, L).permute(0, 3, 1, 2)) < TOL
    assert _rel(dw, gw) < TOL


def test_gemm_window_stride2_and_patch_and_cwin(dev):
    from tecmollm import ops
    Bn, L, N, Cc, Cout = 2, 24, 5, 192, 64
    x = _rand(Bn, L, N, Cc, dev=dev, seed=1)
    w = _rand(Cout, Cc, dev=dev, seed=2, scale=0.1)
    Lo = L // 2
    y = torch.empty(Bn, Lo, N, Cout, device=dev)
    ops.gemm(Bn * Lo * N, Cout, Cc, x, Cc, w, Cc, y, Cout, a_win=ops.win(N, L, Lo, 2, 1, Cc, 0))
    assert _rel(y, x[:, ::2].double() @ w.double().t()) < TOL
    # latent patching 'b (p l) d -> b p (l d)' with l = 4, and its transpose through the C window
    D, pl, P = 128, 4, 3
    c = _rand(Bn, P * pl, N, D, dev=dev, seed=3)
    Wp = _rand(96, pl * D, dev=dev, seed=4, scale=0.05)
    tok = torch.empty(Bn, P, N, 96, device=dev)
    wv = ops.win(N, P * pl, P, pl, pl, D, 0)
    ops.gemm(Bn * P * N, 96, pl * D, c, D, Wp, pl * D, tok, 96, a_win=wv)
    cp = c.view(Bn, P, pl, N, D).permute(0, 1, 3, 2, 4).reshape(Bn, P, N, pl * D).double()
    assert _rel(tok, cp @ Wp.double().t()) < TOL
    dtok = _rand(Bn, P, N, 96, dev=dev, seed=5)
    dc = torch.full((Bn, P * pl, N, D), float("nan"), device=dev)
    ops.gemm(Bn * P * N, pl * D, 96, dtok, 96, Wp, pl * D, dc, D, b_layout=ops.B_KN, c_win=wv)
    ref = (dtok.double() @ Wp.double()).view(Bn, P, N, pl, D).permute(0, 1, 3, 2, 4).reshape(Bn, P * pl, N, D)
    assert _rel(dc, ref) < TOL


def test_gemm_rowbias_wpe(dev):
    from tecmollm import ops
    Bn, P, N, D, K = 2, 3, 7, 64, 32
    A, W = _rand(Bn * P * N, K, dev=dev, seed=1), _rand(D, K, dev=dev, seed=2)
    wpe = _rand(10, D, dev=dev, seed=3)
    C = torch.empty(Bn, P, N, D, device=dev)
    ops.gemm(Bn * P * N, D, K, A, K, W, K, C, D, rowbias=(wpe, D, N, P))
    ref = (A.double() @ W.double().t()).view(Bn, P, N, D) + wpe[:P].double().view(1, P, 1, D)
    assert _rel(C, ref) < TOL


def test_gemm_rejects_bad_arguments(dev):
    from tecmollm import ops
    from tecmollm._lib import TecmError
    A = _rand(8, 8, dev=dev)
    with pytest.raises(TecmError):
        ops.gemm(0, 8, 8, A, 8, A, 8, A, 8)
    with pytest.raises(TecmError):
        ops.gemm(8, 8, 8, A, 8, A, 8, A, 8, a_win=ops.win(2, 2, 2, 1, 3, 5, 1))     # taps*Cw != K


# ----------------------------------------------------------------------------- normalisation
@pytest.mark.parametrize("M,D", [(1000, 768), (37, 256), (5, 1024)])
def test_layernorm_fwd_bwd(dev, M, D):
    from tecmollm import ops
    x = _rand(M, D, dev=dev, seed=1)
    g, b = 1 + 0.1 * _rand(D, dev=dev, seed=2), 0.1 * _rand(D, dev=dev, seed=3)
    y = torch.empty(M, D + 32, device=dev)
    st = torch.empty(M, 2, device=dev)
    ops.layernorm_fwd(x, D, g, b, y, D + 32, st, M, D)
    xd = x.double().requires_grad_(True)
    gd, bd = g.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xd, (D,), gd, bd, 1e-5)
    assert _rel(y[:, :D], ref) < TOL
    dy, dres = _rand(M, D, dev=dev, seed=4), _rand(M, D, dev=dev, seed=5)
    dx = torch.empty(M, D, device=dev)
    dg, db = ops.layernorm_bwd(dy, D, x, D, g, st, dres, dx, M, D)
    gx, gg, gb = torch.autograd.grad(ref, (xd, gd, bd), dy.double())
    assert _rel(dx, gx + dres.double()) < TOL and _rel(dg, gg) < TOL and _rel(db, gb) < TOL
    # second output: dropout(dx) with the device hash (consumed by the GEMM behind the resid dropout)
    from tecmollm import rng
    dx2, dxm = torch.empty(M, D, device=dev), torch.empty(M, D, device=dev)
    ops.layernorm_bwd(dy, D, x, D, g, st, dres, dx2, M, D, dx_masked=dxm, mask_drop=ops.drop(0.1, 99, D))
    mult = torch.from_numpy(rng.keep_mult(99, np.arange(M * D, dtype=np.uint64).reshape(M, D), 0.1)).to(dev)
    assert torch.equal(dx2, dx) and torch.equal(dxm, dx * mult)


@pytest.mark.parametrize("M", [1000, 37])
def test_layernorm_fwd_dropped_bf16_output_is_the_lora_branch_input(dev, M):
    """tecm_layernorm_fwd's third output: bf16(dropout(LN(x))) with the device hash at index row*ld + c -- the cast autocast
    applies to lora_dropout(x) in front of lora_A (modules.py:181, train.py:68) -- next to the fp32 / bf16 outputs."""
    from tecmollm import ops, rng
    D, KE = 768, 800
    x = _rand(M, D, dev=dev, seed=1)
    g, b = 1 + 0.1 * _rand(D, dev=dev, seed=2), 0.1 * _rand(D, dev=dev, seed=3)
    y = torch.empty(M, KE, device=dev)
    y16 = torch.zeros(M, KE, device=dev, dtype=torch.bfloat16)
    y16d = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
    st = torch.empty(M, 2, device=dev)
    ops.layernorm_fwd(x, D, g, b, y, KE, st, M, D, y16=y16, ldy16=KE, y16d=y16d, ldy16d=D, drop16d=ops.drop(0.1, 1234, KE))
    idx = (np.arange(M, dtype=np.uint64)[:, None] * np.uint64(KE) + np.arange(D, dtype=np.uint64)[None, :])
    mult = torch.from_numpy(rng.keep_mult(1234, idx, 0.1)).to(dev)
    assert torch.equal(y16[:, :D], y[:, :D].bfloat16()) and float(y16[:, D:].float().abs().max()) == 0.0
    assert torch.equal(y16d, (y[:, :D] * mult).bfloat16())
    assert 0.05 < float((y16d == 0).float().mean()) < 0.15
    # bf16-only call (the bf16 mode's LN1): no fp32 output at all
    y16b, y16db = torch.empty_like(y16), torch.empty_like(y16d)
    ops.layernorm_fwd(x, D, g, b, None, KE, st, M, D, y16=y16b, ldy16=KE, y16d=y16db, ldy16d=D, drop16d=ops.drop(0.1, 1234, KE))
    assert torch.equal(y16b[:, :D], y16[:, :D]) and torch.equal(y16db, y16d)


@pytest.mark.parametrize("B,T,N,p", [(2, 3, 211, 0.1), (1, 6, 37, 0.1), (3, 3, 5, 0.0)])
def test_layernorm_sequence_major_dropped_output_and_its_gradient(dev, B, T, N, p):
    """ln_f -> F.dropout -> PredictionHead.view(batch, -1) (tec_mollm.py:115, modules.py:307) without a pass of its own:
    tecm_layernorm_fwd writes bf16(dropout(LN(x))) with row (b, t, n) at row (b, n, t), and tecm_layernorm_bwd takes the
    bf16 gradient of that matrix in the same layout and applies the mask (index = the time-major one) itself.  Against the
    plain kernels fed a permuted / pre-masked tensor, bit for bit."""
    from tecmollm import ops, rng
    D = 768
    M = B * T * N
    x = _rand(M, D, dev=dev, seed=1)
    g, b = 1 + 0.1 * _rand(D, dev=dev, seed=2), 0.1 * _rand(D, dev=dev, seed=3)
    st, st2 = torch.empty(M, 2, device=dev), torch.empty(M, 2, device=dev)
    spec = ops.drop(p, 777, D) if p > 0 else ops.NO_DROP
    seq = torch.empty(B, N, T * D, device=dev, dtype=torch.bfloat16)
    ops.layernorm_fwd(x, D, g, b, None, D, st, M, D, y16d=seq, ldy16d=D, drop16d=spec, seq_major=(T, N))
    y = torch.empty(M, D, device=dev)
    ops.layernorm_fwd(x, D, g, b, y, D, st2, M, D)
    idx = np.arange(M * D, dtype=np.uint64).reshape(M, D)
    mult = torch.from_numpy(rng.keep_mult(777, idx, p)).to(dev) if p > 0 else torch.ones(M, D, device=dev)
    want = (y * mult).bfloat16().view(B, T, N, D).permute(0, 2, 1, 3).reshape(B, N, T * D)
    assert torch.equal(seq, want) and torch.equal(st, st2)
    # backward
    dseq = _rand(B, N, T * D, dev=dev, seed=4).bfloat16()
    dres = _rand(M, D, dev=dev, seed=5)
    dx, dxm = torch.empty(M, D, device=dev), torch.empty(M, D, device=dev, dtype=torch.bfloat16)
    dg, db = ops.layernorm_bwd(dseq, D, x, D, g, st, dres, dx, M, D, dx_masked=dxm, mask_drop=ops.drop(0.1, 7, D),
                               dy_seq_major=(T, N, spec if p > 0 else None))
    dy_tm = dseq.view(B, N, T, D).permute(0, 2, 1, 3).reshape(M, D).float() * mult          # what the kernel forms (fp32)
    dx_r, dxm_r = torch.empty_like(dx), torch.empty_like(dxm)
    dg_r, db_r = ops.layernorm_bwd(dy_tm.contiguous(), D, x, D, g, st, dres, dx_r, M, D, dx_masked=dxm_r, mask_drop=ops.drop(0.1, 7, D))
    assert torch.equal(dx, dx_r) and torch.equal(dxm, dxm_r) and torch.equal(dg, dg_r) and torch.equal(db, db_r)
    with pytest.raises(ops._lib.TecmError):               # the remapped dy is a bf16 matrix
        ops.layernorm_bwd(dseq.float(), D, x, D, g, st, dres, dx, M, D, dy_seq_major=(T, N, None))


@pytest.mark.parametrize("M,dt", [(1000, torch.float32), (1000, torch.bfloat16), (37, torch.float32), (4099, torch.bfloat16)])
def test_layernorm_bwd_adds_a_masked_second_gradient_stream(dev, M, dt):
    """tecm_layernorm_bwd with TecmLnAdd: dy += dropmask * dy2 before the LayerNorm backward -- the gradient peft's LoRA
    branch returns for its input (lora_A's d-input GEMM, modules.py:177-186) reaching LN1's output through lora_dropout's
    backward (modules.py:181) -- with dy and dy2 fp32 or bf16, against fp64 from first principles (the mask from the NumPy
    mirror of the device hash), and bit for bit against the plain kernel fed the pre-added gradient."""
    from tecmollm import ops, rng
    D, KE = 768, 800
    x = _rand(M, D, dev=dev, seed=1)
    g = 1 + 0.1 * _rand(D, dev=dev, seed=2)
    st = torch.empty(M, 2, device=dev)
    ops.layernorm_fwd(x, D, g, torch.zeros(D, device=dev), torch.empty(M, D, device=dev), D, st, M, D)
    du = _rand(M, KE, dev=dev, seed=4).to(dt)
    dy2 = _rand(M, D, dev=dev, seed=5, scale=0.3).to(dt)
    dres = _rand(M, D, dev=dev, seed=6)
    spec = ops.drop(0.1, 4321, KE)
    idx = (np.arange(M, dtype=np.uint64)[:, None] * np.uint64(KE) + np.arange(D, dtype=np.uint64)[None, :])
    mult = torch.from_numpy(rng.keep_mult(4321, idx, 0.1)).to(dev)
    dx, dxm = torch.empty(M, D, device=dev), torch.empty(M, D, device=dev)
    dg, db = ops.layernorm_bwd(du, KE, x, D, g, st, dres, dx, M, D, dx_masked=dxm, mask_drop=ops.drop(0.1, 7, D),
                               add=(dy2, D, spec))
    pre = du[:, :D].float() + mult * dy2.float()                       # what the kernel forms in registers (fp32)
    dx_r, dxm_r = torch.empty(M, D, device=dev), torch.empty(M, D, device=dev)
    dg_r, db_r = ops.layernorm_bwd(pre.contiguous(), D, x, D, g, st, dres, dx_r, M, D, dx_masked=dxm_r, mask_drop=ops.drop(0.1, 7, D))
    assert torch.equal(dx, dx_r) and torch.equal(dxm, dxm_r) and torch.equal(dg, dg_r) and torch.equal(db, db_r)
    dy_tot = du[:, :D].double() + mult.double() * dy2.double()
    xd = x.double().requires_grad_(True)
    gd = g.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xd, (D,), gd, None, 1e-5)
    gx, gg = torch.autograd.grad(ref, (xd, gd), dy_tot)
    assert _rel(dx, gx + dres.double()) < TOL and _rel(dg, gg) < TOL and _rel(db, dy_tot.sum(0)) < TOL
    # eval mode: no mask; and a bf16 dy alone (the LN2 / ln_f form of the bf16 mode)
    dx0 = torch.empty(M, D, device=dev)
    ops.layernorm_bwd(du, KE, x, D, g, st, None, dx0, M, D, add=(dy2, D, None))
    gx0, = torch.autograd.grad(torch.nn.functional.layer_norm(xd, (D,), gd, None, 1e-5), (xd,), du[:, :D].double() + dy2.double())
    assert _rel(dx0, gx0) < TOL
    dx1 = torch.empty(M, D, device=dev)
    ops.layernorm_bwd(du, KE, x, D, g, st, None, dx1, M, D)
    gx1, = torch.autograd.grad(torch.nn.functional.layer_norm(xd, (D,), gd, None, 1e-5), (xd,), du[:, :D].double())
    assert _rel(dx1, gx1) < TOL


# register-resident kernels: 4 waves/sequence (2304 quads), 8 waves (4608 quads: L_in = 96), forward-only 2 waves
# (1152 quads, odd sequence count), Cout = 256; (2, 6, 3, 64, 1) and the backward of the 2-wave case take the
# generic multi-pass kernels
@pytest.mark.parametrize("Bn,L,N,Cout,stride", [(2, 48, 5, 64, 2), (1, 24, 9, 128, 2), (2, 6, 3, 64, 1),
                                                (1, 96, 3, 64, 2), (1, 48, 3, 128, 1), (1, 24, 3, 64, 2),
                                                (1, 12, 3, 256, 2)])
def test_groupnorm_gelu_fwd_bwd(dev, Bn, L, N, Cout, stride):
    from tecmollm import ops
    CT = 3 * Cout
    y = _rand(Bn, L, N, CT, dev=dev, seed=1)
    g, b = 1 + 0.1 * _rand(CT, dev=dev, seed=2), 0.1 * _rand(CT, dev=dev, seed=3)
    act = torch.empty_like(y)
    st = torch.empty(Bn * N, 3, 2, device=dev)
    ops.groupnorm_gelu_fwd(y, g, b, act, st, Bn, L, N, Cout)
    yd = y.double().requires_grad_(True)
    gd, bd = g.double().requires_grad_(True), b.double().requires_grad_(True)
    ys = yd.permute(0, 2, 3, 1).reshape(Bn * N, CT, L)
    outs = []
    for j in range(3):
        sl = slice(j * Cout, (j + 1) * Cout)
        outs.append(torch.nn.functional.gelu(torch.nn.functional.group_norm(ys[:, sl], 1, gd[sl], bd[sl], 1e-5)))
    ref = torch.cat(outs, 1).view(Bn, N, CT, L).permute(0, 3, 1, 2)
    assert _rel(act, ref) < TOL
    Lo = (L - 1) // stride + 1
    dact = _rand(Bn, Lo, N, CT, dev=dev, seed=4)
    dy = torch.empty_like(y)
    dg, db, dysum = ops.groupnorm_gelu_bwd(dact, stride, y, g, b, st, dy, Bn, L, N, Cout)
    full = torch.zeros(Bn, L, N, CT, dtype=torch.float64, device=dev)
    full[:, ::stride] = dact.double()
    gy, gg, gb = torch.autograd.grad(ref, (yd, gd, bd), full)
    assert _rel(dy, gy) < TOL and _rel(dg, gg) < TOL and _rel(db, gb) < TOL
    assert _rel(dysum, gy.sum((0, 1, 2))) < TOL


@pytest.mark.parametrize("Bn,L,N,Cout,stride", [(2, 48, 5, 64, 2), (1, 24, 9, 128, 2), (1, 96, 3, 64, 2), (1, 48, 3, 64, 3)])
def test_groupnorm_gelu_compact_strided_act_and_bf16_dact(dev, Bn, L, N, Cout, stride):
    """act_stride = s: only the time steps the stride-s 1x1 conv reads (modules.py:36-41) are written, into a compact
    (B, ceil(L / s), N, CT) tensor, bit-identical to those rows of the full activation; the statistics still cover every
    step.  Backward: a bf16 dact (TECM_GN_DACT_BF16) gives the result of the fp32 kernel fed the same rounded values."""
    from tecmollm import TecmError, ops
    CT = 3 * Cout
    y = _rand(Bn, L, N, CT, dev=dev, seed=1)
    g, b = 1 + 0.1 * _rand(CT, dev=dev, seed=2), 0.1 * _rand(CT, dev=dev, seed=3)
    La = (L + stride - 1) // stride
    for dt in (torch.float32, torch.bfloat16):
        full, st = torch.empty(Bn, L, N, CT, device=dev, dtype=dt), torch.empty(Bn * N, 3, 2, device=dev)
        comp, st2 = torch.full((Bn, La, N, CT), float("nan"), device=dev, dtype=dt), torch.empty(Bn * N, 3, 2, device=dev)
        ops.groupnorm_gelu_fwd(y, g, b, full, st, Bn, L, N, Cout)
        ops.groupnorm_gelu_fwd(y, g, b, comp, st2, Bn, L, N, Cout, act_stride=stride)
        assert torch.equal(comp, full[:, ::stride]) and torch.equal(st, st2)
    with pytest.raises(TecmError):
        ops.groupnorm_gelu_fwd(y, g, b, full, st, Bn, L, N, Cout, act_stride=stride)      # wrong size for a compact act
    dact = _rand(Bn, La, N, CT, dev=dev, seed=4)
    d16 = dact.bfloat16()
    dy_a, dy_b = torch.empty(Bn, L, N, CT, device=dev, dtype=torch.bfloat16), torch.empty(Bn, L, N, CT, device=dev, dtype=torch.bfloat16)
    ra = ops.groupnorm_gelu_bwd(d16, stride, y, g, b, st, dy_a, Bn, L, N, Cout)
    rb = ops.groupnorm_gelu_bwd(d16.float(), stride, y, g, b, st, dy_b, Bn, L, N, Cout)
    assert torch.equal(dy_a, dy_b) and all(torch.equal(p, q) for p, q in zip(ra, rb))
    with pytest.raises(TecmError):
        ops.groupnorm_gelu_bwd(d16, stride, y, g, b, st, torch.empty(Bn, L, N, CT, device=dev), Bn, L, N, Cout)   # bf16 dact, fp32 dy


@pytest.mark.parametrize("Bn,L,N,Cout,stride", [(2, 48, 5, 64, 2), (1, 24, 9, 128, 2), (1, 96, 3, 64, 2), (1, 48, 3, 128, 2),
                                                (2, 16, 7, 64, 1), (1, 12, 3, 256, 2), (1, 40, 3, 64, 2)])
def test_groupnorm_gelu_all_bf16_kernels(dev, Bn, L, N, Cout, stride):
    """TECM_GN_Y_BF16: y, act, dact and dy all bf16 (a lane's unit is 8 channels = one 16-byte access), statistics and
    arithmetic fp32 -- against the fp32-y kernels fed the same rounded y (outputs one bf16 ulp, statistics and parameter
    gradients 1e-5: other lanes sum other elements) and against fp64 GroupNorm(1) + GELU (modules.py:28-29) of that y.
    Sequences that fill the last round of lanes only partly (L = 48 x 24 octs over 256 lanes: 4.5 per lane; L = 40)."""
    from tecmollm import ops
    CT = 3 * Cout
    assert ops.gn_y16_ok(L, N, Cout)
    y16 = _rand(Bn, L, N, CT, dev=dev, seed=1).bfloat16()
    y = y16.float()
    g, b = 1 + 0.1 * _rand(CT, dev=dev, seed=2), 0.1 * _rand(CT, dev=dev, seed=3)
    La = (L + stride - 1) // stride
    a16, st16 = torch.full((Bn, La, N, CT), float("nan"), device=dev, dtype=torch.bfloat16), torch.empty(Bn * N, 3, 2, device=dev)
    st32 = torch.empty(Bn * N, 3, 2, device=dev)
    ops.groupnorm_gelu_fwd(y16, g, b, a16, st16, Bn, L, N, Cout, act_stride=stride)
    quad = ops.gn_reg_ok(L, N, Cout)                               # the fp32-y kernels write bf16 only on their register path
    full = torch.empty(Bn, L, N, CT, device=dev, dtype=torch.bfloat16 if quad else torch.float32)
    ops.groupnorm_gelu_fwd(y, g, b, full, st32, Bn, L, N, Cout)
    a32 = full[:, ::stride].bfloat16()
    assert _rel(st16, st32) < 1e-5
    d = (a16.float() - a32.float()).abs()
    assert bool((d <= a32.float().abs() * 2.0 ** -7 + 1e-6).all())
    yd = y.double().requires_grad_(True)
    gd, bd = g.double().requires_grad_(True), b.double().requires_grad_(True)
    ys = yd.permute(0, 2, 3, 1).reshape(Bn * N, CT, L)
    outs = [torch.nn.functional.gelu(torch.nn.functional.group_norm(ys[:, j * Cout:(j + 1) * Cout], 1, gd[j * Cout:(j + 1) * Cout],
                                                                    bd[j * Cout:(j + 1) * Cout], 1e-5)) for j in range(3)]
    ref = torch.cat(outs, 1).view(Bn, N, CT, L).permute(0, 3, 1, 2)
    assert _rel(a16, ref[:, ::stride]) < 5e-3                     # bf16 output
    dact16 = _rand(Bn, La, N, CT, dev=dev, seed=4).bfloat16()
    dy16 = torch.full((Bn, L, N, CT), float("nan"), device=dev, dtype=torch.bfloat16)
    dyq = torch.empty(Bn, L, N, CT, device=dev, dtype=torch.bfloat16 if quad else torch.float32)
    r16 = ops.groupnorm_gelu_bwd(dact16, stride, y16, g, b, st16, dy16, Bn, L, N, Cout)
    rq = ops.groupnorm_gelu_bwd(dact16 if quad else dact16.float(), stride, y, g, b, st16, dyq, Bn, L, N, Cout)
    dyq = dyq.bfloat16()
    assert torch.isfinite(dy16.float()).all()
    d = (dy16.float() - dyq.float()).abs()
    assert bool((d <= dyq.float().abs() * 2.0 ** -7 + 1e-5 * float(dyq.float().abs().max())).all())
    for a_, b_ in zip(r16, rq):
        assert _rel(a_, b_) < 2e-5
    fullg = torch.zeros(Bn, L, N, CT, dtype=torch.float64, device=dev)
    fullg[:, ::stride] = dact16.double()
    gy, gg, gb = torch.autograd.grad(ref, (yd, gd, bd), fullg)
    assert _rel(dy16, gy) < 5e-3 and _rel(r16[0], gg) < TOL and _rel(r16[1], gb) < TOL
    with pytest.raises(Exception):
        ops.groupnorm_gelu_bwd(dact16.float(), stride, y16, g, b, st16, dy16, Bn, L, N, Cout)      # bf16 y, fp32 dact


def test_colsum_segments_and_dropout(dev):
    from tecmollm import ops, rng
    Bn, P, N, Cn = 3, 4, 50, 70
    x = _rand(Bn * P * N, Cn, dev=dev, seed=1)
    out = ops.colsum(x, Cn, Bn, N, P, Cn)
    assert _rel(out, x.double().view(Bn, P, N, Cn).sum((0, 2))) < TOL
    p, seed = 0.1, 4242
    out = ops.colsum(x, Cn, Bn * P * N, 1, 1, Cn, in_drop=ops.drop(p, seed, Cn), scale=2.0)
    idx = np.arange(Bn * P * N * Cn, dtype=np.uint64).reshape(-1, Cn)
    m = torch.from_numpy(rng.keep_mult(seed, idx, p)).double()
    assert _rel(out[0], 2.0 * (x.double().cpu() * m).sum(0)) < TOL


# ----------------------------------------------------------------------------- attention
def _ref_attention(qkv, Bn, T, N, H, D, keep=None):
    q, k, v = qkv.double().view(Bn, T, N, 3, H, D // H).permute(3, 0, 2, 4, 1, 5)     # (B,N,H,T,hd)
    w = (q @ k.transpose(-1, -2)) / math.sqrt(D // H)
    w = w.masked_fill(~torch.tril(torch.ones(T, T, dtype=torch.bool, device=qkv.device)), float("-inf")).softmax(-1)
    if keep is not None:
        w = w * keep
    return (w @ v).permute(0, 3, 1, 2, 4).reshape(Bn, T, N, D)


@pytest.mark.parametrize("T", [1, 3, 6, 5, 8, 12, 13, 21, 32])
def test_attention_fwd_bwd(dev, T):
    from tecmollm import ops
    Bn, N, H, D = 2, 7, 12, 768
    qkv = _rand(Bn, T, N, 3 * D, dev=dev, seed=1, scale=0.5)
    ctx = torch.empty(Bn, T, N, D, device=dev)
    ops.attention_fwd(qkv, ctx, Bn, T, N, H, D)
    qd = qkv.double().requires_grad_(True)
    ref = _ref_attention(qd, Bn, T, N, H, D)
    assert _rel(ctx, ref) < TOL
    dctx = _rand(Bn, T, N, D, dev=dev, seed=2)
    dqkv = torch.empty_like(qkv)
    ops.attention_bwd(qkv, dctx, dqkv, Bn, T, N, H, D)
    (g,) = torch.autograd.grad(ref, qd, dctx.double())
    assert _rel(dqkv, g) < TOL
    dq16 = torch.empty_like(qkv, dtype=torch.bfloat16)                # bf16 mode's output form: RNE of the same values
    ops.attention_bwd(qkv, dctx, dq16, Bn, T, N, H, D)
    assert torch.equal(dq16, dqkv.bfloat16())
    # a bf16 qkv (TECM_ATT_QKV_BF16: the bf16 mode's c_attn output) is the same arithmetic on the widened values
    q16 = qkv.bfloat16()
    ctx_a, ctx_b = torch.empty_like(ctx), torch.empty_like(ctx)
    ops.attention_fwd(q16, ctx_a, Bn, T, N, H, D, ops.drop(0.1, 5, 1))
    ops.attention_fwd(q16.float(), ctx_b, Bn, T, N, H, D, ops.drop(0.1, 5, 1))
    assert torch.equal(ctx_a, ctx_b)
    dq_a, dq_b = torch.empty_like(dq16), torch.empty_like(dq16)
    ops.attention_bwd(q16, dctx, dq_a, Bn, T, N, H, D, ops.drop(0.1, 5, 1))
    ops.attention_bwd(q16.float(), dctx, dq_b, Bn, T, N, H, D, ops.drop(0.1, 5, 1))
    # (the two instantiations contract multiply-adds differently: last-bit differences in fp32, i.e. the odd flipped bf16
    # rounding (bf16 inputs put many results next to rounding ties: 4 % of the elements with dropout on) -- at most one
    # bf16 ulp (cancelled-out entries: 1e-6 of the largest))
    diff = (dq_a.float() - dq_b.float()).abs()
    assert float((diff > 0).float().mean()) < 0.1
    assert bool((diff <= dq_b.float().abs() * 2.0 ** -7 + 1e-6 * float(dq_b.float().abs().max())).all())


def test_attention_dropout_mask_is_consistent(dev):
    from tecmollm import ops, rng
    Bn, T, N, H, D = 2, 3, 5, 12, 768
    p, seed = 0.1, 777
    qkv = _rand(Bn, T, N, 3 * D, dev=dev, seed=1, scale=0.5)
    ctx = torch.empty(Bn, T, N, D, device=dev)
    ops.attention_fwd(qkv, ctx, Bn, T, N, H, D, ops.drop(p, seed, 1))
    idx = np.arange(Bn * N * H * T * T, dtype=np.uint64)
    keep = torch.from_numpy(rng.keep_mult(seed, idx, p)).double().view(Bn, N, H, T, T).to(dev)
    qd = qkv.double().requires_grad_(True)
    ref = _ref_attention(qd, Bn, T, N, H, D, keep)
    assert _rel(ctx, ref) < TOL
    dctx = _rand(Bn, T, N, D, dev=dev, seed=2)
    dqkv = torch.empty_like(qkv)
    ops.attention_bwd(qkv, dctx, dqkv, Bn, T, N, H, D, ops.drop(p, seed, 1))
    (g,) = torch.autograd.grad(ref, qd, dctx.double())
    assert _rel(dqkv, g) < TOL


@pytest.mark.parametrize("L,Cout", [(48, 64), (24, 128), (96, 64), (336, 64), (168, 128), (100, 64), (12, 256)])
def test_gn_reg_ok_mirrors_the_library(dev, L, Cout):
    """ops.gn_reg_ok (what decides whether a conv block keeps bf16 activations) against the library itself: bf16 act / dy
    are accepted exactly where it says so, and refused loudly elsewhere."""
    from tecmollm import ops
    Bn, N, CT = 1, 3, 3 * Cout
    y = _rand(Bn, L, N, CT, dev=dev, seed=1)
    g, b = 1 + 0.1 * _rand(CT, dev=dev, seed=2), 0.1 * _rand(CT, dev=dev, seed=3)
    act16 = torch.empty(Bn, L, N, CT, device=dev, dtype=torch.bfloat16)
    st = torch.empty(Bn * N, 3, 2, device=dev)
    dact = _rand(Bn, (L + 1) // 2, N, CT, dev=dev, seed=4)
    dy16 = torch.empty_like(act16)
    if ops.gn_reg_ok(L, N, Cout):
        ops.groupnorm_gelu_fwd(y, g, b, act16, st, Bn, L, N, Cout)
        ops.groupnorm_gelu_bwd(dact, 2, y, g, b, st, dy16, Bn, L, N, Cout)
        torch.cuda.synchronize()
        assert torch.isfinite(act16.float()).all() and torch.isfinite(dy16.float()).all()
    else:
        with pytest.raises(Exception):
            ops.groupnorm_gelu_fwd(y, g, b, act16, st, Bn, L, N, Cout)
            ops.groupnorm_gelu_bwd(dact, 2, y, g, b, st, dy16, Bn, L, N, Cout)


# ----------------------------------------------------------------------------- small ops
def test_huber_and_transpose(dev):
    from tecmollm import ops
    pred, tgt = _rand(5000, dev=dev, seed=1, scale=2.0), _rand(5000, dev=dev, seed=2)
    loss, dp = ops.huber_fwd_bwd(pred, tgt, 1.0, 1.0)
    pd = pred.double().requires_grad_(True)
    ref = torch.nn.functional.huber_loss(pd, tgt.double(), delta=1.0)
    (g,) = torch.autograd.grad(ref, pd)
    assert abs(loss.item() - ref.item()) < 1e-5 * abs(ref.item()) + 1e-7 and _rel(dp, g) < TOL
    src = _rand(2304, 32, dev=dev, seed=3)
    dst = torch.zeros(40, 2304, device=dev)
    ops.transpose_scale(src, 32, dst, 2304, 32, 2304, 2.0, dst_off=8 * 2304)
    assert torch.equal(dst[8:], 2.0 * src.t()) and float(dst[:8].abs().max()) == 0.0


def test_huber_strided_on_the_models_permuted_output_view(dev):
    """tecm_huber_fwd_bwd_strided: HuberLoss(delta=1, mean) (train.py:372) of the model's permuted (B, L_out, N, 1) view of
    (B, N, L_out) storage (tec_mollm.py:122-123) against a target in its own layout (train.py:76-78), no contiguous
    copies; the gradient comes back with the PREDICTION's strides and a folded 1/accumulation_steps."""
    from tecmollm import TecmError, ops
    B, N, H = 3, 211, 12
    store = _rand(B, N, H, dev=dev, seed=1, scale=2.0)
    pred = store.permute(0, 2, 1).unsqueeze(-1)                        # (B, H, N, 1), strides (N*H, 1, H, .)
    tgt = _rand(B, 7, 31, H, dev=dev, seed=2)[:, :, :N // 7 + 1].reshape(B, -1, H)[:, :N].permute(0, 2, 1).unsqueeze(-1)
    assert not pred.is_contiguous() and pred.shape == tgt.shape
    loss, dp = ops.huber_fwd_bwd_strided(pred, tgt, 1.0, 0.25)
    pd = pred.double().detach().requires_grad_(True)
    ref = torch.nn.functional.huber_loss(pd, tgt.double(), delta=1.0)
    (g,) = torch.autograd.grad(ref, pd)
    assert abs(loss.item() - ref.item()) < 1e-5 * abs(ref.item()) + 1e-7
    assert dp.shape == pred.shape and dp.stride() == pred.stride() and _rel(dp, 0.25 * g) < TOL
    # contiguous target (bench.py's synthetic y) and the contiguous kernel agree
    tc = tgt.contiguous()
    loss2, dp2 = ops.huber_fwd_bwd_strided(pred, tc, 1.0, 0.25)
    loss3, dp3 = ops.huber_fwd_bwd(pred.contiguous(), tc, 1.0, 0.25)
    assert torch.equal(dp2, dp) and torch.equal(dp2.contiguous(), dp3) and abs(loss2.item() - loss3.item()) < 1e-6
    with pytest.raises(TecmError):
        ops.huber_fwd_bwd_strided(pred.squeeze(-1), tgt.squeeze(-1))
    with pytest.raises(TecmError):
        ops.huber_fwd_bwd_strided(pred.expand(B, H, N, 1)[:, :, ::2], tgt[:, :, ::2])   # not dense: no layout for the gradient


@pytest.mark.parametrize("dt_kn,dt_nk", [(torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16),
                                         (torch.float32, torch.bfloat16), (torch.bfloat16, None), (None, torch.float32)])
def test_lora_fold_writes_only_the_lora_slices(dev, dt_kn, dt_nk):
    """tecm_lora_fold: (alpha/r) * lora_B into rows K.. of [ W ; s B^T ] ([K + r][n]) and columns K.. of [ W^T | s B ]
    ([n][K + r]) (peft Linear on c_attn, modules.py:177-186), fp32 or bf16, everything else untouched."""
    from tecmollm import TecmError, ops
    n, r, K = 2304, 32, 768
    lB = _rand(n, r, dev=dev, seed=5, scale=0.02)
    base_kn = _rand(K + r, n, dev=dev, seed=6)
    base_nk = _rand(n, K + r, dev=dev, seed=7)
    w_kn = base_kn.to(dt_kn).clone() if dt_kn is not None else None
    w_nk = base_nk.to(dt_nk).clone() if dt_nk is not None else None
    ops.lora_fold(lB, 2.0, w_kn, w_nk, K)
    if w_kn is not None:
        assert torch.equal(w_kn[:K], base_kn.to(dt_kn)[:K]) and torch.equal(w_kn[K:], (2.0 * lB).t().to(dt_kn))
    if w_nk is not None:
        assert torch.equal(w_nk[:, :K], base_nk.to(dt_nk)[:, :K]) and torch.equal(w_nk[:, K:], (2.0 * lB).to(dt_nk))
    with pytest.raises(TecmError):
        ops.lora_fold(lB, 2.0, base_kn[:-1], None, K)


def test_pack_vectors_is_torch_cat_in_one_launch(dev):
    from tecmollm import TecmError, ops
    vecs = [_rand(n, dev=dev, seed=i) for i, n in enumerate((64, 64, 64, 128, 5, 1, 300, 64, 64))]
    assert torch.equal(ops.pack_vectors(vecs), torch.cat(vecs))
    assert torch.equal(ops.pack_vectors(vecs[:1]), vecs[0])
    with pytest.raises(TecmError):
        ops.pack_vectors(vecs + vecs)                                  # more than 12
    with pytest.raises(TecmError):
        ops.pack_vectors([vecs[0].double()])


# ----------------------------------------------------------------------------- dropout_apply (tec_mollm.py:115)
def test_dropout_apply_rejects_what_it_cannot_serve(dev):
    from tecmollm import TecmError, ops
    src = _rand(8, 6, dev=dev)
    with pytest.raises(TecmError):
        ops.dropout_apply(src, 8, 6, ops.drop(0.5, 1, 6))          # rows are not 16-byte friendly
    src = _rand(8, 8, dev=dev)
    with pytest.raises(TecmError):
        ops.dropout_apply(src, 8, 8, ops.drop(0.0, 1, 8))          # p = 0: nothing to drop, callers skip the pass


@pytest.mark.parametrize("rows,cols,ld,p", [(1000, 768, 768, 0.1), (37, 100, 800, 0.25), (5, 4, 4, 0.5),
                                            (513, 768, 768, 0.9)])
def test_dropout_apply_matches_numpy_mirror(dev, rows, cols, ld, p):
    """tecm_dropout_apply: dst[r][c] = src[r][c] * keep(seed, r*drop.ld + c) / (1 - p), bit for bit against the NumPy
    mirror of the device hash -- the post-LLM F.dropout and the masked gradient of the embd dropout run through it."""
    from tecmollm import ops, rng
    src = _rand(rows, cols, dev=dev, seed=11)
    seed = ops.splitmix64(20240517)
    spec = ops.drop(p, seed, ld)
    out = ops.dropout_apply(src, rows, cols, spec)
    idx = np.arange(rows, dtype=np.uint64)[:, None] * np.uint64(ld) + np.arange(cols, dtype=np.uint64)[None, :]
    mult = rng.keep_mult(seed, idx, p) if p > 0 else np.ones((rows, cols), np.float32)
    want = src.cpu().numpy() * mult
    assert np.array_equal(out.cpu().numpy(), want)
    if p > 0 and rows * cols > 10000:
        kept = float((out != 0).float().mean())
        assert abs(kept - (1 - p)) < 0.01
    # the same mask again in a second call (pure function of seed and index), and a different one for another seed
    assert torch.equal(ops.dropout_apply(src, rows, cols, spec), out)
    assert torch.equal(ops.dropout_apply(src, rows, cols, spec, out_bf16=True), out.bfloat16())     # the bf16 form: rounded once
    o32, o16 = ops.dropout_apply(src, rows, cols, spec, twin_bf16=True)                           # both from one pass
    assert torch.equal(o32, out) and torch.equal(o16, out.bfloat16())
    if p > 0 and rows * cols > 100:
        assert not torch.equal(ops.dropout_apply(src, rows, cols, ops.drop(p, seed + 1, ld)), out)


@pytest.mark.parametrize("outer,inner,nseg,Cn,p", [(2, 211, 3, 768, 0.1), (1000, 1, 1, 64, 0.0), (37, 5, 1, 576, 0.25)])
def test_colsum_writes_the_masked_input_as_a_bf16_twin(dev, outer, inner, nseg, Cn, p):
    """tecm_colsum_twin: the column sums of tecm_colsum (same bits) and, from the same pass, the (masked) input as a bf16
    matrix -- the gradient a bias / wpe gradient sums in fp32 and two bf16 contractions read rounded."""
    from tecmollm import ops, rng
    rows = outer * nseg * inner
    src = _rand(rows, Cn, dev=dev, seed=5)
    spec = ops.drop(p, 99, Cn) if p > 0 else None
    twin = torch.full((rows, Cn), float("nan"), device=dev, dtype=torch.bfloat16)
    a = ops.colsum(src, Cn, outer, inner, nseg, Cn, in_drop=spec, twin=twin)
    b = ops.colsum(src, Cn, outer, inner, nseg, Cn, in_drop=spec)
    assert torch.equal(a, b)
    mult = torch.from_numpy(rng.keep_mult(99, np.arange(rows * Cn, dtype=np.uint64).reshape(rows, Cn), p)).to(dev) if p > 0 \
        else torch.ones(rows, Cn, device=dev)
    assert torch.equal(twin, (src * mult).bfloat16())
    # wider than 1024 columns (the head's hidden width at L_in = 96): sums + a separate cast, same results
    wide = _rand(64, 1152, dev=dev, seed=6)
    tw = torch.empty(64, 1152, device=dev, dtype=torch.bfloat16)
    assert torch.equal(ops.colsum(wide, 1152, 64, 1, 1, 1152, twin=tw), ops.colsum(wide, 1152, 64, 1, 1, 1152))
    assert torch.equal(tw, wide.bfloat16())
