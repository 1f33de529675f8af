"""GPU parity tests of the bf16 MFMA GEMM (tecm_gemm_bf16): operands rounded to bf16 (RNE), fp32 accumulate.
Reference = fp64 matmul of the bf16-ROUNDED operands, so the only difference left is summation order:
tolerance 2e-4 relative to the reference's max magnitude, the same as for the fp32 kernel."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 2e-4


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _q(t):
    return t.bfloat16().double()


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda")


def _rand(*shape, dev, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev)


@pytest.mark.parametrize("M,N,K", [(300, 200, 96), (513, 768, 800), (1000, 3072, 768), (77, 64, 52), (256, 128, 64)])
def test_bf16_mk_nk_and_kn(dev, M, N, K):
    from tecmollm import ops
    A = _rand(M, K, dev=dev, seed=1)
    Bn, Bk = _rand(N, K, dev=dev, seed=2), _rand(K, N, dev=dev, seed=3)
    bias = _rand(N, dev=dev, seed=4)
    C = torch.full((M, N), float("nan"), device=dev)
    ops.gemm(M, N, K, A, K, Bn, K, C, N, bias=bias, bf16=True)
    assert _rel(C, _q(A) @ _q(Bn).t() + bias.double()) < TOL
    ops.gemm(M, N, K, A, K, Bk, N, C, N, b_layout=ops.B_KN, alpha=0.5, bf16=True)
    assert _rel(C, 0.5 * (_q(A) @ _q(Bk))) < TOL
    # and it really is the bf16 kernel: an fp32-exact result would differ from the rounded-operand reference
    ops.gemm(M, N, K, A, K, Bk, N, C, N, b_layout=ops.B_KN, alpha=0.5)
    assert _rel(C, 0.5 * (_q(A) @ _q(Bk))) > 1e-4


@pytest.mark.parametrize("Mo,No,K,split", [(64, 168, 5000, 7), (768, 512, 3001, 4), (576, 2304, 999, 1),
                                           (128, 64, 130, 1)])
def test_bf16_km_kn_splitk(dev, Mo, No, K, split):
    from tecmollm import ops
    A, B = _rand(K, Mo, dev=dev, seed=1), _rand(K, No, dev=dev, seed=2)
    C = torch.full((Mo, No), float("nan"), device=dev)
    ops.gemm(Mo, No, K, A, Mo, B, No, C, No, a_layout=ops.A_KM, b_layout=ops.B_KN, split_k=split, bf16=True)
    assert _rel(C, _q(A).t() @ _q(B)) < TOL


def test_bf16_small_n_falls_back_to_fp32_kernel(dev):
    """N < 64 (LoRA down-projection, head output): the exact fp32 kernel runs -- mirrored by ops.uses_bf16."""
    from tecmollm import ops
    A, B = _rand(200, 768, dev=dev, seed=1), _rand(32, 768, dev=dev, seed=2)
    C = torch.empty(200, 32, device=dev)
    assert not ops.uses_bf16(32, 768, 768, 768)
    ops.gemm(200, 32, 768, A, 768, B, 768, C, 32, bf16=True)
    assert _rel(C, A.double() @ B.double().t()) < TOL


@pytest.mark.parametrize("Bn,L,N,Cin,Cout,k", [(2, 48, 5, 24, 64, 3), (2, 24, 3, 64, 128, 7)])
def test_bf16_window_conv_fwd_dx_dw(dev, Bn, L, N, Cin, Cout, k):
    from tecmollm import ops
    x = _rand(Bn, L, N, Cin, dev=dev, seed=1)
    w = _rand(Cout, Cin, k, dev=dev, seed=2, scale=0.2)
    fp, bp = ops.conv_weight_pack(w)
    y = torch.empty(Bn, L, N, Cout, device=dev)
    ops.gemm(Bn * L * N, Cout, k * Cin, x, Cin, fp, k * Cin, y, Cout, a_win=ops.win(N, L, L, 1, k, Cin, (k - 1) // 2),
             bf16=True)
    xs = _q(x).permute(0, 2, 3, 1).reshape(Bn * N, Cin, L)
    ref = torch.nn.functional.conv1d(xs, _q(w), None, padding=(k - 1) // 2).view(Bn, N, Cout, L).permute(0, 3, 1, 2)
    assert _rel(y, ref) < TOL
    dy = _rand(Bn, L, N, Cout, dev=dev, seed=4)
    dx = torch.empty(Bn, L, N, Cin, device=dev)
    ops.gemm(Bn * L * N, Cin, k * Cout, dy, Cout, bp, Cin, dx, Cin, b_layout=ops.B_KN,
             a_win=ops.win(N, L, L, 1, k, Cout, (k - 1) // 2), bf16=True)
    dpack = torch.empty(Cout, k * Cin, device=dev)
    ops.gemm(Cout, k * Cin, Bn * L * N, dy, Cout, x, Cin, dpack, k * Cin, a_layout=ops.A_KM, b_layout=ops.B_KN,
             b_win=ops.win(N, L, L, 1, k, Cin, (k - 1) // 2), split_k=3, bf16=True)
    dw = ops.conv_weight_unpack(dpack, Cout, Cin, k)
    gy = _q(dy).permute(0, 2, 3, 1).reshape(Bn * N, Cout, L)
    gx = torch.nn.grad.conv1d_input(xs.shape, _q(w), gy, padding=(k - 1) // 2)
    gw = torch.nn.grad.conv1d_weight(xs, w.shape, gy, padding=(k - 1) // 2)
    if ops.uses_bf16(Cin, k * Cout, Cout, Cin, ops.A_MK, ops.B_KN, Cout, 4):
        assert _rel(dx, gx.view(Bn, N, Cin, L).permute(0, 3, 1, 2)) < TOL
    assert _rel(dw, gw) < TOL


def test_bf16_dropout_epilogue_window_scatter(dev):
    from tecmollm import ops, rng
    M, K, N = 300, 128, 192
    A, B = _rand(M, K, dev=dev, seed=1), _rand(N, K, dev=dev, seed=2)
    p, seedA, seedO = 0.1, 1234567, 7654321
    res = _rand(M, N, dev=dev, seed=3)
    C = torch.empty(M, N, device=dev)
    ops.gemm(M, N, K, A, K, B, K, C, N, a_drop=ops.drop(p, seedA, 800), out_drop=ops.drop(p, seedO, N),
             residual=(res, N), act=ops.ACT_GELU_TANH, bf16=True)
    ia = (np.arange(M)[:, None] * 800 + np.arange(K)[None, :]).astype(np.uint64)
    io = (np.arange(M)[:, None] * N + np.arange(N)[None, :]).astype(np.uint64)
    ma = torch.from_numpy(rng.keep_mult(seedA, ia, p)).to(dev)
    mo = torch.from_numpy(rng.keep_mult(seedO, io, p)).double().to(dev)
    z = _q(A * ma) @ _q(B).t()                       # the mask is applied in fp32 BEFORE the bf16 rounding
    ref = torch.nn.functional.gelu(z, approximate="tanh") * mo + res.double()
    assert _rel(C, ref) < TOL
    # transposed operand with dropout (LoRA dA form): dA = dz^T . drop(u)
    dz, u = _rand(M, 64, dev=dev, seed=5), _rand(M, K, dev=dev, seed=6)
    dA = torch.empty(64, K, device=dev)
    ops.gemm(64, K, M, dz, 64, u, K, dA, K, a_layout=ops.A_KM, b_layout=ops.B_KN, b_drop=ops.drop(p, seedA, 800),
             bf16=True)
    assert _rel(dA, _q(dz).t() @ _q(u * ma)) < TOL


@pytest.mark.parametrize("M,N,K", [(300, 256, 96), (513, 768, 800), (1000, 3072, 768), (77, 64, 72)])
def test_bf16_resident_operands_and_output_are_bit_identical(dev, M, N, K):
    """Operands that already are bf16 in HBM (TecmGemm::io_bf16) and a bf16 output: rounding once at the producer
    instead of in the consumer's loader must not change a single bit of the result."""
    from tecmollm import ops
    A, Bn = _rand(M, K, dev=dev, seed=1), _rand(N, K, dev=dev, seed=2)
    bias = _rand(N, dev=dev, seed=4)
    want = torch.empty(M, N, device=dev)
    pre_w = torch.empty(M, N, device=dev)
    ops.gemm(M, N, K, A, K, Bn, K, want, N, bias=bias, act=ops.ACT_GELU_TANH, preact=(pre_w, N), bf16=True)
    A16, B16 = A.bfloat16(), Bn.bfloat16()
    for a_, b_ in ((A16, B16), (A16, Bn), (A, B16)):
        got = torch.full((M, N), float("nan"), device=dev)
        pre = torch.empty(M, N, device=dev)
        ops.gemm(M, N, K, a_, K, b_, K, got, N, bias=bias, act=ops.ACT_GELU_TANH, preact=(pre, N), bf16=True)
        assert torch.equal(got, want) and torch.equal(pre, pre_w)
    got16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ops.gemm(M, N, K, A16, K, B16, K, got16, N, bias=bias, act=ops.ACT_GELU_TANH, preact=(pre, N), bf16=True)
    assert torch.equal(got16, want.bfloat16()) and torch.equal(pre, pre_w)        # RNE at the store == .bfloat16()
    with pytest.raises(Exception):
        ops.gemm(M, N, K, A16, K, Bn, K, got, N)                                  # fp32 kernel refuses bf16 tensors


@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (257, 800, 2304), (700, 132, 96), (1031, 2304, 800), (4099, 768, 3072),
                                   (513, 3072, 768), (300, 256, 32 * 7)])
@pytest.mark.parametrize("geometry", ["1", "2", "3", "4"], ids=["256x256", "256x128x2blocks", "256x256ring", "256x256antiphase"])
def test_bf16_dma_kernel_matches_the_register_staged_kernel_bit_for_bit(dev, M, N, K, geometry, monkeypatch):
    """The 256x256 LDS-DMA kernel (gemm_bf16_dma.hip: both operands bf16 in HBM) against the 256x128 register-staged
    kernel on the same tensors: ragged M and N tiles, K tails of 32, single K-tile, every epilogue the GPT-2 stack
    uses (bias + GELU + pre-activation store + bf16 C; bias + dropout + residual; GELU' from a saved pre-activation)."""
    from tecmollm import ops
    A16, B16 = _rand(M, K, dev=dev, seed=1).bfloat16(), _rand(N, K, dev=dev, seed=2, scale=0.05).bfloat16()
    bias, res, pre_src = _rand(N, dev=dev, seed=4), _rand(M, N, dev=dev, seed=5), _rand(M, N, dev=dev, seed=6)

    def run():
        outs = []
        c16, pre = torch.empty(M, N, device=dev, dtype=torch.bfloat16), torch.full((M, N), float("nan"), device=dev)
        ops.gemm(M, N, K, A16, K, B16, K, c16, N, bias=bias, act=ops.ACT_GELU_TANH, preact=(pre, N), bf16=True)
        outs += [c16, pre]
        c = torch.full((M, N), float("nan"), device=dev)
        ops.gemm(M, N, K, A16, K, B16, K, c, N, bias=bias, out_drop=ops.drop(0.1, 77, N), residual=(res, N), bf16=True)
        outs.append(c)
        d16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        ops.gemm(M, N, K, A16, K, B16, K, d16, N, act=ops.ACT_GELU_TANH, dact_src=(pre_src, N), bf16=True)
        outs.append(d16)
        # bf16 pre-activation (TECM_IO_PRE_BF16): the forward store and the backward read of the GPT-2 MLP
        e16, pre16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16), torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        ops.gemm(M, N, K, A16, K, B16, K, e16, N, bias=bias, act=ops.ACT_GELU_TANH, preact=(pre16, N), bf16=True)
        outs += [e16, pre16]
        f16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        ops.gemm(M, N, K, A16, K, B16, K, f16, N, act=ops.ACT_GELU_TANH, dact_src=(pre_src.bfloat16(), N), bf16=True)
        outs.append(f16)
        torch.cuda.synchronize()
        return outs

    monkeypatch.setenv("TECM_BF16_DMA", "0")
    want = run()
    monkeypatch.setenv("TECM_BF16_DMA", geometry)
    got = run()
    for g_, w_ in zip(got, want):
        assert torch.equal(g_, w_)
    ref = A16.double() @ B16.double().t()
    assert _rel(want[1], ref + bias.double()) < TOL


@pytest.mark.parametrize("M,N,K", [(300, 256, 96), (1000, 3072, 768), (77, 64, 72)])
def test_bf16_preactivation_is_rounded_before_the_activation(dev, M, N, K):
    """TECM_IO_PRE_BF16: preact = bf16(acc + bias), C = gelu(float(preact)); the backward form multiplies by gelu'
    evaluated at the same bf16 values.  The stored pre-activation equals the fp32 one rounded (RNE) bit for bit."""
    from tecmollm import ops
    A16, B16 = _rand(M, K, dev=dev, seed=1).bfloat16(), _rand(N, K, dev=dev, seed=2, scale=0.2).bfloat16()
    bias = _rand(N, dev=dev, seed=4)
    c32, pre32 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    ops.gemm(M, N, K, A16, K, B16, K, c32, N, bias=bias, act=ops.ACT_GELU_TANH, preact=(pre32, N), bf16=True)
    c, pre16 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ops.gemm(M, N, K, A16, K, B16, K, c, N, bias=bias, act=ops.ACT_GELU_TANH, preact=(pre16, N), bf16=True)
    assert torch.equal(pre16, pre32.bfloat16())
    ref = torch.nn.functional.gelu(pre16.double(), approximate="tanh")
    assert float((c.double() - ref).abs().max()) < 2e-6 * max(1.0, float(ref.abs().max()))
    # backward form: d = (A.B^T) * gelu'(pre16) against autograd's derivative at the rounded point
    d, d32 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    ops.gemm(M, N, K, A16, K, B16, K, d, N, act=ops.ACT_GELU_TANH, dact_src=(pre16, N), bf16=True)
    ops.gemm(M, N, K, A16, K, B16, K, d32, N, act=ops.ACT_GELU_TANH, dact_src=(pre16.float(), N), bf16=True)
    assert torch.equal(d, d32)
    x = pre16.double().requires_grad_(True)
    torch.nn.functional.gelu(x, approximate="tanh").sum().backward()
    want = (A16.double() @ B16.double().t()) * x.grad
    assert _rel(d, want) < TOL
    with pytest.raises(Exception):       # the fp32 kernel has no bf16 pre-activation
        ops.gemm(M, N, K, A16.float(), K, B16.float(), K, c, N, act=ops.ACT_GELU_TANH, preact=(pre16, N))


@pytest.mark.parametrize("geometry", ["5", "6", "7", "8"], ids=["ring", "4w256x128", "4w128x256", "ring288"])
@pytest.mark.parametrize("M,N,K", [(257, 800, 2304), (1031, 2304, 800), (4099, 768, 3072), (513, 3072, 768), (300, 256, 64)])
def test_bf16_dma_16x16x32_geometry_agrees_to_fp32_rounding(dev, M, N, K, geometry, monkeypatch):
    """gemm_bf16_dma5_kernel (v_mfma_f32_16x16x32_bf16 on the four-slot ring; TECM_BF16_DMA=5, diagnostics): another
    summation order inside the MFMA, so not bit-identical -- fp32 outputs agree with the two-slot kernel to 2e-6 of the
    largest value, bf16 outputs to one ulp; ragged tiles, the GELU + pre-activation, residual + dropout and GELU' forms."""
    from tecmollm import ops
    A16, B16 = _rand(M, K, dev=dev, seed=1).bfloat16(), _rand(N, K, dev=dev, seed=2, scale=0.05).bfloat16()
    bias, res, pre_src = _rand(N, dev=dev, seed=4), _rand(M, N, dev=dev, seed=5), _rand(M, N, dev=dev, seed=6)

    def run():
        c16, pre = torch.empty(M, N, device=dev, dtype=torch.bfloat16), torch.full((M, N), float("nan"), device=dev)
        ops.gemm(M, N, K, A16, K, B16, K, c16, N, bias=bias, act=ops.ACT_GELU_TANH, preact=(pre, N), bf16=True)
        c = torch.full((M, N), float("nan"), device=dev)
        ops.gemm(M, N, K, A16, K, B16, K, c, N, bias=bias, out_drop=ops.drop(0.1, 77, N), residual=(res, N), bf16=True)
        d = torch.full((M, N), float("nan"), device=dev)
        ops.gemm(M, N, K, A16, K, B16, K, d, N, act=ops.ACT_GELU_TANH, dact_src=(pre_src, N), bf16=True)
        torch.cuda.synchronize()
        return c16, pre, c, d

    monkeypatch.setenv("TECM_BF16_DMA", "1")
    want = run()
    monkeypatch.setenv("TECM_BF16_DMA", geometry)
    got = run()
    for g_, w_ in zip(got[1:], want[1:]):
        assert torch.isfinite(g_).all() and _rel(g_, w_) < 2e-6
    diff = (got[0].float() - want[0].float()).abs()
    assert bool((diff <= want[0].float().abs() * 2.0 ** -7 + 1e-6).all())
    if geometry == "8":                                  # the 288-row tile sums every element in the 256-row tile's order
        monkeypatch.setenv("TECM_BF16_DMA", "5")
        for g_, w_ in zip(got, run()):
            assert torch.equal(g_, w_)


def test_bf16_dma_ring_kernel_is_race_free_over_repeated_full_size_launches(dev, monkeypatch):
    """The four-slot ring with anti-phase wave groups (gemm_bf16_dma4_kernel) synchronises by counted vmcnt + raw barriers:
    a misplaced wait would show as rare wrong tiles that come and go with load.  Twelve back-to-back launches of the
    step's own long-K shapes (M = 69 864 rows: 273 m-tiles over 256 CUs, every CU busy) must all equal the two-slot
    kernel's result bit for bit."""
    from tecmollm import ops
    M = 69864
    for N, K in ((768, 3072), (2304, 800)):
        A16 = _rand(M, K, dev=dev, seed=11).bfloat16()
        B16 = _rand(N, K, dev=dev, seed=12, scale=0.05).bfloat16()
        bias = _rand(N, dev=dev, seed=13)
        monkeypatch.setenv("TECM_BF16_DMA", "1")
        want = torch.empty(M, N, device=dev)
        ops.gemm(M, N, K, A16, K, B16, K, want, N, bias=bias, bf16=True)
        monkeypatch.setenv("TECM_BF16_DMA", "4")
        outs = [torch.empty(M, N, device=dev) for _ in range(12)]
        for o in outs:
            ops.gemm(M, N, K, A16, K, B16, K, o, N, bias=bias, bf16=True)
        torch.cuda.synchronize()
        for o in outs:
            assert torch.equal(o, want)
        # the dispatched 16x16x32 ring (another summation order inside the MFMA): every launch the same bits, and those
        # within fp32 rounding of the two-slot kernel's
        monkeypatch.delenv("TECM_BF16_DMA")
        rec = ops.enable_gemm_timing()
        for o in outs:
            ops.gemm(M, N, K, A16, K, B16, K, o, N, bias=bias, bf16=True)
        torch.cuda.synchronize()
        assert all("gemm_bf16_p8_kernel" in name for name in ops.summarize_gemm_timing(rec))
        ops.disable_gemm_timing()
        for o in outs[1:]:
            assert torch.equal(o, outs[0])
        assert _rel(outs[0], want) < 2e-6
        # and the ring kernel it replaced (TECM_BF16_P8 = 0)
        monkeypatch.setenv("TECM_BF16_P8", "0")
        rec = ops.enable_gemm_timing()
        for o in outs:
            ops.gemm(M, N, K, A16, K, B16, K, o, N, bias=bias, bf16=True)
        torch.cuda.synchronize()
        assert all("dma5" in name for name in ops.summarize_gemm_timing(rec))
        ops.disable_gemm_timing()
        for o in outs[1:]:
            assert torch.equal(o, outs[0])
        assert _rel(outs[0], want) < 2e-6
        monkeypatch.delenv("TECM_BF16_P8")


@pytest.mark.parametrize("rows", [128, 112, 96])
@pytest.mark.parametrize("M,N,K", [(257, 768, 128), (1031, 2304, 800), (4099, 768, 3072), (513, 3072, 768), (300, 256, 160),
                                   (256, 800, 2304), (70000, 768, 768)])
def test_bf16_p8_geometry_agrees_to_fp32_rounding(dev, M, N, K, rows, monkeypatch):
    """gemm_bf16_p8_kernel (round 5: 256 x 256 x 64 eight-phase K loop, half-tile LDS-DMA ring across raw barriers, wave
    groups half a phase apart; csrc/gemm_bf16_p8_loop.h) against the two-slot 32x32x16 kernel: another summation order,
    so fp32 outputs agree to 2e-6 of the largest value and bf16 outputs to one ulp.  Shapes: ragged M and N tiles, the
    shortest K (two K-tiles: the prologue IS the pipeline), K % 64 == 32 (the half-deep last tile of c_attn + LoRA, also
    with an odd tile count 160 = 2.5 tiles), N = 800 forced through it (TECM_BF16_P8 = 1), and 274 m-tiles (more blocks
    than CUs); epilogues: GELU + bf16 C + pre-activation, residual + dropout, GELU'.  rows: the three tile heights (2 x 128
    and the short tiles 2 x 112 / 2 x 96 whose last accumulator row tiles stay unused), pinned by TECM_P8_ROWS."""
    from tecmollm import ops
    A16, B16 = _rand(M, K, dev=dev, seed=1).bfloat16(), _rand(N, K, dev=dev, seed=2, scale=0.05).bfloat16()
    bias, res, pre_src = _rand(N, dev=dev, seed=4), _rand(M, N, dev=dev, seed=5), _rand(M, N, dev=dev, seed=6)

    def run(expect):
        rec = ops.enable_gemm_timing()
        c16, pre = torch.empty(M, N, device=dev, dtype=torch.bfloat16), torch.full((M, N), float("nan"), device=dev)
        ops.gemm(M, N, K, A16, K, B16, K, c16, N, bias=bias, act=ops.ACT_GELU_TANH, preact=(pre, N), bf16=True)
        c = torch.full((M, N), float("nan"), device=dev)
        ops.gemm(M, N, K, A16, K, B16, K, c, N, bias=bias, out_drop=ops.drop(0.1, 77, N), residual=(res, N), bf16=True)
        d = torch.full((M, N), float("nan"), device=dev)
        ops.gemm(M, N, K, A16, K, B16, K, d, N, act=ops.ACT_GELU_TANH, dact_src=(pre_src, N), bf16=True)
        torch.cuda.synchronize()
        names = set(ops.summarize_gemm_timing(rec))
        ops.disable_gemm_timing()
        assert all(expect in n for n in names), names
        return c16, pre, c, d

    monkeypatch.setenv("TECM_BF16_DMA", "1")
    want = run("gemm_bf16_dma_kernel")
    monkeypatch.delenv("TECM_BF16_DMA")
    monkeypatch.setenv("TECM_BF16_P8", "1")
    monkeypatch.setenv("TECM_P8_ROWS", str(rows))
    got = run(f"gemm_bf16_p8_kernel<{rows}>")
    for g_, w_ in zip(got[1:], want[1:]):
        assert torch.isfinite(g_).all() and _rel(g_, w_) < 2e-6
    diff = (got[0].float() - want[0].float()).abs()
    assert bool((diff <= want[0].float().abs() * 2.0 ** -7 + 1e-6).all())
    again = run(f"gemm_bf16_p8_kernel<{rows}>")          # bit-reproducible
    for g_, w_ in zip(got, again):
        assert torch.equal(g_, w_)
    # against an fp64 product of the same bf16 operands (an independent reference, not another kernel of this library)
    ref = A16.double() @ B16.double().t()
    d2 = torch.empty(M, N, device=dev)
    ops.gemm(M, N, K, A16, K, B16, K, d2, N, bf16=True)
    assert _rel(d2, ref) < 1e-5


@pytest.mark.parametrize("Bn,L,N,Cin,Cout,k", [(2, 48, 5, 24, 64, 3), (2, 24, 7, 64, 128, 7), (1, 48, 300, 24, 64, 5)])
def test_bf16_resident_conv_block_operands_are_bit_identical(dev, Bn, L, N, Cin, Cout, k):
    """bf16 mode keeps the conv-block tensors behind the GroupNorm in HBM as bf16 (ConvBlockFn): y written as bf16 by the
    conv GEMM (column slice of the 3-branch buffer), act / dy read as bf16 through the window (HStagerW) and transposing
    (TStager16) stagers.  Each form must equal the fp32-tensor form of the same call bit for bit: the loader would have
    rounded the same values."""
    from tecmollm import ops
    CT, M, pad = 3 * Cout, Bn * L * N, (k - 1) // 2
    x = _rand(Bn, L, N, Cin, dev=dev, seed=1)
    w = _rand(Cout, Cin, k, dev=dev, seed=2, scale=0.2)
    bias = _rand(Cout, dev=dev, seed=3)
    fp, bp = ops.conv_weight_pack(w)
    # conv forward into column slice 1 of a (.., 3*Cout) buffer: fp32 C vs bf16 C
    y32 = torch.zeros(Bn, L, N, CT, device=dev)
    y16 = torch.zeros(Bn, L, N, CT, device=dev, dtype=torch.bfloat16)
    for y in (y32, y16):
        ops.gemm(M, Cout, k * Cin, x, Cin, fp, k * Cin, y, CT, c_off=Cout, a_win=ops.win(N, L, L, 1, k, Cin, pad), bias=bias,
                 bf16=True)
    assert torch.equal(y16, y32.bfloat16())
    # dX: A = dy through the window (+ column offset), accumulate into dx
    dy32 = _rand(Bn, L, N, CT, dev=dev, seed=4).bfloat16().float()       # values that are exactly representable
    dy16 = dy32.bfloat16()
    outs = []
    for dy in (dy32, dy16):
        dx = torch.full((Bn, L, N, Cin), 0.5, device=dev)
        ops.gemm(M, Cin, k * Cout, dy, CT, bp, Cin, dx, Cin, b_layout=ops.B_KN, a_off=Cout,
                 a_win=ops.win(N, L, L, 1, k, Cout, pad), accumulate=True, bf16=True)
        outs.append(dx)
    if ops.uses_bf16(Cin, k * Cout, CT, Cin, ops.A_MK, ops.B_KN, Cout, 4):          # else the fp32 call ran the exact kernel
        assert torch.equal(outs[0], outs[1])
    gy = dy32[..., Cout:2 * Cout].double().permute(0, 2, 3, 1).reshape(Bn * N, Cout, L)
    xs = _q(x).permute(0, 2, 3, 1).reshape(Bn * N, Cin, L)
    gx = torch.nn.grad.conv1d_input(xs.shape, _q(w), gy, padding=pad).view(Bn, N, Cin, L).permute(0, 3, 1, 2)
    assert _rel(outs[1], gx + 0.5) < TOL
    # dW: A = dy [k][m] (+ column offset), B = x through the window, split-K
    outs = []
    for dy in (dy32, dy16):
        dpack = torch.empty(Cout, k * Cin, device=dev)
        ops.gemm(Cout, k * Cin, M, dy, CT, x, Cin, dpack, k * Cin, a_layout=ops.A_KM, b_layout=ops.B_KN, a_off=Cout,
                 b_win=ops.win(N, L, L, 1, k, Cin, pad), split_k=3, bf16=True)
        outs.append(dpack)
    assert torch.equal(outs[0], outs[1])
    gw = torch.nn.grad.conv1d_weight(xs, w.shape, gy, padding=pad)
    assert _rel(ops.conv_weight_unpack(outs[1], Cout, Cin, k), gw) < TOL
    # strided 1x1 conv: A = act through the stride-2 window; its dW: B = act [k][n] through the same window
    Lo = (L - 1) // 2 + 1
    a32 = _rand(Bn, L, N, CT, dev=dev, seed=5).bfloat16().float()
    a16 = a32.bfloat16()
    wf = _rand(Cout, CT, dev=dev, seed=6, scale=0.1)
    dout = _rand(Bn, Lo, N, Cout, dev=dev, seed=7)
    fo, wo = [], []
    for a in (a32, a16):
        out = torch.empty(Bn, Lo, N, Cout, device=dev)
        ops.gemm(Bn * Lo * N, Cout, CT, a, CT, wf, CT, out, Cout, a_win=ops.win(N, L, Lo, 2, 1, CT, 0), bias=bias, bf16=True)
        fo.append(out)
        dwf = torch.empty(Cout, CT, device=dev)
        ops.gemm(Cout, CT, Bn * Lo * N, dout, Cout, a, CT, dwf, CT, a_layout=ops.A_KM, b_layout=ops.B_KN,
                 b_win=ops.win(N, L, Lo, 2, 1, CT, 0), split_k=2, bf16=True)
        wo.append(dwf)
    assert torch.equal(fo[0], fo[1]) and torch.equal(wo[0], wo[1])
    ref = a32[:, ::2].double() @ _q(wf).t() + bias.double()
    assert _rel(fo[1], ref) < TOL
    refw = _q(dout).reshape(-1, Cout).t() @ a32[:, ::2].double().reshape(-1, CT)
    assert _rel(wo[1], refw) < TOL


@pytest.mark.parametrize("Bn,L,N,Cout,stride", [(2, 48, 5, 64, 2), (1, 24, 9, 128, 2), (1, 96, 3, 64, 2)])
def test_groupnorm_gelu_bf16_outputs(dev, Bn, L, N, Cout, stride):
    """act / dy written as bf16 == RNE of the fp32 outputs; statistics and parameter gradients unchanged."""
    from tecmollm import ops
    CT = 3 * Cout
    y = _rand(Bn, L, N, CT, dev=dev, seed=1)
    g, b = 1 + 0.1 * _rand(CT, dev=dev, seed=2), 0.1 * _rand(CT, dev=dev, seed=3)
    act32, st32 = torch.empty_like(y), torch.empty(Bn * N, 3, 2, device=dev)
    act16, st16 = torch.empty_like(y, dtype=torch.bfloat16), torch.empty(Bn * N, 3, 2, device=dev)
    ops.groupnorm_gelu_fwd(y, g, b, act32, st32, Bn, L, N, Cout)
    ops.groupnorm_gelu_fwd(y, g, b, act16, st16, Bn, L, N, Cout)
    assert torch.equal(st16, st32) and torch.equal(act16, act32.bfloat16())
    Lo = (L - 1) // stride + 1
    dact = _rand(Bn, Lo, N, CT, dev=dev, seed=4)
    dy32, dy16 = torch.empty_like(y), torch.empty_like(y, dtype=torch.bfloat16)
    r32 = ops.groupnorm_gelu_bwd(dact, stride, y, g, b, st32, dy32, Bn, L, N, Cout)
    r16 = ops.groupnorm_gelu_bwd(dact, stride, y, g, b, st16, dy16, Bn, L, N, Cout)
    assert torch.equal(dy16, dy32.bfloat16())
    for a_, b_ in zip(r16, r32):                          # block partials are combined with LDS atomics: order varies
        assert _rel(a_, b_) < 1e-5
    # (a bf16 y used to be refused; round 4 serves it with its own kernels: tests/test_gpu_ops.py,
    #  test_groupnorm_gelu_all_bf16_kernels)
    with pytest.raises(Exception):
        ops.groupnorm_gelu_fwd(y.bfloat16(), g, b, act32, st16, Bn, L, N, Cout)     # a bf16 y with an fp32 activation


@pytest.mark.parametrize("f32", [False, True], ids=["bf16", "fp32"])
@pytest.mark.parametrize("Bn,Lc,N,cin,ld_in,Cout", [(2, 48, 5, 22, 24, 64), (1, 24, 9, 64, 64, 128), (1, 96, 3, 22, 24, 64),
                                                  (2, 8, 4, 22, 24, 64), (1, 48, 6, 64, 64, 128), (1, 16, 2911, 22, 24, 64)])
def test_conv_dx_sequence_tile_kernel(dev, Bn, Lc, N, cin, ld_in, Cout, f32):
    """csrc/conv_seq.hip: the input gradient of the three parallel Conv1d (modules.py:43-60) in one launch that reads dy
    once -- bf16 dy (bf16 mode: operands rounded to bf16, fp32 accumulate) and fp32 dy (exact f32 MFMA) -- against an
    fp64 conv-backward of the same operands (2e-4 / 2e-5) and against the three accumulating window GEMMs it replaces
    (same operands, another summation order: 1e-5).  Ragged node blocks (N % 4 != 0), time chunks (Lc = 96 with halos),
    both channel widths, the padding columns (cin .. ld_in) exactly zero."""
    from tecmollm import ops
    g = torch.Generator().manual_seed(Lc * 1000 + N)
    CT = 3 * Cout
    dy = torch.randn(Bn, Lc, N, CT, generator=g) * 0.5
    dy = dy if f32 else dy.bfloat16()
    ws = [torch.randn(Cout, cin, k, generator=g) / (cin * k) ** 0.5 for k in (3, 5, 7)]
    # fp64 reference on the operands as the kernel sees them, sequence-major
    S = Bn * N
    dys = dy.double().permute(0, 2, 3, 1).reshape(S, CT, Lc)
    ref = torch.zeros(S, cin, Lc, dtype=torch.float64)
    for j, (k, w) in enumerate(zip((3, 5, 7), ws)):
        wk = w.double() if f32 else w.bfloat16().double()
        ref += torch.nn.grad.conv1d_input((S, cin, Lc), wk, dys[:, j * Cout:(j + 1) * Cout].contiguous(), padding=(k - 1) // 2)
    ref = ref.reshape(Bn, N, cin, Lc).permute(0, 3, 1, 2)                      # (B, Lc, N, cin)
    dyd = dy.to(dev)
    wd = [w.to(dev) for w in ws]
    out = torch.full((Bn, Lc, N, ld_in), float("nan"), device=dev)
    assert ops.conv_dx_seq_ok(Lc, Cout, ld_in)
    ops.conv_dx(dyd, wd[0], wd[1], wd[2], out, Bn, Lc, N, Cout, cin, ld_in)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    assert _rel(out[..., :cin], ref) < (2e-5 if f32 else TOL)
    if ld_in > cin:
        assert float(out[..., cin:].abs().max()) == 0.0
    # the window-GEMM path on the same tensors
    M = Bn * Lc * N
    old = torch.empty(Bn, Lc, N, ld_in, device=dev)
    for j, (k, w) in enumerate(zip((3, 5, 7), wd)):
        wp = w if ld_in == cin else torch.nn.functional.pad(w, (0, 0, 0, ld_in - cin))
        _, bp = ops.conv_weight_pack(wp.contiguous(), want_bwd=True)
        ops.gemm(M, ld_in, k * Cout, dyd, CT, bp, ld_in, old, ld_in, b_layout=ops.B_KN, a_off=j * Cout,
                 a_win=ops.win(N, Lc, Lc, 1, k, Cout, (k - 1) // 2), accumulate=(j > 0), bf16=not f32)
    assert _rel(out, old) < 1e-5


@pytest.mark.parametrize("Bn,Lc,N,cin,ld_in,Cout", [(2, 48, 5, 22, 24, 64), (1, 24, 9, 64, 64, 128), (2, 8, 4, 22, 24, 64),
                                                  (1, 12, 7, 64, 64, 128), (1, 16, 2911, 22, 24, 64), (3, 4, 3, 64, 64, 64),
                                                  (1, 24, 1203, 64, 64, 128), (2, 40, 6, 64, 64, 128), (1, 96, 3, 22, 24, 128)])
@pytest.mark.parametrize("f32", [False, True], ids=["bf16", "fp32"])
def test_conv_dw_sequence_tile_kernel(dev, Bn, Lc, N, cin, ld_in, Cout, f32, monkeypatch):
    """csrc/conv_dw_seq.hip: the weight gradients of the three parallel Conv1d (modules.py:43-60) from the block input
    and dy -- both bf16 (bf16 mode) or both fp32 (exact f32 MFMA) -- in one persistent launch (+ the fixed-order slab
    reduction), against an fp64 conv-backward of the same operands (2e-4 / 2e-5).  Ragged node blocks, sequences that are not a multiple of 8 or longer than one tile
    (time chunks of 24 with re-read halos, a ragged last chunk), every channel-width pair, more tiles than blocks
    (persistent loop, two register sets in flight) and fewer; the padding columns of the input carry garbage and must
    not matter; two runs agree bit for bit."""
    from tecmollm import ops
    g = torch.Generator().manual_seed(Lc * 1000 + N)
    CT = 3 * Cout
    dt = torch.float32 if f32 else torch.bfloat16
    dy = (torch.randn(Bn, Lc, N, CT, generator=g) * 0.5).to(dt)
    x = torch.randn(Bn, Lc, N, ld_in, generator=g).to(dt)
    tol = 2e-5 if f32 else TOL
    S = Bn * N
    xs = x[..., :cin].double().permute(0, 2, 3, 1).reshape(S, cin, Lc)
    dys = dy.double().permute(0, 2, 3, 1).reshape(S, CT, Lc)
    refs = [torch.nn.grad.conv1d_weight(xs, (Cout, cin, k), dys[:, j * Cout:(j + 1) * Cout].contiguous(), padding=(k - 1) // 2)
            for j, k in enumerate((3, 5, 7))]
    assert ops.conv_dw_seq_ok(Lc, Cout, ld_in)
    xd, dyd = x.to(dev), dy.to(dev)
    got = ops.conv_dw(xd, dyd, Bn, Lc, N, Cout, cin, ld_in)
    torch.cuda.synchronize()
    for w, r in zip(got, refs):
        assert w.shape == r.shape and torch.isfinite(w).all()
        assert _rel(w, r) < tol
    again = ops.conv_dw(xd, dyd, Bn, Lc, N, Cout, cin, ld_in)
    for w, v in zip(got, again):
        assert torch.equal(w, v)
    # a handful of blocks: every block walks several tiles (the register prefetch of the next tile is exercised)
    monkeypatch.setenv("TECM_CONV_DW_BLOCKS", "6")
    ops._cu_count.clear()
    few = ops.conv_dw(xd, dyd, Bn, Lc, N, Cout, cin, ld_in)
    ops._cu_count.clear()
    for w, r in zip(few, refs):
        assert _rel(w, r) < tol


@pytest.mark.parametrize("f32", [False, True], ids=["bf16", "fp32"])
@pytest.mark.parametrize("Bn,Lc,N,cin,ld_in,Cout", [(2, 48, 5, 22, 24, 64), (1, 24, 9, 64, 64, 128), (1, 96, 3, 22, 24, 64),
                                                  (2, 8, 4, 22, 24, 64), (1, 16, 2911, 22, 24, 64), (1, 48, 7, 64, 64, 128)])
def test_conv_fwd_sequence_tile_kernel(dev, Bn, Lc, N, cin, ld_in, Cout, f32):
    """csrc/conv_seq.hip: the three parallel Conv1d (k = 3, 5, 7, modules.py:43-60) of a bf16 input in one launch (bias
    included, whole rows of y written once) against an fp64 convolution of the same bf16-rounded operands (2e-4) and
    against the three window GEMMs it replaces (same operands, same k order: 1e-5).  Ragged node blocks, time chunks
    (Lc = 96), both channel widths; the padding columns of the input (cin .. ld_in) carry garbage and must not matter."""
    from tecmollm import ops
    g = torch.Generator().manual_seed(Lc * 1000 + N + 1)
    CT = 3 * Cout
    x = torch.randn(Bn, Lc, N, ld_in, generator=g)
    x16 = x if f32 else x.bfloat16()
    ws = [torch.randn(Cout, cin, k, generator=g) / (cin * k) ** 0.5 for k in (3, 5, 7)]
    bs = [torch.randn(Cout, generator=g) * 0.1 for _ in range(3)]
    S = Bn * N
    xs = x16[..., :cin].double().permute(0, 2, 3, 1).reshape(S, cin, Lc)
    ref = torch.cat([torch.nn.functional.conv1d(xs, (w if f32 else w.bfloat16()).double(), b.double(), padding=(k - 1) // 2)
                     for k, w, b in zip((3, 5, 7), ws, bs)], dim=1)                       # (S, CT, Lc)
    ref = ref.reshape(Bn, N, CT, Lc).permute(0, 3, 1, 2)                                # (B, Lc, N, CT)
    xd = x16.to(dev)
    wd = [w.to(dev) for w in ws]
    bias = torch.cat(bs).to(dev)
    y = torch.full((Bn, Lc, N, CT), float("nan"), device=dev)
    assert ops.conv_fwd_seq_ok(Lc, Cout, ld_in)
    ops.conv_fwd(xd, wd[0], wd[1], wd[2], bias, y, Bn, Lc, N, Cout, cin, ld_in)
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    assert _rel(y, ref) < (2e-5 if f32 else TOL)
    M = Bn * Lc * N
    old = torch.empty(Bn, Lc, N, CT, device=dev)
    for j, (k, w) in enumerate(zip((3, 5, 7), wd)):
        wp = w if ld_in == cin else torch.nn.functional.pad(w, (0, 0, 0, ld_in - cin))
        fp, _ = ops.conv_weight_pack(wp.contiguous(), want_bwd=False)
        ops.gemm(M, Cout, k * ld_in, xd, ld_in, fp, k * ld_in, old, CT, c_off=j * Cout,
                 a_win=ops.win(N, Lc, Lc, 1, k, ld_in, (k - 1) // 2), bias=bias[j * Cout:(j + 1) * Cout].contiguous(), bf16=not f32)
    assert _rel(y, old) < 1e-5
    if not f32:
        # y as the bf16 tensor a bf16 Conv1d returns under autocast (TecmConvFwd.y_bf16): the same accumulators, rounded once
        y16 = torch.full((Bn, Lc, N, CT), float("nan"), device=dev, dtype=torch.bfloat16)
        ops.conv_fwd(xd, wd[0], wd[1], wd[2], bias, y16, Bn, Lc, N, Cout, cin, ld_in)
        assert torch.equal(y16, y.bfloat16())


# ------------------------------------------------------------------ natural-orientation weight-gradient kernel (round 4)
_TN_SHAPES = [  # Mo, No, K rows (B, Lout, N), window of the B operand (Lin, stride, taps, Cw) or None, alpha
    (576, 2304, (3, 1, 1471), (3, 3, 3, 768), 1.0),        # head W1: the view(S, P * 768) of the time-major hidden state
    (768, 512, (2, 3, 1013), (12, 4, 4, 128), 1.0),        # patch projection: 'b (p l) d -> b p (l d)'
    (64, 192, (2, 24, 211), None, 1.0),                    # first 1x1 conv (compact activations)
    (128, 384, (1, 12, 701), None, 1.0),                   # second 1x1 conv
    (2304, 32, (1, 3, 1900), None, 2.0),                   # lora_B (the 32 z columns of the 800-wide row buffer)
    (32, 768, (1, 3, 1900), None, 1.0),                    # lora_A (dz = columns 768.. of the 800-wide gradient buffer)
    (576, 2304, (1, 1, 4096), None, 1.0),                  # K a multiple of every stage / split size
]


@pytest.mark.parametrize("Mo,No,rows,bwin,alpha", _TN_SHAPES, ids=[f"{s[0]}x{s[1]}" + ("w" if s[3] else "") + f"_{i}"
                                                                   for i, s in enumerate(_TN_SHAPES)])
def test_bf16_weight_gradient_natural_orientation_kernel(dev, Mo, No, rows, bwin, alpha, monkeypatch):
    """csrc/gemm_bf16_tn.hip against fp64 on the bf16 values and against the register-transposing kernel it replaces
    (TECM_BF16_TN=0): same products, another summation order.  K has a ragged tail (rows % 32 != 0) in most cases, split
    counts as the library picks them AND a count that leaves the last split short."""
    from tecmollm import ops, _lib
    Bq, Lout, N = rows
    K = Bq * Lout * N
    lda = Mo if Mo != 32 else 800                           # lora_A reads dz out of the [d LN1-out | dz] buffer
    A = _rand(K, lda, dev=dev, seed=11).bfloat16()
    a_off = lda - Mo
    if bwin:
        Lin, stride, taps, Cw = bwin
        src = _rand(Bq, Lin, N, Cw, dev=dev, seed=12).bfloat16()
        w = ops.win(N, Lin, Lout, stride, taps, Cw, 0)
        # logical B[(b, to, n)][tap * Cw + c] = src[b, to * stride + tap, n, c]
        idx = (torch.arange(Lout, device=dev)[:, None] * stride + torch.arange(taps, device=dev)[None, :])   # (Lout, taps)
        Bl = src[:, idx]                                     # (Bq, Lout, taps, N, Cw)
        Bl = Bl.permute(0, 1, 3, 2, 4).reshape(K, taps * Cw)
        Bt, ldb, b_off = src, Cw, 0
    else:
        ldb = No if No != 32 else 800
        Bt = _rand(K, ldb, dev=dev, seed=12).bfloat16()
        b_off = ldb - No
        Bl = Bt[:, b_off:]
        w = None
    ref = alpha * (A[:, a_off:].double().t() @ Bl.double())
    want = _lib.lib().tecm_gemm_tn_splits(Mo, No, K)
    assert want >= 2 and ops.pick_split_k(Mo, No, K, prec=ops.PREC_BF16) == want
    outs = {}
    for tn, split in (("1", want), ("1", 3), ("0", want)):
        monkeypatch.setenv("TECM_BF16_TN", tn)
        C = torch.full((Mo, No), float("nan"), device=dev)
        rec = ops.enable_gemm_timing(detail=True)
        ops.gemm(Mo, No, K, A, lda, Bt, ldb, C, No, a_layout=ops.A_KM, b_layout=ops.B_KN, a_off=a_off, b_off=b_off, b_win=w,
                 alpha=alpha, split_k=split, bf16=True)
        agg = ops.summarize_gemm_timing(rec)
        ops.disable_gemm_timing()
        assert len(agg) == 1 and all(("gemm_bf16_tn_kernel" in k) == (tn == "1") for k in agg), agg.keys()
        assert _rel(C, ref) < TOL, (tn, split)
        outs[(tn, split)] = C
    assert _rel(outs[("1", want)], outs[("0", want)]) < 1e-5
    # bit-reproducible: fixed summation order inside a block and across the split-K slabs, no race in the LDS ring
    monkeypatch.setenv("TECM_BF16_TN", "1")
    for _ in range(3):
        C2 = torch.full((Mo, No), float("nan"), device=dev)
        ops.gemm(Mo, No, K, A, lda, Bt, ldb, C2, No, a_layout=ops.A_KM, b_layout=ops.B_KN, a_off=a_off, b_off=b_off, b_win=w,
                 alpha=alpha, split_k=want, bf16=True)
        assert torch.equal(C2, outs[("1", want)])


@pytest.mark.parametrize("p8", ["0", ""])
@pytest.mark.parametrize("Bq,Lin,N,taps,Cw,Nout", [(2, 12, 211, 4, 128, 768), (1, 8, 300, 2, 64, 256), (3, 12, 2911, 4, 128, 768)])
def test_bf16_dma_kernel_walks_a_window_view_of_a(dev, Bq, Lin, N, taps, Cw, Nout, p8, monkeypatch):
    """The patch projection 'b (p l) d -> b p (l d)' + Linear (modules.py:114-116) with bf16 operands: the LDS-DMA kernels
    move their A source pointers one time step on at every tap boundary -- the eight-phase geometry (round 5) and, with
    TECM_BF16_P8 = 0, the first geometry.  Against fp64 on the bf16 values and the register-staged kernel fed an fp32 copy
    of the same (already rounded) weight; epilogue: bias + row bias (wpe) + dropout, the straight-line form 5 of gemm_impl.h.
    The third case has 103 m-tiles of 256 rows: tiles that straddle the (b, p) boundaries of the view."""
    from tecmollm import ops
    if p8:
        monkeypatch.setenv("TECM_BF16_P8", p8)
    Lout = Lin // taps
    M, K = Bq * Lout * N, taps * Cw
    src = _rand(Bq, Lin, N, Cw, dev=dev, seed=21).bfloat16()
    W = _rand(Nout, K, dev=dev, seed=22, scale=0.05)
    W16 = W.bfloat16()
    bias = _rand(Nout, dev=dev, seed=23)
    wpe = _rand(Lout, Nout, dev=dev, seed=24)
    w = ops.win(N, Lin, Lout, taps, taps, Cw, 0)
    spec = ops.drop(0.1, 4242, Nout)
    outs = []
    for Wop in (W16, W16.float()):
        C = torch.full((M, Nout), float("nan"), device=dev)
        rec = ops.enable_gemm_timing(detail=True)
        ops.gemm(M, Nout, K, src, Cw, Wop, K, C, Nout, a_win=w, bias=bias, rowbias=(wpe, Nout, N, Lout), out_drop=spec, bf16=True)
        names = list(ops.summarize_gemm_timing(rec))
        ops.disable_gemm_timing()
        want = ("gemm_bf16_dma_kernel" if p8 == "0" else "gemm_bf16_p8_kernel") if Wop.dtype == torch.bfloat16 else "gemm_bf16_kernel"
        assert want in names[0], names
        outs.append(C)
    A = src.view(Bq, Lout, taps, N, Cw).permute(0, 1, 3, 2, 4).reshape(M, K).double()
    ref = A @ W16.double().t() + bias.double() + wpe.double().repeat_interleave(N, 0).repeat(Bq, 1)
    from tecmollm import rng
    mult = torch.from_numpy(rng.keep_mult(4242, np.arange(M * Nout, dtype=np.uint64).reshape(M, Nout), 0.1)).to(dev).double()
    assert _rel(outs[0], ref * mult) < TOL and _rel(outs[1], ref * mult) < TOL
    assert _rel(outs[0], outs[1]) < 1e-6


@pytest.mark.parametrize("p8", ["0", ""])
@pytest.mark.parametrize("Bq,Lin,N,taps,Cw,Kc", [(2, 12, 211, 4, 128, 768), (1, 8, 300, 2, 64, 256), (3, 12, 2911, 4, 128, 768)])
def test_bf16_window_scatter_of_the_result_takes_the_straight_line_epilogue(dev, Bq, Lin, N, taps, Cw, Kc, p8, monkeypatch):
    """The patch projection's input gradient (autograd of modules.py:114-116): d conv[b, p*l + tap, n, :] = (d h . W)[(b, p, n),
    tap*Cw ..] -- a plain bf16 contraction whose fp32 result is scattered through a window view of C (form 6 of
    gemm_impl.h's straight-line epilogue; rounds 1-4 ran it through the rolled generic loop).  Against fp64 on the bf16
    values, on the eight-phase geometry and (TECM_BF16_P8 = 0) the older ones; every element of the target is written once."""
    from tecmollm import ops
    if p8:
        monkeypatch.setenv("TECM_BF16_P8", p8)
    Lout = Lin // taps
    M, Ncols = Bq * Lout * N, taps * Cw
    dh = _rand(M, Kc, dev=dev, seed=31).bfloat16()
    Wt = _rand(Ncols, Kc, dev=dev, seed=32, scale=0.05).bfloat16()            # W^T as [row = (tap, c)][k]
    bias = _rand(Ncols, dev=dev, seed=33)
    out = torch.full((Bq, Lin, N, Cw), float("nan"), device=dev)
    w = ops.win(N, Lin, Lout, taps, taps, Cw, 0)
    rec = ops.enable_gemm_timing(detail=True)
    ops.gemm(M, Ncols, Kc, dh, Kc, Wt, Kc, out, Cw, c_win=w, bias=bias, bf16=True)
    names = list(ops.summarize_gemm_timing(rec))
    ops.disable_gemm_timing()
    takes = p8 == "" and not (1 <= Ncols % 256 <= 128 and Ncols < 768)      # tecm_gemm16_p8_try declines mostly empty n-tiles
    assert ("gemm_bf16_p8_kernel" in names[0]) == takes, names
    ref = dh.double() @ Wt.double().t() + bias.double()                       # (M, taps*Cw), row (b, p, n)
    ref = ref.view(Bq, Lout, N, taps, Cw).permute(0, 1, 3, 2, 4).reshape(Bq, Lin, N, Cw)
    assert torch.isfinite(out).all() and _rel(out, ref) < TOL


@pytest.mark.parametrize("Bn,Lc,N,cin,ld_in,Cout,stride", [(2, 48, 7, 22, 24, 64, 2), (1, 24, 9, 64, 64, 128, 2), (1, 8, 5, 22, 24, 64, 1)])
def test_conv_fwd_hands_the_groupnorm_statistics_over(dev, Bn, Lc, N, cin, ld_in, Cout, stride):
    """tecm_conv_fwd_bf16 with TecmConvFwd::stats: (mean, rstd) of GroupNorm(1, Cout) per sequence and branch from the rounded
    y it holds in registers, then the elementwise norm + GELU (TECM_GN_STATS_GIVEN) -- against fp64 statistics of the y it
    wrote and against the sequence-resident kernel that computes its own."""
    from tecmollm import ops
    CT = 3 * Cout
    inp = torch.zeros(Bn, Lc, N, ld_in, device=dev)
    inp[..., :cin] = _rand(Bn, Lc, N, cin, dev=dev, seed=31)
    inp16 = inp.bfloat16()
    ws = [_rand(Cout, cin, k, dev=dev, seed=32 + k, scale=0.2) for k in (3, 5, 7)]
    bias = _rand(CT, dev=dev, seed=40)
    gamma, beta = 1 + 0.1 * _rand(CT, dev=dev, seed=41), 0.1 * _rand(CT, dev=dev, seed=42)
    y = torch.empty(Bn, Lc, N, CT, device=dev, dtype=torch.bfloat16)
    y0 = torch.empty_like(y)
    stats = torch.full((Bn * N, 3, 2), float("nan"), device=dev)
    ops.conv_fwd(inp16, *ws, bias, y, Bn, Lc, N, Cout, cin, ld_in, stats=stats)
    ops.conv_fwd(inp16, *ws, bias, y0, Bn, Lc, N, Cout, cin, ld_in)
    assert torch.equal(y, y0)
    yd = y.double().view(Bn, Lc, N, 3, Cout).permute(0, 2, 3, 1, 4).reshape(Bn * N, 3, -1)
    mean, var = yd.mean(-1), yd.var(-1, unbiased=False)
    assert _rel(stats[..., 0], mean) < 1e-5 and _rel(stats[..., 1], 1.0 / torch.sqrt(var + 1e-5)) < 1e-5
    La = (Lc + stride - 1) // stride
    act = torch.empty(Bn, La, N, CT, device=dev, dtype=torch.bfloat16)
    ops.groupnorm_gelu_fwd(y, gamma, beta, act, stats, Bn, Lc, N, Cout, act_stride=stride, stats_given=True)
    xh = (yd - mean[..., None]) / torch.sqrt(var[..., None] + 1e-5)
    xh = xh.view(Bn, N, 3, Lc, Cout).permute(0, 3, 1, 2, 4).reshape(Bn, Lc, N, CT)
    ref = torch.nn.functional.gelu(xh * gamma.double() + beta.double())[:, ::stride]
    assert float((act.double() - ref).abs().max()) < 2.0 ** -8 * float(ref.abs().max()) + 1e-6      # one bf16 rounding
    if ops.gn_y16_ok(Lc, N, Cout):                           # the sequence-resident kernel (own statistics)
        act1, st1 = torch.empty_like(act), torch.empty_like(stats)
        ops.groupnorm_gelu_fwd(y, gamma, beta, act1, st1, Bn, Lc, N, Cout, act_stride=stride)
        assert _rel(st1, stats) < 1e-5
        assert float((act1.float() != act.float()).float().mean()) < 2e-3      # rounding flips only


def test_bf16_weight_gradient_kernel_is_race_free_at_full_size(dev):
    """The head's W1 gradient at the B = 8 size (576 x 2304 over 23 288 rows, 243 blocks on a four-slot LDS ring with counted
    waits): ten launches, bit-identical, and right against fp64 on a sample of the output."""
    from tecmollm import ops
    Mo, No, K = 576, 2304, 23288
    A = _rand(K, Mo, dev=dev, seed=51).bfloat16()
    Bm = _rand(K, No, dev=dev, seed=52).bfloat16()
    split = ops.pick_split_k(Mo, No, K, prec=ops.PREC_BF16)
    first = None
    for _ in range(10):
        C = torch.full((Mo, No), float("nan"), device=dev)
        ops.gemm(Mo, No, K, A, Mo, Bm, No, C, No, a_layout=ops.A_KM, b_layout=ops.B_KN, split_k=split, bf16=True)
        if first is None:
            first = C
        else:
            assert torch.equal(C, first)
    rows = torch.arange(0, Mo, 37, device=dev)
    ref = A[:, rows].double().t() @ Bm.double()
    assert _rel(first[rows], ref) < TOL
