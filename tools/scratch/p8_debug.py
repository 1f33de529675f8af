import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tec-mollm_amd"))
from tecmollm import ops
dev = torch.device("cuda")
def rnd(*s, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed); return (torch.randn(*s, generator=g) * scale).to(dev)
def run(M, N, K):
    A16, B16 = rnd(M, K, seed=1).bfloat16(), rnd(N, K, seed=2, scale=0.05).bfloat16()
    bias, res, pre_src = rnd(N, seed=4), rnd(M, N, seed=5), rnd(M, N, seed=6)
    c16, pre = torch.empty(M, N, device=dev, dtype=torch.bfloat16), torch.full((M, N), float("nan"), device=dev)
    ops.gemm(M, N, K, A16, K, B16, K, c16, N, bias=bias, act=ops.ACT_GELU_TANH, preact=(pre, N), bf16=True)
    c = torch.full((M, N), float("nan"), device=dev)
    ops.gemm(M, N, K, A16, K, B16, K, c, N, bias=bias, out_drop=ops.drop(0.1, 77, N), residual=(res, N), bf16=True)
    d = torch.full((M, N), float("nan"), device=dev)
    ops.gemm(M, N, K, A16, K, B16, K, d, N, act=ops.ACT_GELU_TANH, dact_src=(pre_src, N), bf16=True)
    torch.cuda.synchronize()
    return c16.float(), pre, c, d
for (M, N, K) in [(256, 800, 2304), (256, 800, 128)]:
    os.environ["TECM_BF16_DMA"] = "1"; os.environ.pop("TECM_BF16_P8", None)
    want = run(M, N, K)
    os.environ.pop("TECM_BF16_DMA"); os.environ["TECM_BF16_P8"] = "1"
    got = run(M, N, K)
    for name, g_, w_ in zip(("c16", "pre", "c", "d"), got, want):
        wrong = ((g_.double() - w_.double()).abs() > 1e-2 * (1 + w_.double().abs())) | (torch.isnan(g_) != torch.isnan(w_))
        cols = wrong.any(0).nonzero().flatten(); rows = wrong.any(1).nonzero().flatten()
        print(M, N, K, name, "wrong", int(wrong.sum()), "cols", cols[:12].tolist(), "...", cols[-3:].tolist(), "rows", rows[:8].tolist(), "nan got/want", int(torch.isnan(g_).sum()), int(torch.isnan(w_).sum()))
