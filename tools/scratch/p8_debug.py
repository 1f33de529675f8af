import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tec-mollm_amd"))
from tecmollm import ops
dev = torch.device("cuda")
def rnd(*s, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed); return (torch.randn(*s, generator=g) * scale).to(dev)
os.environ["TECM_BF16_P8"] = "1"
for (M, N, K) in [(257, 512, 128), (257, 768, 128), (256, 1024, 128), (1024, 1024, 128)]:
    A16, B16 = rnd(M, K, seed=1).bfloat16(), rnd(N, K, seed=2, scale=0.05).bfloat16()
    for fill in (7.0,):
        c = torch.full((M, N), fill, device=dev)
        ops.gemm(M, N, K, A16, K, B16, K, c, N, bf16=True)
        torch.cuda.synchronize()
        ref = (A16.double() @ B16.double().t())
        unw = (c == fill); nan = torch.isnan(c)
        wrong = ((c.double() - ref).abs() > 1e-3) & ~unw & ~nan
        def span(mask):
            cols = mask.any(0).nonzero().flatten(); rows = mask.any(1).nonzero().flatten()
            return (int(cols.min()), int(cols.max()), len(cols), int(rows.min()), int(rows.max()), len(rows)) if len(cols) else None
        print(M, N, K, "unwritten", span(unw), "nan", span(nan), "wrong", span(wrong))
