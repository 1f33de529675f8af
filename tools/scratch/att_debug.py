import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tec-mollm_amd")); sys.path.insert(0, ROOT)
from tecmollm import ops
from tests.test_gpu_ops import _ref_attention, _rand, _rel
dev = torch.device("cuda")
for T in (13, 16, 20, 21, 22, 24):
    Bn, N, H, D = 2, 7, 12, 768
    qkv = _rand(Bn, T, N, 3 * D, dev=dev, seed=1, scale=0.5)
    qd = qkv.double().requires_grad_(True)
    ref = _ref_attention(qd, Bn, T, N, H, D)
    dctx = _rand(Bn, T, N, D, dev=dev, seed=2)
    (g,) = torch.autograd.grad(ref, qd, dctx.double())
    outs = []
    for rep in range(3):
        dqkv = torch.full_like(qkv, float("nan"))
        ops.attention_bwd(qkv, dctx, dqkv, Bn, T, N, H, D)
        torch.cuda.synchronize()
        outs.append(dqkv.clone())
    err = (outs[0].double() - g).abs()
    bad = err > 1e-3 * g.abs().max()
    # which of q / k / v parts, which time rows
    parts = [int(bad[..., i * D:(i + 1) * D].sum()) for i in range(3)]
    trows = bad.any(-1).any(-1).any(0).nonzero().flatten().tolist()
    print(T, "rel", _rel(outs[0], g), "bad q/k/v", parts, "time rows", trows, "nan", int(torch.isnan(outs[0]).sum()), "repro", torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2]), flush=True)
