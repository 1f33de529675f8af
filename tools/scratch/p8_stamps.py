"""Per-block time stamps of the eight-phase GEMM (variant build -DP8_STAMPS): where a tile's time goes inside the full kernel.
   TECM_LIB=tec-mollm_amd/tecmollm/variants/libtecmollm_hip_stamps.so python tools/scratch/p8_stamps.py"""
import ctypes, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tec-mollm_amd"))
import numpy as np, torch
from tecmollm import ops
from tecmollm._lib import lib
dev = torch.device("cuda")
h = ctypes.CDLL(os.environ["TECM_LIB"])
M = 69864
for (N, K, form) in ((3072, 768, "plain"), (3072, 768, "c_fc"), (3072, 768, "dact"), (768, 3072, "plain"), (768, 3072, "resid"), (768, 768, "resid")):
    A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    kw = {}
    if form == "c_fc":
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        kw = dict(bias=torch.randn(N, device=dev), act=ops.ACT_GELU_TANH, preact=(torch.empty(M, N, device=dev, dtype=torch.bfloat16), N))
    elif form == "dact":
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        kw = dict(act=ops.ACT_GELU_TANH, dact_src=(torch.randn(M, N, device=dev).bfloat16(), N))
    elif form == "resid":
        C = torch.empty(M, N, device=dev)
        kw = dict(bias=torch.randn(N, device=dev), residual=(torch.randn(M, N, device=dev), N), out_drop=ops.drop(0.1, 1234, N))
    else:
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm(M, N, K, A, K, B, K, C, N, bf16=True, **kw)
    torch.cuda.synchronize()
    nb = min(8192, ((M + 255) // 256) * ((N + 255) // 256))
    buf = (ctypes.c_ulonglong * (8 * 8192))()
    assert h.tecm_p8_stamps_read(buf, 8 * 8192) == 0
    s = np.ctypeslib.as_array(buf).reshape(8192, 8)[:nb].astype(np.int64)
    t0 = s[:, 0].min()
    ent, kl, st, ack = (s[:, 0] - t0) / 100.0, (s[:, 1] - s[:, 0]) / 100.0, (s[:, 2] - s[:, 1]) / 100.0, (s[:, 3] - s[:, 2]) / 100.0
    span = (s[:, 3].max() - t0) / 100.0
    # per-CU timeline: key = (xcc, se/sh/cu bits of HW_ID)
    key = (s[:, 5] & 0xf) * 65536 + (s[:, 4] & 0xff00)
    gaps, per_cu = [], collections.defaultdict(list)
    for i in range(nb):
        per_cu[int(key[i])].append((s[i, 0], s[i, 3]))
    for k, v in per_cu.items():
        v.sort()
        gaps += [(v[j + 1][0] - v[j][1]) / 100.0 for j in range(len(v) - 1)]
    gaps = np.array(gaps) if gaps else np.zeros(1)
    q = lambda a: f"{np.median(a):7.2f} (p10 {np.percentile(a, 10):6.2f}, p90 {np.percentile(a, 90):6.2f})"
    first = ent < 1.0
    ebuf = (ctypes.c_ulonglong * (16 * 8192))()
    assert h.tecm_p8_epi_stamps_read(ebuf, 16 * 8192) == 0
    e = np.ctypeslib.as_array(ebuf).reshape(8192, 16)[:nb].astype(np.int64)
    rel = (e[:, :12] - s[:, 1:2]) / 100.0            # us after the K loop's end, wave 0
    med = np.median(rel, axis=0)
    print("   wave 0, us after the K loop: " + " | ".join(f"slab {i}: park {med[3*i]:5.2f}->{med[3*i+1]:5.2f}, rows ->{med[3*i+2]:5.2f}" for i in range(4)))
    print(f"N={N} K={K} {form:6s}: {nb} blocks on {len(per_cu)} CUs, span {span:7.1f} us | K loop {q(kl)} | epilogue to stores issued {q(st)} | "
          f"stores acked {q(ack)} | gap to next block on the CU {q(gaps)} | K loop of the first round {np.median(kl[first]):6.2f}, later {np.median(kl[~first]):6.2f}", flush=True)
