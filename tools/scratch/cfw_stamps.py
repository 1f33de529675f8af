"""Time stamps of conv_fwd_seq_kernel's wave 0 (variant build -DCFW_STAMPS): staging, per-unit K loop and store phases.
   python tools/build_variant.py conv_seq.hip cfwstamps -DCFW_STAMPS
   TECM_LIB=tec-mollm_amd/tecmollm/variants/libtecmollm_hip_cfwstamps.so python tools/scratch/cfw_stamps.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tec-mollm_amd"))
import numpy as np, torch
from tecmollm import ops
dev = torch.device("cuda")
h = ctypes.CDLL(os.environ["TECM_LIB"])
B, N = 8, 2911
for ld_in, cin, Cout, L in ((24, 22, 64, 48), (64, 64, 128, 24)):
    torch.manual_seed(0)
    inp = torch.randn(B, L, N, ld_in, device=dev).bfloat16()
    w = [torch.randn(Cout, cin, k, device=dev) * 0.1 for k in (3, 5, 7)]
    bias = torch.randn(3 * Cout, device=dev)
    y = torch.empty(B, L, N, 3 * Cout, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        ops.conv_fwd(inp, w[0], w[1], w[2], bias, y, B, L, N, Cout, cin, ld_in)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (16 * 8192))()
    assert h.tecm_cfw_stamps_read(buf, 16 * 8192) == 0
    nb = min(8192, B * ((N + 3) // 4))
    s = np.ctypeslib.as_array(buf).reshape(8192, 16)[:nb].astype(np.int64)
    d = lambda a, b: float(np.median((s[:, b] - s[:, a]) / 100.0))
    span = (s[:, 8].max() - s[:, 0].min()) / 100.0
    print(f"Cout={Cout} L={L}: {nb} blocks, span {span:.1f} us | staging {d(0,1):.2f} | unit0 K loop {d(1,2):.2f}, stores {d(2,3):.2f} | "
          f"unit1 K loop {d(3,4):.2f}, stores {d(4,5):.2f} | unit2 K loop {d(5,6):.2f}, stores {d(6,7):.2f} | block {d(0,8):.2f} us", flush=True)
