import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tec-mollm_amd"))
from oracle import ref_cpu as R
from tests.parity import compare_forward_backward
for seed in (36, 37, 38):
    cfg = R.default_config(L_in=336, L_out=12, num_nodes=20, llm_layers=2)
    res = compare_forward_backward(cfg, B=2, grid=(4, 5), threshold_km=170.0, gat_graphs="per_timestep", seed=seed, train=True, precision="bf16")
    top = sorted(res["per_param"].items(), key=lambda kv: -kv[1][1])[:5]
    print(seed, "fwd", round(res["fwd_rel"], 4), "grad_rel_max", round(res["grad_rel_max"], 4), [(k[-40:], round(v[1], 2)) for k, v in top], flush=True)
