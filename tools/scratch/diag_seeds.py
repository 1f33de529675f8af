import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tec-mollm_amd"))
from oracle import ref_cpu as R
from tests.parity import compare_forward_backward
for name, kw in (("N20", dict(L_in=48, L_out=12, num_nodes=20)), ("L96", dict(L_in=96, L_out=24, num_nodes=20))):
    for seed in range(31, 41):
        cfg = R.default_config(**kw)
        res = compare_forward_backward(cfg, B=2, grid=(4, 5), threshold_km=170.0, gat_graphs="per_timestep", seed=seed, train=True,
                                       precision="bf16")
        top = sorted(res["per_param"].items(), key=lambda kv: -kv[1][1])[:3]
        topr = sorted(res["per_param"].items(), key=lambda kv: -kv[1][0])[:2]
        print(name, seed, "fwd", round(res["fwd_rel"], 4), "grad max-norm", round(res["grad_rel_max"], 4), [(k[-30:], round(v[0], 4)) for k, v in topr],
              "elem", [(k[-30:], round(v[1], 2)) for k, v in top], flush=True)
