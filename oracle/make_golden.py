"""Generate the golden vectors under tests/golden/ from the REFERENCE's own code.

Run in the authoring container only (needs /root/reference, which never travels):

    python oracle/make_golden.py

What is imported from the reference: the pure-torch classes of
`/root/reference/src/model/modules.py` -- SpatioTemporalEmbedding (:211-266),
TemporalEncoder (:121-154, with Multi_Scale_Conv_Block :13-60, MultiScaleConvEmbedder
:62-88, LatentPatchingProjection :90-119) and PredictionHead (:268-313).  That module
also imports `peft` and `torch_geometric`, which are not installed anywhere in this
environment (ordinary ModuleNotFoundError, not a permission denial); the two names are
pre-seeded in `sys.modules` as EMPTY placeholders purely so the import statement
succeeds -- nothing of GATv2Conv or LoRA is emulated, and the classes that need them
(SpatialEncoder, LLMBackbone) are never instantiated here.  Those two stages stay
"parity unpinned" (see oracle/ref_cpu.py header).

GPT-2 block math is pinned against the installed `transformers` GPT2Model
(the reference's call sites modules.py:165,170,208) with config-initialised weights.

Fixtures store inputs + expected outputs only.  Parameters are regenerated on the test
side from `ref_cpu.init_params(cfg, seed)`; a checksum of them is stored so a drift in
the generator is caught instead of silently changing the comparison.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_cpu as R  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def _import_reference_modules():
    for name in ("peft", "torch_geometric", "torch_geometric.nn"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["peft"].get_peft_model = None
    sys.modules["peft"].LoraConfig = None
    sys.modules["torch_geometric.nn"].GATv2Conv = None
    sys.path.insert(0, REF)
    import importlib
    return importlib.import_module("src.model.modules")


def param_checksum(p: dict) -> float:
    return float(sum(v.double().abs().sum().item() * (1 + (i % 7)) for i, (k, v) in enumerate(sorted(p.items()))))


def sub_state(p: dict, prefix: str) -> dict:
    return {k[len(prefix):]: v for k, v in p.items() if k.startswith(prefix)}


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    M = _import_reference_modules()

    # ---------------------------------------------------------------- a-1 embedding
    cfg = R.default_config(L_in=48, L_out=12, num_nodes=37, c_in=6, d_emb=16)
    p = R.init_params(cfg, seed=11)
    x, tf, _ = R.synthetic_batch(2, 5, 37, 6, 12, seed=21)
    emb = M.SpatioTemporalEmbedding(d_emb=16, num_nodes=37, num_years=13)
    emb.load_state_dict(sub_state(p, R.P_EMB))
    with torch.no_grad():
        out = emb(x, tf)
    np.savez(os.path.join(OUT, "embed_small.npz"), x=x.numpy(), tf=tf[:, :, 0, :].contiguous().numpy(),
             out=out.numpy(), seed=11, num_nodes=37, checksum=param_checksum(p))

    # full-N gather (bit-exactness of the index path at N=2911): one (b,t) slab
    cfgN = R.default_config(num_nodes=2911)
    pN = {k: v for k, v in R.init_params(cfgN, seed=12).items() if k.startswith(R.P_EMB)}
    xN, tfN, _ = R.synthetic_batch(1, 2, 2911, 6, 12, seed=22)
    embN = M.SpatioTemporalEmbedding(d_emb=16, num_nodes=2911, num_years=13)
    embN.load_state_dict(sub_state(pN, R.P_EMB))
    with torch.no_grad():
        outN = embN(xN, tfN)
    np.savez_compressed(os.path.join(OUT, "embed_fullN.npz"), tf=tfN[:, :, 0, :].contiguous().numpy(),
                        out_emb=outN[..., 6:].numpy(), seed=12, data_seed=22)

    # ---------------------------------------------------------------- a-4/a-5 temporal encoder
    for L_in, tag in ((48, "L48"), (96, "L96")):
        cfg = R.default_config(L_in=L_in, L_out=12, num_nodes=8)
        p = R.init_params(cfg, seed=13)
        g = torch.Generator().manual_seed(23)
        xt = torch.randn(6, L_in, 22, generator=g)
        te = M.TemporalEncoder(22, cfg["temporal_channel_list"], cfg["temporal_strides"], cfg["patch_len"], 768)
        te.load_state_dict(sub_state(p, "temporal_encoder."))
        with torch.no_grad():
            out = te(xt)
            blk0 = te.conv_embedder.embedder[0](xt.permute(0, 2, 1))
        np.savez(os.path.join(OUT, f"temporal_{tag}.npz"), x=xt.numpy(), out=out.numpy(),
                 block0=blk0.numpy(), seed=13, L_in=L_in, checksum=param_checksum(p))

    # ---------------------------------------------------------------- a-8 prediction head
    cfg = R.default_config(L_in=48, L_out=12, num_nodes=8)
    p = R.init_params(cfg, seed=14)
    g = torch.Generator().manual_seed(24)
    xh = torch.randn(5, 3, 768, generator=g)
    ph = M.PredictionHead(input_dim=2304, output_dim=12).eval()
    ph.load_state_dict(sub_state(p, "prediction_head."))
    with torch.no_grad():
        out = ph(xh)
    np.savez(os.path.join(OUT, "head.npz"), x=xh.numpy(), out=out.numpy(), seed=14, checksum=param_checksum(p))

    # ---------------------------------------------------------------- a-6 GPT-2 trunk (LoRA B = 0)
    from transformers import GPT2Config, GPT2Model
    for T, tag in ((3, "T3"), (6, "T6")):
        cfg = R.default_config(L_in=16 * T, L_out=12, num_nodes=8)
        p = R.init_params(cfg, seed=15)
        gm = GPT2Model(GPT2Config())
        gm.h = gm.h[:3]
        sd = gm.state_dict()
        for k in list(sd.keys()):
            if k == "wte.weight":
                continue
            src = (R.P_GPT + k).replace("attn.c_attn.weight", "attn.c_attn.base_layer.weight") \
                               .replace("attn.c_attn.bias", "attn.c_attn.base_layer.bias")
            if src not in p:                           # persistent buffers only; a weight the oracle does not carry is a bug
                assert k.endswith((".attn.bias", ".attn.masked_bias")), f"GPT2Model key {k} has no oracle parameter"
                continue
            assert tuple(sd[k].shape) == tuple(p[src].shape), (k, tuple(sd[k].shape), tuple(p[src].shape))
            sd[k] = p[src]
        used = {(R.P_GPT + k).replace("attn.c_attn.weight", "attn.c_attn.base_layer.weight")
                .replace("attn.c_attn.bias", "attn.c_attn.base_layer.bias") for k in sd}
        unused = [k for k in p if k.startswith(R.P_GPT) and "lora_" not in k and k not in used]
        assert not unused, f"oracle GPT-2 parameters without a GPT2Model key: {unused}"
        gm.load_state_dict(sd)
        gm.eval()
        g = torch.Generator().manual_seed(25)
        xe = torch.randn(4, T, 768, generator=g) * 0.5
        with torch.no_grad():
            out = gm(inputs_embeds=xe, attention_mask=torch.ones(4, T, dtype=torch.long)).last_hidden_state
        np.savez(os.path.join(OUT, f"gpt2_{tag}.npz"), x=xe.numpy(), out=out.numpy(), seed=15, T=T,
                 checksum=param_checksum(p))
    print("golden vectors written to", OUT)
    for f in sorted(os.listdir(OUT)):
        print(f"  {f:24s} {os.path.getsize(os.path.join(OUT, f)) / 1024:8.1f} KiB")


if __name__ == "__main__":
    main()
