"""Developer study (CPU, not collected by pytest): where would the 16-bit error of the SequenceCNNTransformer come from --
weights rounded to fp16, or activations rounded to fp16 at the GEMM inputs?  Decides whether hi + lo WEIGHT fragments
(the Hyena path's `fp16c`) could bring a transformer throughput mode inside the reference's 1e-3.
    python tests/tf_error_probe.py [L] [seeds]"""
import math
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from oracle import transformer_oracle as O  # noqa: E402


def r16(t):
    return t.to(torch.float16).to(torch.float32)


def forward(ids, sd, cfg, act16: bool, scale=1.0):
    a = r16 if act16 else (lambda t: t)
    x = sd["embedding.weight"][ids].transpose(1, 2)
    for i in (0, 3, 6):
        x = F.max_pool1d(F.relu(F.conv1d(a(x), sd[f"cnn.{i}.weight"], sd[f"cnn.{i}.bias"], padding=1)), 2, 2)
    x = x.transpose(1, 2)
    x = x + sd["pos_encoder.pe"][:, : x.shape[1]]
    x = F.layer_norm(x, (cfg.d_model,), sd["norm.weight"], sd["norm.bias"], cfg.ln_eps)
    for li in range(cfg.num_encoder_layers):
        p = f"transformer_encoder.layers.{li}."
        B, L, d = x.shape
        H, dh = cfg.nhead, d // cfg.nhead
        qkv = a(F.linear(a(x), sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"]))
        q, k, v = (t.reshape(B, L, H, dh).transpose(1, 2) for t in qkv.split(d, dim=-1))
        s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dh)
        at = torch.matmul(a(torch.softmax(s, dim=-1)), v).transpose(1, 2).reshape(B, L, d)
        at = F.linear(a(at), sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
        x = F.layer_norm(x + at, (d,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], cfg.ln_eps)
        f = F.linear(a(F.relu(F.linear(a(x), sd[p + "linear1.weight"], sd[p + "linear1.bias"]))), sd[p + "linear2.weight"],
                     sd[p + "linear2.bias"])
        x = F.layer_norm(x + f, (d,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], cfg.ln_eps)
    w = torch.softmax(F.linear(x, sd["attn_pool.weight"], sd["attn_pool.bias"]), dim=1)
    pooled = (w * x).sum(dim=1)
    h = F.relu(F.linear(pooled, sd["classifier.0.weight"], sd["classifier.0.bias"]))
    return F.linear(h, sd["classifier.3.weight"], sd["classifier.3.bias"])


GEMM_W = ("cnn.", "in_proj_weight", "out_proj.weight", "linear1.weight", "linear2.weight")

if __name__ == "__main__":
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    cfg = O.PRODUCTION
    torch.set_num_threads(8)
    for seed in range(seeds):
        sd = {k: v.double() for k, v in O.make_state_dict(seed, cfg).items()}
        ids = torch.from_numpy(O.synthetic_ids(100 + seed, 2, L))
        ref = O.forward(ids, sd, cfg, dtype=torch.float64)
        sd32 = {k: v.float() for k, v in sd.items()}
        sdw = {k: (r16(v) if any(g in k for g in GEMM_W) and k.endswith("weight") else v) for k, v in sd32.items()}
        e32 = (forward(ids, sd32, cfg, False).double() - ref).abs().max().item()
        ew = (forward(ids, sdw, cfg, False).double() - ref).abs().max().item()
        ea = (forward(ids, sd32, cfg, True).double() - ref).abs().max().item()
        eb = (forward(ids, sdw, cfg, True).double() - ref).abs().max().item()
        print(f"seed {seed} L {L}: fp32 {e32:.2e} | fp16 weights only {ew:.2e} | fp16 activations only {ea:.2e} | both {eb:.2e}"
              f"   (|logit| max {ref.abs().max().item():.3f})")
