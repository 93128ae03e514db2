"""Developer study (CPU, not collected by pytest): ONE 16-bit activation tensor of the SequenceCNNTransformer rounded to fp16 at a
time (fp64 otherwise) -- which rounding moves the logits?  Result (HISTORY.md section 5b): the attention output, 2.2-3.9e-3 alone.
    python tests/tf_error_rank.py"""
import math, sys
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, ".")
from oracle import transformer_oracle as O

def r16(t): return t.to(torch.float16).to(torch.float64)
ident = lambda t: t

def forward(ids, sd, cfg, R):
    g = lambda k: (r16 if k in R else ident)
    x = sd["embedding.weight"][ids].transpose(1, 2)
    for i in (0, 3, 6):
        x = F.max_pool1d(F.relu(F.conv1d(g("conv")(x), sd[f"cnn.{i}.weight"], sd[f"cnn.{i}.bias"], padding=1)), 2, 2)
    x = g("conv")(x).transpose(1, 2)
    x = x + sd["pos_encoder.pe"][:, : x.shape[1]]
    x = F.layer_norm(x, (cfg.d_model,), sd["norm.weight"], sd["norm.bias"], cfg.ln_eps)
    for li in range(cfg.num_encoder_layers):
        p = f"transformer_encoder.layers.{li}."
        B, L, d = x.shape
        H, dh = cfg.nhead, d // cfg.nhead
        qkv = g("qkv")(F.linear(g("hx")(x), sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"]))
        q, k, v = (t.reshape(B, L, H, dh).transpose(1, 2) for t in qkv.split(d, dim=-1))
        s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dh)
        at = torch.matmul(g("P")(torch.softmax(s, dim=-1)), v).transpose(1, 2).reshape(B, L, d)
        at = F.linear(g("att")(at), sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
        x = F.layer_norm(x + at, (d,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], cfg.ln_eps)
        f = F.linear(g("u")(F.relu(F.linear(g("x1")(x), sd[p + "linear1.weight"], sd[p + "linear1.bias"]))), sd[p + "linear2.weight"], sd[p + "linear2.bias"])
        x = F.layer_norm(x + f, (d,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], cfg.ln_eps)
    w = torch.softmax(F.linear(x, sd["attn_pool.weight"], sd["attn_pool.bias"]), dim=1)
    pooled = (w * x).sum(dim=1)
    h = F.relu(F.linear(pooled, sd["classifier.0.weight"], sd["classifier.0.bias"]))
    return F.linear(h, sd["classifier.3.weight"], sd["classifier.3.bias"])

cfg = O.PRODUCTION
torch.set_num_threads(8)
ALL = ["conv", "hx", "qkv", "P", "att", "x1", "u"]
for seed, B, L, pads in ((0, 2, 1000, 0), (1, 3, 777, 40), (3, 2, 2055, 0)):
    sd = {k: v.double() for k, v in O.make_state_dict(seed, cfg, scale=3.0).items()}
    ids = torch.from_numpy(O.synthetic_ids(100 + seed, B, L, pads))
    ref = O.forward(ids, sd, cfg, dtype=torch.float64)
    row = f"seed {seed} L {L} |logit| {ref.abs().max():.1f}:"
    for R in [[k] for k in ALL] + [ALL, ["hx", "x1"], ["qkv", "P", "att"], ["conv", "u"]]:
        e = (forward(ids, sd, cfg, set(R)).double() - ref).abs().max().item()
        row += f"  {'+'.join(R) if len(R) < 7 else 'ALL'} {e:.1e}"
    print(row, flush=True)
