"""Developer study (CPU, not collected by pytest), round 4: which 16-bit ACTIVATIONS of the SequenceCNNTransformer would have to carry an e5m2 lo byte (the
Hyena path's round-4 machinery, ~15 bits) for its fp16c mode to pass the 1e-3 gate on the two cases the reference module itself was run on?  float64 forward,
weights exact (they are compensated), one operand class switched at a time.    python tests/tf_lo_probe.py   (output: profiles/r04_tf_lo_probe.txt)"""
import math, sys
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, ".")
from oracle import transformer_oracle as O
def r16(t): return t.to(torch.float16).to(t.dtype)
def h5(t):
    hi = t.to(torch.float16).to(t.dtype); lo = t - hi
    m, e = torch.frexp(lo); return hi + torch.ldexp(torch.round(m * 8) / 8, e)
def rd(t, mode): return t if mode is None else (r16(t) if mode == "h" else h5(t))
def forward(ids, sd, cfg, c):
    g = lambda k: c.get(k, "h")
    x = sd["embedding.weight"][ids].transpose(1, 2)
    for i in (0, 3, 6):
        x = F.max_pool1d(F.relu(F.conv1d(rd(x, g("conv")), sd[f"cnn.{i}.weight"], sd[f"cnn.{i}.bias"], padding=1)), 2, 2)
    x = x.transpose(1, 2)
    x = x + sd["pos_encoder.pe"][:, : x.shape[1]]
    x = F.layer_norm(x, (cfg.d_model,), sd["norm.weight"], sd["norm.bias"], cfg.ln_eps)
    for li in range(cfg.num_encoder_layers):
        p = f"transformer_encoder.layers.{li}."
        B, L, d = x.shape; H, dh = cfg.nhead, d // cfg.nhead
        qkv = rd(F.linear(rd(x, g("hx")), sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"]), g("qkv"))
        q, k, v = (t.reshape(B, L, H, dh).transpose(1, 2) for t in qkv.split(d, dim=-1))
        if g("v") != g("qkv"): v = rd(F.linear(rd(x, g("hx")), sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"]), g("v")).split(d, dim=-1)[2].reshape(B, L, H, dh).transpose(1, 2)
        s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dh)
        at = torch.matmul(rd(torch.softmax(s, dim=-1), g("p")), v).transpose(1, 2).reshape(B, L, d)
        at = F.linear(rd(at, g("att")), sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
        x = F.layer_norm(x + at, (d,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], cfg.ln_eps)
        f = F.linear(rd(F.relu(F.linear(rd(x, g("x1")), sd[p + "linear1.weight"], sd[p + "linear1.bias"])), g("hid")), sd[p + "linear2.weight"], sd[p + "linear2.bias"])
        x = F.layer_norm(x + f, (d,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], cfg.ln_eps)
    w = torch.softmax(F.linear(x, sd["attn_pool.weight"], sd["attn_pool.bias"]), dim=1)
    pooled = (w * x).sum(dim=1)
    h = F.relu(F.linear(pooled, sd["classifier.0.weight"], sd["classifier.0.bias"]))
    return F.linear(h, sd["classifier.3.weight"], sd["classifier.3.bias"])
cfg = O.PRODUCTION
torch.set_num_threads(8)
for seed, B, L, pads in ((0, 2, 1000, 0), (1, 3, 777, 40)):
    sd = {k: v.double() for k, v in O.make_state_dict(seed, cfg, scale=3.0).items()}
    ids = torch.from_numpy(O.synthetic_ids(100 + seed, B, L, pads))
    ref = forward(ids, sd, cfg, {k: None for k in ("conv","hx","qkv","v","p","att","x1","hid")})
    base = {"att": "h5"}     # the shipped fp16c: attention output hi + lo, weights compensated (exact here)
    cases = {"fp16c r03 (att hi+lo)": base, "+ qkv": {**base, "qkv": "h5", "v": "h5"}, "+ qk only": {**base, "qkv": "h5", "v": "h"},
             "+ hx": {**base, "hx": "h5"}, "+ x1": {**base, "x1": "h5"}, "+ hid": {**base, "hid": "h5"}, "+ conv": {**base, "conv": "h5"}, "+ p": {**base, "p": "h5"},
             "+ qkv + hx": {**base, "qkv": "h5", "v": "h5", "hx": "h5"}, "+ qkv + hx + x1": {**base, "qkv": "h5", "v": "h5", "hx": "h5", "x1": "h5"},
             "+ qkv + hx + x1 + hid + conv": {**base, "qkv": "h5", "v": "h5", "hx": "h5", "x1": "h5", "hid": "h5", "conv": "h5"}}
    for n, c in cases.items():
        e = (forward(ids, sd, cfg, c) - ref).abs().max().item()
        print(f"seed {seed} {B}x{L}: {n:34s} max |dlogit| {e:.2e}  (max |logit| {ref.abs().max().item():.2f})", flush=True)
