"""CPU oracle of ChimeraLM's second net, `SequenceCNNTransformer`.

TEST INFRASTRUCTURE ONLY -- imported by tests/ and tests/golden/make_golden.py; the product never touches it.

Restates /root/reference/chimeralm/models/components/transformer.py:
  * SinusoidalPositionalEncoding            :7-25   pe[p, 2i] = sin(p * 10000^(-2i/d)), pe[p, 2i+1] = cos(...)
  * SequenceCNNTransformer.__init__         :28-86  Embedding(vocab, d, padding_idx) -> 3 x (Conv1d(d, d, k, padding=1), ReLU,
                                                    MaxPool1d(2, 2)) -> +PE -> LayerNorm -> nn.TransformerEncoder(num_layers x
                                                    TransformerEncoderLayer(d, nhead, dim_feedforward, batch_first=True))
                                                    -> Linear(d, 1) softmax pooling -> Linear(d, d/2), ReLU, Linear(d/2, classes)
  * SequenceCNNTransformer.forward          :88-104
with `nn.TransformerEncoderLayer` at its defaults as the reference constructs it (:64-66): post-norm (norm_first=False), ReLU,
layer_norm_eps 1e-5, no masks -- every position, pads included, attends and is attended (the reference passes neither `mask` nor
`src_key_padding_mask`, :98).  One layer, x [B, L', d]:
    qkv = x W_in^T + b_in ; heads of d/nhead ; a = softmax(q k^T / sqrt(d/nhead)) v ; x = LN1(x + a W_o^T + b_o)
    x = LN2(x + relu(x W_1^T + b_1) W_2^T + b_2)
Configuration of the reference experiment (configs/model/transformer.yaml:3-12): vocab 12, max_len 32768, d_model 256, kernel 3,
12 layers, 8 heads, feed-forward 1024.

PARITY PINNED: tests/golden/transformer_golden.npz holds the logits of the reference module itself (loaded by file path in the build
container) on the seeded weights and inputs that `make_state_dict` / `synthetic_ids` regenerate here.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F


@dataclass(frozen=True)
class Config:
    vocab_size: int = 12
    max_len: int = 32768
    d_model: int = 256
    cnn_kernel_size: int = 3
    num_encoder_layers: int = 12
    nhead: int = 8
    dim_feedforward: int = 1024
    number_of_classes: int = 2
    padding_idx: int = 4
    ln_eps: float = 1e-5


PRODUCTION = Config()


def positional_encoding(max_len: int, d_model: int) -> torch.Tensor:
    """transformer.py:10-19 -- computed in fp32 exactly as the reference does."""
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.unsqueeze(0)


def make_state_dict(seed: int, cfg: Config = PRODUCTION, scale: float = 1.0) -> dict[str, torch.Tensor]:
    """Seeded weights with the reference module's state_dict keys and shapes (numpy PCG64: identical everywhere).
    Magnitudes are those of a trained-ish network rather than of `_init_weights` (biases and LayerNorm affine are non-trivial so
    that every term is exercised); `scale` stretches the classifier for well-separated logits."""
    rng = np.random.default_rng(seed)
    d, ff, k = cfg.d_model, cfg.dim_feedforward, cfg.cnn_kernel_size

    def t(shape, std):
        return torch.from_numpy((rng.standard_normal(shape) * std).astype(np.float32))

    sd = {"embedding.weight": t((cfg.vocab_size, d), 0.5), "pos_encoder.pe": positional_encoding(cfg.max_len, d)}
    for i in (0, 3, 6):
        sd[f"cnn.{i}.weight"] = t((d, d, k), 1.0 / math.sqrt(d * k))
        sd[f"cnn.{i}.bias"] = t((d,), 0.1)
    sd["norm.weight"] = 1.0 + t((d,), 0.1)
    sd["norm.bias"] = t((d,), 0.1)
    for i in range(cfg.num_encoder_layers):
        p = f"transformer_encoder.layers.{i}."
        sd[p + "self_attn.in_proj_weight"] = t((3 * d, d), 1.5 / math.sqrt(d))
        sd[p + "self_attn.in_proj_bias"] = t((3 * d,), 0.1)
        sd[p + "self_attn.out_proj.weight"] = t((d, d), 1.0 / math.sqrt(d))
        sd[p + "self_attn.out_proj.bias"] = t((d,), 0.1)
        sd[p + "linear1.weight"] = t((ff, d), 1.0 / math.sqrt(d))
        sd[p + "linear1.bias"] = t((ff,), 0.1)
        sd[p + "linear2.weight"] = t((d, ff), 1.0 / math.sqrt(ff))
        sd[p + "linear2.bias"] = t((d,), 0.1)
        for n in ("norm1", "norm2"):
            sd[p + n + ".weight"] = 1.0 + t((d,), 0.1)
            sd[p + n + ".bias"] = t((d,), 0.1)
    sd["attn_pool.weight"] = t((1, d), 2.0 / math.sqrt(d))
    sd["attn_pool.bias"] = t((1,), 0.1)
    sd["classifier.0.weight"] = t((d // 2, d), scale / math.sqrt(d))
    sd["classifier.0.bias"] = t((d // 2,), 0.1)
    sd["classifier.3.weight"] = t((cfg.number_of_classes, d // 2), scale / math.sqrt(d // 2))
    sd["classifier.3.bias"] = t((cfg.number_of_classes,), 0.1)
    return sd


def synthetic_ids(seed: int, batch: int, length: int, pads: int = 0) -> np.ndarray:
    """A/C/G/T uniform (7..10), N (11) with p = 0.001, optional left padding with [PAD] = 4 (int64 [B, L])."""
    rng = np.random.default_rng(seed)
    ids = rng.integers(7, 11, size=(batch, length)).astype(np.int64)
    ids[rng.random((batch, length)) < 0.001] = 11
    if pads:
        ids[:, :pads] = 4
    return ids


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """softmax(q k^T / sqrt(dh)) v, all positions, [B, H, L, dh] -> [B, H, L, dh] (no mask: transformer.py:98)."""
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(q.shape[-1])
    return torch.matmul(torch.softmax(s, dim=-1), v)


def encoder_layer(x: torch.Tensor, sd: dict, p: str, cfg: Config) -> torch.Tensor:
    B, L, d = x.shape
    H, dh = cfg.nhead, d // cfg.nhead
    qkv = F.linear(x, sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"])
    q, k, v = (t.reshape(B, L, H, dh).transpose(1, 2) for t in qkv.split(d, dim=-1))
    a = attention(q, k, v).transpose(1, 2).reshape(B, L, d)
    a = F.linear(a, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
    x = F.layer_norm(x + a, (d,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], cfg.ln_eps)
    f = F.linear(F.relu(F.linear(x, sd[p + "linear1.weight"], sd[p + "linear1.bias"])), sd[p + "linear2.weight"],
                 sd[p + "linear2.bias"])
    return F.layer_norm(x + f, (d,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], cfg.ln_eps)


def cnn_stack(x: torch.Tensor, sd: dict, cfg: Config, trace: dict | None = None) -> torch.Tensor:
    """[B, L, d] -> [B, L // 8, d]: three times Conv1d(k, padding=1) + ReLU + MaxPool1d(2, 2) (floor)."""
    x = x.transpose(1, 2)
    for n, i in enumerate((0, 3, 6)):
        x = F.max_pool1d(F.relu(F.conv1d(x, sd[f"cnn.{i}.weight"], sd[f"cnn.{i}.bias"], padding=1)), 2, 2)
        if trace is not None:
            trace[f"cnn{n}"] = x.transpose(1, 2)
    return x.transpose(1, 2)


def forward(ids: torch.Tensor, sd: dict, cfg: Config = PRODUCTION, dtype: torch.dtype = torch.float32,
            trace: dict | None = None) -> torch.Tensor:
    """ids int64 [B, L] -> logits [B, classes]; `trace` collects intermediates by name."""
    sd = {k: v.to(dtype) for k, v in sd.items()}
    x = sd["embedding.weight"][ids]
    x = cnn_stack(x, sd, cfg, trace)
    if x.shape[1] > cfg.max_len:
        raise AssertionError(f"Sequence too long ({x.shape[1]} > {cfg.max_len})")
    x = x + sd["pos_encoder.pe"][:, : x.shape[1]]
    x = F.layer_norm(x, (cfg.d_model,), sd["norm.weight"], sd["norm.bias"], cfg.ln_eps)
    if trace is not None:
        trace["embedded"] = x
    for i in range(cfg.num_encoder_layers):
        x = encoder_layer(x, sd, f"transformer_encoder.layers.{i}.", cfg)
        if trace is not None:
            trace[f"layer{i}"] = x
    w = torch.softmax(F.linear(x, sd["attn_pool.weight"], sd["attn_pool.bias"]), dim=1)
    pooled = (w * x).sum(dim=1)
    if trace is not None:
        trace["pool_weights"], trace["pooled"] = w, pooled
    h = F.relu(F.linear(pooled, sd["classifier.0.weight"], sd["classifier.0.bias"]))
    return F.linear(h, sd["classifier.3.weight"], sd["classifier.3.bias"])
