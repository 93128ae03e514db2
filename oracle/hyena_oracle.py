"""CPU restatement of the forward pass behind `chimeralm predict` (test oracle).

TEST INFRASTRUCTURE -- never imported by the product package `chimeralm_amd`.

What is restated, and from where:

* Call protocol `HyenaDna.forward`          /root/reference/chimeralm/models/components/hyena.py:244-256
  (backbone gets only `input_ids`; the head is called with attention_mask=None,
  so pad tokens are *not* masked anywhere).
* Head `BinarySequenceClassifier`           hyena.py:50-53 (attention pooling), :56-74 (MLP),
  `ResidualBlock`                           hyena.py:149-180.   PINNED by tests/golden/head_*.npz,
  generated from the reference class itself.
* Backbone HyenaDNA-small-32k               NOT in /root/reference.  It is Hugging Face Hub remote
  code `LongSafari/hyenadna-small-32k-seqlen-hf` (unpinned revision, loaded at hyena.py:237 via
  transformers; uv.lock pins transformers==4.57.0, torch==2.5.1).  The functions below restate the
  published HyenaDNA algorithm (embedding -> 4 x [LN, Hyena operator order 2, LN, GELU-tanh MLP]
  -> LN) as summarised in SURVEY.md section 8(a) rows 5-11 / Appendix A.  PARITY UNPINNED: the
  reference has no golden vector for it and the remote code/weights are unreachable offline.

All state lives in a flat `dict[str, Tensor]` keyed exactly like the reference checkpoint
(`net.backbone.backbone.layers.0.mixer.in_proj.weight`, `net.head.attention.0.weight`, ...), see
`make_state_dict`.  Arithmetic is torch-CPU eager; `dtype=torch.float32` reproduces the reference's
precision, `torch.float64` is used by tests as a "ground truth" to rank two fp32 results.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------- config
D_MODEL = 256
N_LAYER = 4
D_INNER = 1024
VOCAB_ROWS = 16          # vocab_size 12 padded to a multiple of 8
FILTER_ORDER = 64
EMB_DIM = 5
MAX_SEQ_LEN = 32770
LN_EPS = 1e-5
HEAD_HIDDEN = 512
MOD_SHIFT = 0.05

BB = "net.backbone.backbone."   # HyenaDna.backbone (HF model) . backbone (HyenaLMBackbone)
HD = "net.head."


# ----------------------------------------------------------------------------- weights
def _linear_init(rng, out_f, in_f, bias=True):
    bound = 1.0 / math.sqrt(in_f)
    w = torch.from_numpy(rng.uniform(-bound, bound, size=(out_f, in_f))).float()
    b = torch.from_numpy(rng.uniform(-bound, bound, size=(out_f,))).float() if bias else None
    return w, b


def positional_embedding_init(seq_len=MAX_SEQ_LEN, emb_dim=EMB_DIM):
    """Initial value of `filter_fn.pos_emb.{z,t}` (HyenaDNA `HyenaPositionalEmbedding.__init__`)."""
    t = torch.linspace(0, 1, seq_len, dtype=torch.float64)[None, :, None]
    bands = (emb_dim - 1) // 2
    t_rescaled = torch.linspace(0, seq_len - 1, seq_len, dtype=torch.float64)[None, :, None]
    w = 2 * math.pi * t_rescaled / seq_len
    f = torch.linspace(1e-4, bands - 1, bands, dtype=torch.float64)[None, None]
    z = torch.exp(-1j * f * w)
    z = torch.cat([t, z.real, z.imag], dim=-1)
    return z.float(), t.float()


def modulation_deltas_init(d_model=D_MODEL, fast_decay_pct=0.3, slow_decay_pct=1.5, target=1e-2):
    """`filter_fn.modulation.deltas` buffer (HyenaDNA `HyenaExponentialModulation.__init__`)."""
    max_decay = math.log(target) / fast_decay_pct
    min_decay = math.log(target) / slow_decay_pct
    return torch.linspace(min_decay, max_decay, d_model, dtype=torch.float64)[None, None].float()


def make_state_dict(seed: int = 0, *, head_scale: float = 1.0) -> dict[str, torch.Tensor]:
    """Seeded synthetic weights with the reference checkpoint's key names and shapes.

    Values are nn.Linear-style uniform(+-1/sqrt(fan_in)); LayerNorm affine parameters are perturbed
    away from (1, 0) and the learned `freq`/`z` are perturbed so every parameter is exercised.
    `head_scale` widens the final logit margins so label-identity checks are meaningful.
    """
    import numpy as np

    g = np.random.default_rng(seed)      # PCG64: stream is stable across platforms / numpy versions
    sd: dict[str, torch.Tensor] = {}

    def rnd(*shape, scale=1.0):
        return torch.from_numpy(g.standard_normal(shape) * scale).float()

    sd[BB + "embeddings.word_embeddings.weight"] = rnd(VOCAB_ROWS, D_MODEL)
    z0, t0 = positional_embedding_init()
    for i in range(N_LAYER):
        p = f"{BB}layers.{i}."
        for nm in ("norm1", "norm2"):
            sd[p + nm + ".weight"] = 1.0 + rnd(D_MODEL, scale=0.1)
            sd[p + nm + ".bias"] = rnd(D_MODEL, scale=0.1)
        w, b = _linear_init(g, 3 * D_MODEL, D_MODEL)
        sd[p + "mixer.in_proj.weight"], sd[p + "mixer.in_proj.bias"] = w, b
        w, b = _linear_init(g, D_MODEL, D_MODEL)
        sd[p + "mixer.out_proj.weight"], sd[p + "mixer.out_proj.bias"] = w, b
        # depthwise Conv1d(768, 768, 3, padding=2, groups=768): fan_in = 3
        bound = 1.0 / math.sqrt(3.0)
        sd[p + "mixer.short_filter.weight"] = torch.from_numpy(g.uniform(-bound, bound, size=(3 * D_MODEL, 1, 3))).float()
        sd[p + "mixer.short_filter.bias"] = torch.from_numpy(g.uniform(-bound, bound, size=(3 * D_MODEL,))).float()
        f = p + "mixer.filter_fn."
        sd[f + "bias"] = rnd(D_MODEL)
        sd[f + "pos_emb.z"] = z0 + rnd(1, MAX_SEQ_LEN, EMB_DIM, scale=1e-3)
        sd[f + "pos_emb.t"] = t0.clone()
        w, b = _linear_init(g, FILTER_ORDER, EMB_DIM)
        sd[f + "implicit_filter.0.weight"], sd[f + "implicit_filter.0.bias"] = w, b
        freq = 10.0 + rnd(1, FILTER_ORDER, scale=0.5)
        for j in (1, 3, 5):     # one shared HyenaSin module registered three times
            sd[f + f"implicit_filter.{j}.freq"] = freq
        for j in (2, 4):
            w, b = _linear_init(g, FILTER_ORDER, FILTER_ORDER)
            sd[f + f"implicit_filter.{j}.weight"], sd[f + f"implicit_filter.{j}.bias"] = w, b
        w, _ = _linear_init(g, D_MODEL, FILTER_ORDER, bias=False)
        sd[f + "implicit_filter.6.weight"] = w
        sd[f + "modulation.deltas"] = modulation_deltas_init()
        w, b = _linear_init(g, D_INNER, D_MODEL)
        sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"] = w, b
        w, b = _linear_init(g, D_MODEL, D_INNER)
        sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"] = w, b
    sd[BB + "ln_f.weight"] = 1.0 + rnd(D_MODEL, scale=0.1)
    sd[BB + "ln_f.bias"] = rnd(D_MODEL, scale=0.1)

    def head_lin(name, out_f, in_f):
        w, b = _linear_init(g, out_f, in_f)
        sd[HD + name + ".weight"], sd[HD + name + ".bias"] = w * head_scale, b * head_scale

    head_lin("attention.0", HEAD_HIDDEN // 2, D_MODEL)
    head_lin("attention.2", 1, HEAD_HIDDEN // 2)
    head_lin("classifier.0", HEAD_HIDDEN, D_MODEL)
    head_lin("classifier.3", HEAD_HIDDEN, HEAD_HIDDEN)
    head_lin("classifier.6.layers.0", HEAD_HIDDEN, HEAD_HIDDEN)
    head_lin("classifier.6.layers.3", HEAD_HIDDEN, HEAD_HIDDEN)
    head_lin("output_layer", 2, HEAD_HIDDEN)
    return sd


# ----------------------------------------------------------------------------- backbone pieces
def _lin(x, sd, key, dt):
    b = sd.get(key + ".bias")
    return F.linear(x, sd[key + ".weight"].to(dt), None if b is None else b.to(dt))


def _ln(x, sd, key, dt):
    return F.layer_norm(x, (x.shape[-1],), sd[key + ".weight"].to(dt), sd[key + ".bias"].to(dt), LN_EPS)


def hyena_filter(sd, layer: int, L: int, dt=torch.float32) -> torch.Tensor:
    """Implicit long filter k[L, 256] of one layer (HyenaDNA `HyenaFilter.filter`).

    z[:L] -> Linear(5,64) -> sin(freq*.) -> 2 x (Linear(64,64) -> sin(freq*.)) -> Linear(64,256,no bias)
    -> * (exp(-t*|deltas|) + 0.05).   SURVEY.md section 8(a) row 8.
    """
    f = f"{BB}layers.{layer}.mixer.filter_fn."
    z = sd[f + "pos_emb.z"][0, :L].to(dt)
    t = sd[f + "pos_emb.t"][0, :L].to(dt)              # [L, 1]
    freq = sd[f + "implicit_filter.1.freq"].to(dt)     # [1, 64]
    h = torch.sin(freq * _lin(z, sd, f + "implicit_filter.0", dt))
    h = torch.sin(freq * _lin(h, sd, f + "implicit_filter.2", dt))
    h = torch.sin(freq * _lin(h, sd, f + "implicit_filter.4", dt))
    h = _lin(h, sd, f + "implicit_filter.6", dt)       # [L, 256]
    deltas = sd[f + "modulation.deltas"].to(dt)[0]     # [1, 256]
    decay = torch.exp(-t * deltas.abs())
    return h * (decay + MOD_SHIFT)


def fftconv(u, k, D):
    """HyenaDNA `fftconv`: causal linear convolution through a size-2L real FFT plus skip term.

    u [B, C, L], k [C, L], D [C].  y = irfft(rfft(u,2L) * rfft(k,2L)/2L, n=2L, norm='forward')[:L] + u*D
    """
    L = u.shape[-1]
    n = 2 * L
    k_f = torch.fft.rfft(k, n=n) / n
    u_f = torch.fft.rfft(u, n=n)
    y = torch.fft.irfft(u_f * k_f, n=n, norm="forward")[..., :L]
    return y + u * D.unsqueeze(-1)


def direct_causal_conv(u, k, D):
    """Same operator as `fftconv`, written as the defining sum (float64-friendly cross-check)."""
    B, C, L = u.shape
    y = torch.zeros_like(u)
    for t in range(L):
        y[:, :, t] = (u[:, :, : t + 1] * k[:, : t + 1].flip(-1)).sum(-1)
    return y + u * D.unsqueeze(-1)


def short_filter(z, sd, layer, dt):
    """Depthwise Conv1d(768,768,k=3,padding=2,groups=768)(z)[..., :L]  (causal 3-tap FIR per channel)."""
    p = f"{BB}layers.{layer}.mixer.short_filter."
    L = z.shape[-1]
    return F.conv1d(z, sd[p + "weight"].to(dt), sd[p + "bias"].to(dt), padding=2, groups=z.shape[1])[..., :L]


def hyena_operator(u, sd, layer, dt, trace=None):
    """Order-2 Hyena mixer (HyenaDNA `HyenaOperator.forward`), SURVEY.md section 8(a) row 7.

    u [B, L, 256] (already LayerNorm-ed) -> [B, L, 256]
    """
    p = f"{BB}layers.{layer}.mixer."
    L = u.shape[1]
    z = _lin(u, sd, p + "in_proj", dt).transpose(1, 2)        # [B, 768, L]
    if trace is not None:
        trace[f"l{layer}.in_proj"] = z
    uc = short_filter(z, sd, layer, dt)
    x0, x1, v = uc.split(D_MODEL, dim=1)
    k = hyena_filter(sd, layer, L, dt).transpose(0, 1)        # [256, L]
    bias = sd[p + "filter_fn.bias"].to(dt)
    v = v * x1
    v = fftconv(v, k, bias)
    y = (v * x0)
    if trace is not None:
        trace[f"l{layer}.filter"] = k
        trace[f"l{layer}.gated"] = y                          # [B, 256, L] channel-major
    return _lin(y.transpose(1, 2), sd, p + "out_proj", dt)


def hyena_mlp(x, sd, layer, dt):
    """fc2(gelu_tanh(fc1(x)))  (HyenaDNA `HyenaMlp`), SURVEY.md section 8(a) row 9."""
    p = f"{BB}layers.{layer}.mlp."
    return _lin(F.gelu(_lin(x, sd, p + "fc1", dt), approximate="tanh"), sd, p + "fc2", dt)


def backbone_forward(ids, sd, dt=torch.float32, trace=None):
    """HyenaDNA backbone: ids int64 [B, L] -> last_hidden_state [B, L, 256]."""
    h = F.embedding(ids, sd[BB + "embeddings.word_embeddings.weight"].to(dt))
    if trace is not None:
        trace["embed"] = h
    for i in range(N_LAYER):
        p = f"{BB}layers.{i}."
        r = hyena_operator(_ln(h, sd, p + "norm1", dt), sd, i, dt, trace) + h
        if trace is not None:
            trace[f"l{i}.mixer_out"] = r
        h = hyena_mlp(_ln(r, sd, p + "norm2", dt), sd, i, dt) + r
        if trace is not None:
            trace[f"l{i}.out"] = h
    h = _ln(h, sd, BB + "ln_f", dt)
    if trace is not None:
        trace["ln_f"] = h
    return h


# ----------------------------------------------------------------------------- head
def head_forward(hidden, sd, dt=torch.float32, trace=None):
    """`BinarySequenceClassifier.forward(hidden, None)` with pooling_type='attention' (hyena.py:117-146).

    scores = Linear(256,1)(GELU_erf(Linear(256,256)(h))); a = softmax over dim=1 (ALL positions, pads
    included); pooled = sum_L a*h; classifier = Linear(256,512) GELU Linear(512,512) GELU
    ResidualBlock(512) (hyena.py:160-180); logits = Linear(512,2).  Dropouts are identity in eval.
    """
    s = _lin(F.gelu(_lin(hidden, sd, HD + "attention.0", dt)), sd, HD + "attention.2", dt)   # [B, L, 1]
    a = torch.softmax(s, dim=1)
    pooled = (hidden * a).sum(dim=1)
    if trace is not None:
        trace["scores"] = s[..., 0]
        trace["attn_weights"] = a
        trace["pooled"] = pooled
    x = F.gelu(_lin(pooled, sd, HD + "classifier.0", dt))
    x = F.gelu(_lin(x, sd, HD + "classifier.3", dt))
    r = _lin(F.gelu(_lin(x, sd, HD + "classifier.6.layers.0", dt)), sd, HD + "classifier.6.layers.3", dt)
    x = r + x
    return _lin(x, sd, HD + "output_layer", dt)


def forward(ids, sd, dt=torch.float32, trace=None):
    """`HyenaDna.forward(input_ids, input_quals=None)` (hyena.py:244-256): ids [B, L] -> logits [B, 2]."""
    ids = torch.as_tensor(ids, dtype=torch.int64)
    with torch.no_grad():
        hidden = backbone_forward(ids, sd, dt, trace)
        return head_forward(hidden, sd, dt, trace)


# ----------------------------------------------------------------------------- synthetic reads
def synthetic_batch(batch_index: int, batch: int, bases: int, seed: int = 1234):
    """Seeded synthetic reads as defined in SURVEY.md section 8(d): A/C/G/T uniform (ids 7..10), N (11)
    with p=0.001, then [SEP]=1 appended.  Returns (ids uint8 [B, bases+1], names list[str])."""
    import numpy as np

    rng = np.random.default_rng(seed + batch_index)
    ids = rng.integers(7, 11, size=(batch, bases), dtype=np.uint8)
    ids[rng.random((batch, bases)) < 0.001] = 11
    ids = np.concatenate([ids, np.ones((batch, 1), np.uint8)], axis=1)
    names = [f"synthetic_{batch_index * batch + i:08d}" for i in range(batch)]
    return ids, names
