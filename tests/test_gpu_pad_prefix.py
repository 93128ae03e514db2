"""GPU (MI355X): the [PAD]-prefix reuse of left-padded batches (csrc/pad_prefix.hip).

The reference pads a batch on the left to its longest read and masks nothing (chimeralm/data/tokenizer.py:152-159,
models/components/hyena.py:244-256); the backbone is causal, so every position inside a read's pad prefix has the same hidden state
in every read.  The engine computes one all-[PAD] read per weight load and arithmetic, skips the 128-token tiles that lie wholly
inside a prefix and fills what later stages read of them from that table.  What must hold: the logits of a padded batch are the
full computation's (CLM_DEBUG=no_pad_skip: every tile computed) up to the rounding of the transform size the table's convolution
ran in -- 2e-5 in exact fp32, the mode's own noise in the 16-bit modes -- AND the oracle's, at every prefix shape: none, shorter than
a tile, exactly k tiles, all but the last token, a read of nothing but pads; lengths of 128 k + 1 tokens (peeled last token) and
lengths past the one-shot convolution."""
from __future__ import annotations

import numpy as np
import pytest
import torch

from oracle import hyena_oracle as ho

pytestmark = pytest.mark.gpu
GATE = 1e-3
# |skip - full| per mode: exact arithmetic differs only by the rounding of the table's transform size; the 16-bit modes also by
# their own rounding noise on rows that change in the last bit
SKIP_VS_FULL = {"fp32": 2e-5, "fp16x3": 2e-5, "fp16c": 2e-4, "fp16": 1e-3}
VS_ORACLE = {"fp32": 5e-5, "fp16x3": 1e-4, "fp16c": GATE, "fp16": 5e-3}


def _padded_batch(L, prefixes, seed):
    rng = np.random.default_rng(seed)
    ids = rng.integers(7, 11, size=(len(prefixes), L)).astype(np.uint8)
    ids[:, -1] = 1                                              # [SEP]
    for b, p in enumerate(prefixes):
        ids[b, :p] = 4
    return ids


def _engines(prec, sd, monkeypatch, chunk=8):
    from chimeralm_amd.engine import Engine

    skip = Engine("cuda:0", precision=prec, chunk_reads=chunk)
    skip.load_state_dict(sd)
    monkeypatch.setenv("CLM_DEBUG", "no_pad_skip")              # read by clm_create
    full = Engine("cuda:0", precision=prec, chunk_reads=chunk)
    monkeypatch.delenv("CLM_DEBUG")
    full.load_state_dict(sd)
    if prec == "fp16c":                                         # the 16-bit kernels themselves at every length
        skip.set_f16c_min_len(1)
        full.set_f16c_min_len(1)
    return skip, full


@pytest.fixture(scope="module")
def sd():
    return ho.make_state_dict(0, head_scale=3.0)


@pytest.mark.parametrize("prec", ["fp32", "fp16x3", "fp16c", "fp16"])
@pytest.mark.parametrize("L,prefixes", [
    (700, (0, 1, 127, 128, 129, 500, 699)),                    # a partial last tile; prefixes around one tile; all but [SEP]
    (1025, (1024, 0, 256, 300, 1000)),                         # 128 k + 1 tokens: the peeled last token; a read that is ALL pads but [SEP]
    (3000, (2944, 2000, 0, 2943)),                             # a prefix that ends exactly at a tile boundary, and one token before it
])
def test_skipping_prefix_tiles_equals_the_full_computation(built_lib, sd, monkeypatch, prec, L, prefixes):
    ids = _padded_batch(L, prefixes, seed=500 + L)
    skip, full = _engines(prec, sd, monkeypatch)
    t = torch.from_numpy(ids).cuda()
    a, f = skip.forward(t).cpu().numpy(), full.forward(t).cpu().numpy()
    ref = ho.forward(torch.from_numpy(ids.astype(np.int64)), sd).numpy()
    d, e = float(np.abs(a - f).max()), float(np.abs(a - ref).max())
    print(f"{prec} {len(prefixes)} x {L}, prefixes {prefixes}: |skip - full| {d:.2e}, |skip - oracle| {e:.2e}, |full - oracle| {np.abs(f - ref).max():.2e}")
    assert np.isfinite(a).all() and d <= SKIP_VS_FULL[prec] and e <= VS_ORACLE[prec]
    assert np.array_equal(a, skip.forward(t).cpu().numpy())                      # deterministic, table reused
    # the same reads in another order and batch composition (other workgroup ranges, other pair partners): same logits
    perm = list(reversed(range(len(prefixes))))
    a2 = skip.forward(t[perm].contiguous()).cpu().numpy()
    assert np.abs(a2[np.argsort(perm)] - a).max() <= SKIP_VS_FULL[prec]
    # a batch WITHOUT pads through the same engine: bit-identical to the engine that never skips (the list is every tile)
    plain = _padded_batch(L, (0,) * 3, seed=77)
    tp = torch.from_numpy(plain).cuda()
    assert np.array_equal(skip.forward(tp).cpu().numpy(), full.forward(tp).cpu().numpy())
    skip.close(), full.close()


@pytest.mark.parametrize("prec", ["fp32", "fp16c"])
def test_long_reads_and_a_growing_table(built_lib, sd, monkeypatch, prec):
    """Past the one-shot convolution (segmented kernel, 2 segments) with prefixes on both sides of a segment boundary; the table is
    rebuilt when a longer batch arrives and serves the shorter ones after it."""
    skip, full = _engines(prec, sd, monkeypatch, chunk=4)
    for L, prefixes in ((2100, (1500, 0, 640)), (9000, (8200, 100, 4000)), (2100, (1500, 0, 640))):
        ids = _padded_batch(L, prefixes, seed=900 + L)
        t = torch.from_numpy(ids).cuda()
        a, f = skip.forward(t).cpu().numpy(), full.forward(t).cpu().numpy()
        ref = ho.forward(torch.from_numpy(ids.astype(np.int64)), sd).numpy()
        d, e = float(np.abs(a - f).max()), float(np.abs(a - ref).max())
        print(f"{prec} 3 x {L}: |skip - full| {d:.2e}, |skip - oracle| {e:.2e}")
        assert d <= SKIP_VS_FULL[prec] and e <= VS_ORACLE[prec]
    skip.close(), full.close()


@pytest.mark.parametrize("prec,L,prefixes", [
    # 4 segments + the peeled last token (its dot product rides through the segments): a pair that skips three segments, one that
    # skips one, one with an unpadded read (nothing skipped), an unpaired last read
    ("fp16c", 32769, (30000, 28000, 17000, 9000, 0, 25000, 20000)),
    # 3 segments, table built for 4: prefixes either side of a segment boundary (8,320 / 8,319: tile 64 is the tile before the first
    # real one -- its segment is transformed), one pair exactly one tile into the second segment
    ("fp16c", 20000, (19000, 16500, 8320, 8319, 8448, 8400, 12000, 100)),
    ("fp16", 20000, (19000, 16500, 8320, 8319, 8448, 8400, 12000, 100)),
])
def test_prefix_segments_of_long_reads_are_not_transformed(built_lib, sd, monkeypatch, prec, L, prefixes):
    """Round 5, csrc/hyena_conv.hip SegPrefix: in the segmented convolution of the fused 16-bit path a segment wholly inside the [PAD]
    prefix of BOTH reads of a pair is not transformed -- its spectrum comes from the all-[PAD] table (times 1 + i for the packed
    pair), its outputs lie in tail tiles nobody computes, its share of the last token's dot product is the table's partial sum.
    Against the same engine with every segment transformed (CLM_DEBUG=no_seg_skip), the engine that skips nothing, and the oracle."""
    from chimeralm_amd.engine import Engine

    ids = _padded_batch(L, prefixes, seed=1300 + L)
    skip, full = _engines(prec, sd, monkeypatch)
    monkeypatch.setenv("CLM_DEBUG", "no_seg_skip")
    noseg = Engine("cuda:0", precision=prec, chunk_reads=8)
    monkeypatch.delenv("CLM_DEBUG")
    noseg.load_state_dict(sd)
    if prec == "fp16c":
        noseg.set_f16c_min_len(1)
    t = torch.from_numpy(ids).cuda()
    a, n, f = skip.forward(t).cpu().numpy(), noseg.forward(t).cpu().numpy(), full.forward(t).cpu().numpy()
    rows = [0, len(prefixes) - 1]
    ref = ho.forward(torch.from_numpy(ids[rows].astype(np.int64)), sd).numpy()
    d1, d2, e = float(np.abs(a - n).max()), float(np.abs(a - f).max()), float(np.abs(a[rows] - ref).max())
    print(f"{prec} {len(prefixes)} x {L}: |skip - all segments| {d1:.2e}, |skip - full| {d2:.2e}, |skip - oracle| {e:.2e}")
    assert np.isfinite(a).all() and d1 <= SKIP_VS_FULL[prec] and d2 <= SKIP_VS_FULL[prec] and e <= VS_ORACLE[prec]
    assert np.array_equal(a, skip.forward(t).cpu().numpy())                      # deterministic
    # other pair partners: the same reads reversed
    perm = list(reversed(range(len(prefixes))))
    a2 = skip.forward(t[perm].contiguous()).cpu().numpy()
    assert np.abs(a2[np.argsort(perm)] - a).max() <= SKIP_VS_FULL[prec]
    skip.close(), full.close(), noseg.close()


def test_guarded_module_on_a_padded_batch(built_lib, sd):
    """The product path: `HyenaDna(precision="fp16c")` with its guard (both arithmetics of the self-check build their own table) on a
    ragged batch; new weights drop the tables."""
    import warnings

    from chimeralm_amd import lm

    ids = torch.from_numpy(_padded_batch(5000, (4000, 0, 2500, 4864, 1), seed=31).astype(np.int64)).cuda()
    m = lm.ChimeraLM.new(precision="fp16c")
    m.load_state_dict(sd, strict=True)
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        out = m.net(ids).cpu()
    rep = m.net.selfcheck_report
    assert rep["fallback"] is False and rep["max_abs_dlogit"] <= 5e-4
    assert (out - ho.forward(ids.cpu(), sd)).abs().max() <= GATE
    sd2 = ho.make_state_dict(3, head_scale=3.0)
    m.load_state_dict(sd2, strict=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        out2 = m.net(ids).cpu()
    assert (out2 - ho.forward(ids.cpu(), sd2)).abs().max() <= GATE and not torch.allclose(out, out2)
