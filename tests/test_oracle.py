"""CPU: the oracle against the golden fixtures made from the reference's own code (tests/golden/make_golden.py)
and against the reference's known-answer tests."""
from __future__ import annotations

import json

import numpy as np
import pytest
import torch

from oracle import data_oracle as do
from oracle import hyena_oracle as ho


def test_head_matches_reference_head(golden_dir):
    """oracle.head_forward == reference BinarySequenceClassifier (hyena.py:79-146) on seeded weights."""
    g = np.load(golden_dir / "head_golden.npz")
    for seed in (0, 1):
        _, head_scale, b, l = g[f"meta_{seed}"]
        sd = ho.make_state_dict(int(seed), head_scale=float(head_scale))
        rng = np.random.default_rng(1000 + seed)
        hidden = torch.from_numpy(rng.standard_normal((int(b), int(l), 256))).float()
        trace = {}
        with torch.no_grad():
            logits = ho.head_forward(hidden, sd, trace=trace)
        np.testing.assert_allclose(logits.numpy(), g[f"logits_{seed}"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(trace["attn_weights"].numpy(), g[f"attn_{seed}"], rtol=1e-5, atol=1e-8)


def test_tokenizer_known_answers():
    # /root/reference/tests/test_tokenzier.py:11-12 (in-tree CharacterTokenizer: [CLS] ... [SEP])
    assert do.tokenize("ATCG", 100, add_cls=True) == [0, 7, 10, 8, 9, 1]
    # production (HF remote) tokenizer: one trailing [SEP] only (notebooks/attention.ipynb:167,296)
    assert do.tokenize("ATCG", 100) == [7, 10, 8, 9, 1]
    assert do.tokenize("ATCGX", 4) == [7, 10, 8, 1]           # truncation keeps room for [SEP]


def test_collate_matches_reference(golden_dir):
    gold = json.loads((golden_dir / "collate_golden.json").read_text())
    for case in gold["cases"]:
        feats = []
        for (name, seq), ids_ref, id_ref in zip(gold["reads"], case["per_read_input_ids"], case["per_read_id"]):
            ids = do.tokenize(seq, case["max_length"], add_cls=True)
            assert ids == ids_ref
            row = do.pack_read_name(name)
            assert row == id_ref
            feats.append({"input_ids": ids, "id": row, "labels": -1})
        batch = do.collate(feats, padding_side=case["padding_side"])
        assert batch["input_ids"].tolist() == case["input_ids"]
        assert batch["id"].tolist() == case["id_int8"]
        assert batch["labels"].tolist() == case["labels"]
    # names of 128+ characters: the reference collator raises (tokenizer.py:168); the packing itself is pinned
    for ov in gold["overflow"]:
        assert ov["collator_error"] and "int8" in ov["collator_error"]
        assert do.pack_read_name(("x" if ov["name_len"] == 128 else "y") * ov["name_len"]) == ov["per_read_id"]


def test_resume_read_name_matches_reference(golden_dir):
    for row in json.loads((golden_dir / "readname_golden.json").read_text()):
        if isinstance(row["resumed"], dict):
            with pytest.raises(ValueError):
                do.resume_read_name(row["row_int8"])
        else:
            assert do.resume_read_name(row["row_int8"]) == row["resumed"]


def test_prediction_lines_format():
    logits = np.array([[0.2, -1.0], [-3.0, 0.5], [0.0, 0.0]])
    ids = np.array([do.pack_read_name("r1"), do.pack_read_name("r2"), [0] * 256], dtype=np.int64).astype(np.int8)
    assert do.prediction_lines(logits, ids) == ["r1\t0\n", "r2\t1\n", "error_read_2\t0\n"]


def test_bam_fixture_selection(golden_dir):
    """100 primary records, all chimeric, 524..137,138 bp, 11 reads > 32,767 bp, names <= 73."""
    recs = list(do.chimeric_reads(golden_dir / "test_chimric_reads.bam"))
    lens = [len(r["seq"]) for r in recs]
    assert len(recs) == 100 and min(lens) == 524 and max(lens) == 137138
    assert sum(n > 32767 for n in lens) == 11 and max(len(r["id"]) for r in recs) == 73
    assert set("".join(r["seq"] for r in recs[:5])) <= set("ACGTN")


def test_parquet_batch_shape(golden_dir):
    """/root/reference/tests/test_data_module.py:55-73: model_max_length=100, left padding -> (12, 98)."""
    import pyarrow.parquet as pq

    tab = pq.read_table(golden_dir / "tests.parquet").to_pydict()
    seqs = tab["seq"][:12]
    feats = [{"input_ids": do.tokenize(s, 98, add_cls=True)} for s in seqs]
    assert do.collate(feats, padding_side="left")["input_ids"].shape == (12, 98)


def test_fftconv_is_causal_convolution():
    g = torch.Generator().manual_seed(0)
    u = torch.randn(2, 5, 97, generator=g, dtype=torch.float64)
    k = torch.randn(5, 97, generator=g, dtype=torch.float64)
    d = torch.randn(5, generator=g, dtype=torch.float64)
    assert (ho.fftconv(u, k, d) - ho.direct_causal_conv(u, k, d)).abs().max() < 1e-10


def test_backbone_shapes_param_count_and_padding_dependence():
    sd = ho.make_state_dict(0)
    n_backbone = sum(v.numel() for k, v in sd.items() if k.startswith(ho.BB) and not k.endswith("pos_emb.t")
                     and "deltas" not in k and ".3.freq" not in k and ".5.freq" not in k)
    n_head = sum(v.numel() for k, v in sd.items() if k.startswith(ho.HD))
    assert n_backbone == 3_932_968 and n_head == 986_627
    ids, _ = ho.synthetic_batch(0, 2, 64)
    out = ho.forward(torch.from_numpy(ids).long(), sd)
    assert out.shape == (2, 2) and torch.isfinite(out).all()
    # fact 5 of SURVEY.md: pads are not masked, so left padding changes a read's logits
    padded = np.concatenate([np.full((2, 7), 4, np.uint8), ids], axis=1)
    out_p = ho.forward(torch.from_numpy(padded).long(), sd)
    assert (out - out_p).abs().max() > 1e-6
    # fp32 vs fp64 evaluation agree to fp32 rounding
    out64 = ho.forward(torch.from_numpy(ids).long(), sd, dt=torch.float64)
    assert (out.double() - out64).abs().max() < 1e-4


def test_synthetic_batch_spec():
    ids, names = ho.synthetic_batch(3, 4, 1000)
    assert ids.shape == (4, 1001) and ids.dtype == np.uint8 and (ids[:, -1] == 1).all()
    assert set(np.unique(ids[:, :-1])) <= {7, 8, 9, 10, 11} and names[0] == "synthetic_00000012"
    ids2, _ = ho.synthetic_batch(3, 4, 1000)
    assert (ids == ids2).all()


def test_transformer_oracle_matches_reference_module(golden_dir):
    """SequenceCNNTransformer: oracle restatement vs the reference module's own outputs
    (tests/golden/make_golden.py::transformer_golden) -- production configuration, padded batch, small configuration."""
    from oracle import transformer_oracle as to

    g = np.load(golden_dir / "transformer_golden.npz")
    small = to.Config(max_len=512, d_model=64, num_encoder_layers=2, nhead=4, dim_feedforward=128)
    for name, cfg in (("prod", to.PRODUCTION), ("prod_pad", to.PRODUCTION), ("small", small)):
        seed, B, L, pads = (int(v) for v in g[f"{name}_meta"])
        sd = to.make_state_dict(seed, cfg, scale=3.0)
        ids = torch.from_numpy(to.synthetic_ids(100 + seed, B, L, pads))
        trace = {}
        logits = to.forward(ids, sd, cfg, trace=trace).numpy()
        assert np.abs(logits - g[f"{name}_logits"]).max() < 5e-5
        assert np.abs(trace["embedded"][:, ::37, ::5].numpy() - g[f"{name}_embedded_sample"]).max() < 1e-5
        last = trace[f"layer{cfg.num_encoder_layers - 1}"]
        assert np.abs(last[:, ::37, ::5].numpy() - g[f"{name}_encoded_sample"]).max() < 5e-5
        assert trace["embedded"].shape[1] == L // 8                      # three floor-halvings
        assert abs(float(trace["pool_weights"].sum(dim=1).mean()) - 1.0) < 1e-5
    ref64 = to.forward(ids, sd, cfg, dtype=torch.float64).numpy()       # fp64 referee close to the fp32 run
    assert np.abs(ref64 - logits).max() < 1e-4
