"""Generate the golden fixtures under tests/golden/ from the reference's own code.

Run ONLY in the build container (needs /root/reference):   python tests/golden/make_golden.py
The reference modules are loaded by file path (their package __init__ pulls pysam/lightning, which are
not installed here); nothing of the reference is copied -- the fixtures hold inputs and expected outputs.

Fixtures written:
  head_golden.npz      reference `BinarySequenceClassifier` (hyena.py:6-146) on seeded weights/input:
                       expected logits + pooling weights.  Weights/inputs are regenerated from the seed
                       by oracle.hyena_oracle.make_state_dict, so only outputs are stored.
  collate_golden.json  reference `tokenize_and_align_labels_and_quals_ids` (tokenizer.py:85-114) +
                       `DataCollator.torch_call` (:136-187) with the in-tree `CharacterTokenizer`
                       (left padding) on hand-made reads incl. truncation / long-name cases.
  readname_golden.json reference `resume_read_name` (callbacks.py:38-63) on packed id rows.
  transformer_golden.npz  reference `SequenceCNNTransformer` (models/components/transformer.py:28-104) in eval mode on the
                       seeded weights / ids that oracle.transformer_oracle regenerates: logits (production configuration of
                       configs/model/transformer.yaml and a small one), pooled vector, pooling weights.
  test_chimric_reads.bam, tests.parquet   data files copied from the reference's tests/data/.
"""
from __future__ import annotations

import ast
import importlib.util
import json
import shutil
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
REF = Path("/root/reference")
sys.path.insert(0, str(REPO))

from oracle import hyena_oracle as ho  # noqa: E402


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, REF / rel)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def head_golden():
    ref = _load("ref_hyena", "chimeralm/models/components/hyena.py")
    out = {}
    for seed, head_scale, shape in ((0, 1.0, (2, 100)), (1, 3.0, (3, 257))):
        sd = ho.make_state_dict(seed, head_scale=head_scale)
        head = ref.BinarySequenceClassifier(
            input_dim=256, hidden_dim=512, num_layers=2, dropout=0.1, pooling_type="attention",
            activation="gelu", use_residual=True, save_attention=True,
        ).eval()
        head.load_state_dict({k[len(ho.HD):]: v for k, v in sd.items() if k.startswith(ho.HD)}, strict=True)
        rng = np.random.default_rng(1000 + seed)
        hidden = torch.from_numpy(rng.standard_normal((*shape, 256))).float()
        with torch.no_grad():
            logits = head(hidden, None)
        out[f"logits_{seed}"] = logits.numpy()
        out[f"attn_{seed}"] = head.attention_weights.numpy()
        out[f"meta_{seed}"] = np.array([seed, head_scale, *shape], dtype=np.float64)
    np.savez(HERE / "head_golden.npz", **out)
    print("head_golden.npz", {k: v.shape for k, v in out.items()})


READS = [
    ("read_a", "ACGTNACGT"),
    ("3f1c6a2e-aaaa-bbbb-cccc-0123456789ab;9d8e7f6a-1111-2222-3333-abcdefabcdef", "ACGT" * 10),
    ("x" * 127, "TTTTGGGGCCCCAAAAN"),           # longest name the reference collator accepts
    ("lower_and_iupac", "acgtRYKMacgtACGT"),    # unknown characters -> [UNK]
    ("long_read", "ACGTTGCA" * 8),              # 64 bases, truncated at max_length below
]


def collate_golden():
    tok = _load("ref_tok", "chimeralm/data/tokenizer.py")
    cases = []
    for model_max_length, side in ((50, "left"), (50, "right"), (32770, "left")):
        tokenizer = tok.CharacterTokenizer(model_max_length=model_max_length, padding_side=side)
        max_length = tokenizer.max_len_single_sentence
        feats = [
            dict(tok.tokenize_and_align_labels_and_quals_ids({"id": n, "seq": s}, tokenizer, max_length))
            for n, s in READS
        ]
        batch = tok.DataCollator(tokenizer).torch_call(feats)
        cases.append({
            "model_max_length": model_max_length, "padding_side": side, "max_length": max_length,
            "per_read_input_ids": [f["input_ids"] for f in feats],
            "per_read_id": [f["id"] for f in feats],
            "input_ids": batch["input_ids"].tolist(),
            "batch_keys": sorted(batch.keys()),
            "id_int8": batch["id"].tolist(),
            "labels": batch["labels"].tolist(),
        })
    # names of 128+ characters: torch.tensor(..., dtype=torch.int8) at tokenizer.py:168 refuses the length byte
    tokenizer = tok.CharacterTokenizer(model_max_length=50, padding_side="left")
    overflow = []
    for name in ("x" * 128, "y" * 300):
        feat = dict(tok.tokenize_and_align_labels_and_quals_ids({"id": name, "seq": "GATTACA"}, tokenizer, 48))
        try:
            tok.DataCollator(tokenizer).torch_call([feat])
            err = None
        except RuntimeError as e:
            err = str(e)
        overflow.append({"name_len": len(name), "per_read_id": feat["id"], "collator_error": err})
    (HERE / "collate_golden.json").write_text(json.dumps({"reads": READS, "cases": cases, "overflow": overflow}))
    print("collate_golden.json", [np.array(c["input_ids"]).shape for c in cases])


def readname_golden():
    # callbacks.py imports lightning (absent here); lift the one pure function out of its source.
    src = (REF / "chimeralm/models/callbacks.py").read_text()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "resume_read_name")
    ns = {"torch": torch}
    exec(compile(ast.Module([fn], []), "callbacks.py", "exec"), ns)  # noqa: S102
    resume = ns["resume_read_name"]
    tok = _load("ref_tok", "chimeralm/data/tokenizer.py")
    tokenizer = tok.CharacterTokenizer(model_max_length=50, padding_side="left")
    names = [n for n, _ in READS] + ["", "tab\tname", "n" * 127, "n" * 128, "n" * 255, "é_non_ascii"]
    rows = []
    for n in names:
        row = tok.tokenize_and_align_labels_and_quals_ids({"id": n, "seq": "A"}, tokenizer, 48)["id"]
        row_i8 = torch.tensor(row, dtype=torch.int64).to(torch.int8)   # what DataCollator produces
        try:
            got = resume(row_i8)
        except ValueError as e:
            got = {"error": str(e)}
        rows.append({"name": n, "row_int8": row_i8.tolist(), "resumed": got})
    (HERE / "readname_golden.json").write_text(json.dumps(rows))
    print("readname_golden.json", [(r["name"][:8], r["resumed"] if isinstance(r["resumed"], dict) else r["resumed"][:8]) for r in rows])


def transformer_golden():
    from oracle import transformer_oracle as to

    ref = _load("ref_transformer", "chimeralm/models/components/transformer.py")
    small = to.Config(max_len=512, d_model=64, num_encoder_layers=2, nhead=4, dim_feedforward=128)
    out = {}
    cases = [("prod", to.PRODUCTION, 0, 2, 1000, 0), ("prod_pad", to.PRODUCTION, 1, 3, 777, 40), ("small", small, 2, 4, 301, 9)]
    for name, cfg, seed, B, L, pads in cases:
        sd = to.make_state_dict(seed, cfg, scale=3.0)
        net = ref.SequenceCNNTransformer(vocab_size=cfg.vocab_size, max_len=cfg.max_len, d_model=cfg.d_model,
                                         cnn_kernel_size=cfg.cnn_kernel_size, dropout=0.1,
                                         num_encoder_layers=cfg.num_encoder_layers, nhead=cfg.nhead,
                                         dim_feedforward=cfg.dim_feedforward, number_of_classes=cfg.number_of_classes,
                                         padding_idx=cfg.padding_idx).eval()
        net.load_state_dict(sd, strict=True)
        ids = torch.from_numpy(to.synthetic_ids(100 + seed, B, L, pads))
        with torch.no_grad():
            logits = net(ids)
            # intermediates through the reference's own submodules, for stage-level pins
            x = net.cnn(net.embedding(ids).transpose(1, 2)).transpose(1, 2)
            x = net.norm(net.pos_encoder(x))
            enc = net.transformer_encoder(x)
        out[f"{name}_logits"] = logits.numpy()
        out[f"{name}_embedded_sample"] = x[:, ::37, ::5].numpy()
        out[f"{name}_encoded_sample"] = enc[:, ::37, ::5].numpy()
        out[f"{name}_meta"] = np.array([seed, B, L, pads], dtype=np.int64)
        mine = to.forward(ids, sd, cfg)
        print(f"transformer {name}: logits {logits.numpy().round(4).tolist()}  |oracle - reference| = {(mine - logits).abs().max():.2e}")
    np.savez(HERE / "transformer_golden.npz", **out)


def data_files():
    for f in ("test_chimric_reads.bam", "tests.parquet"):
        shutil.copyfile(REF / "tests/data" / f, HERE / f)
        print("copied", f)


if __name__ == "__main__":
    head_golden()
    collate_golden()
    readname_golden()
    transformer_golden()
    data_files()
