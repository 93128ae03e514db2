"""GPU (MI355X): BASELINE.json's configurations as they are stated, against the oracle.

  C1  `chimeralm predict tests/data BAM --batch-size 12`: the FIRST BATCH OF 12 READS of the reference's test BAM, untruncated
      (the production tokenizer's model_max_length 32770, left padding: /root/reference/chimeralm/data/tokenizer.py:52-55,136-187,
      data/bam.py:148-174), collated by the Python data module AND by the native feeder (byte-identical), logits vs the oracle
      in fp32 mode and in the fp16c throughput mode.  Pad tokens are attended and pooled like the reference does (SURVEY fact 5:
      logits depend on the batch composition), so the batch is compared as a whole.
  C2  synthetic 4k-bp reads (4097 tokens), batch 64: bf16 as the config names it (a reduced-precision mode: its own bound) and
      fp16c at the gate -- size-independent properties on the full batch plus an oracle sample.
"""
from __future__ import annotations

import numpy as np
import pytest
import torch

from oracle import data_oracle as do
from oracle import hyena_oracle as ho

pytestmark = pytest.mark.gpu
GATE = 1e-3


@pytest.fixture(scope="module")
def sd():
    return ho.make_state_dict(0, head_scale=3.0)


def test_c1_first_batch_of_12_untruncated(sd, golden_dir, built_lib, monkeypatch):
    from chimeralm_amd import bam, tokenizer as T
    from chimeralm_amd.engine import Engine
    from chimeralm_amd.feeder import BamFeeder

    path = golden_dir / "test_chimric_reads.bam"
    tok = T.load_tokenizer_from_hyena_model("hyenadna-small-32k-seqlen")
    assert tok.model_max_length == 32770 and tok.padding_side == "left"
    dm = bam.BamDataModule(tokenizer=tok, predict_data_path=path, batch_size=12, max_predict_samples=12)
    dm.setup("predict")
    batch = next(iter(dm.predict_dataloader()))
    ids = batch["input_ids"]
    assert ids.shape[0] == 12 and (ids == 4).any() and (ids[:, -1] == 1).all()          # left-padded, [SEP] last
    lengths = (ids != 4).sum(1)
    assert int(lengths.max()) == ids.shape[1] and int(lengths.min()) < ids.shape[1]       # ragged, longest read unpadded
    # the oracle's own data path gives the same batch (reference functions restated in numpy, pinned by the reference's goldens)
    recs = list(do.chimeric_reads(path))[:12]
    want = do.collate([{"input_ids": do.tokenize(r["seq"], tok.max_len_single_sentence), "id": do.pack_read_name(r["id"]),
                        "labels": -1} for r in recs], padding_side="left")
    assert np.array_equal(want["input_ids"], ids.numpy()) and np.array_equal(want["id"], batch["id"].numpy())
    # ... and so does the native feeder, byte for byte
    with BamFeeder(path, batch_size=12, max_tokens=tok.max_len_single_sentence, max_reads=12, slots=2) as fd:
        fb = fd.next()
        assert fb.n_reads == 12 and fb.n_tokens == ids.shape[1]
        assert np.array_equal(fb.ids[:, :fb.n_tokens], ids.numpy().astype(np.uint8))
        assert np.array_equal(fb.names, batch["id"].numpy())
        fd.release(fb)
    ref = ho.forward(ids, sd).numpy()
    tiles = (ids.shape[1] + 127) // 128
    npad = int(((ids == 4).int().cumprod(1).sum(1) // 128).sum())
    print(f"C1 batch: 12 x {ids.shape[1]} tokens, {npad} of {12 * tiles} tail tiles wholly inside a [PAD] prefix")
    for prec in ("fp32", "fp16c"):
        e = Engine("cuda:0", precision=prec, chunk_reads=64)
        e.load_state_dict(sd)
        got = e.forward(ids.cuda()).cpu().numpy()
        e.close()
        err = np.abs(got - ref).max()
        assert err <= GATE, f"C1 batch of 12, {prec}: max |logit error| {err:.2e}"
        assert do.prediction_lines(got, batch["id"].numpy()) == do.prediction_lines(ref, batch["id"].numpy())
        # round 5: the tiles inside the [PAD] prefixes came from the all-[PAD] table (csrc/pad_prefix.hip) -- against the engine that
        # computes every tile (2e-5 in fp32, the mode's bound in fp16c)
        monkeypatch.setenv("CLM_DEBUG", "no_pad_skip")
        full = Engine("cuda:0", precision=prec, chunk_reads=64)
        monkeypatch.delenv("CLM_DEBUG")
        full.load_state_dict(sd)
        got_full = full.forward(ids.cuda()).cpu().numpy()
        full.close()
        d = np.abs(got - got_full).max()
        print(f"C1 {prec}: |logits - oracle| {err:.2e}; [PAD]-prefix tiles skipped vs computed: {d:.2e}")
        assert d <= (2e-5 if prec == "fp32" else 2e-4) and np.abs(got_full - ref).max() <= GATE


@pytest.mark.parametrize("prec,tol,margin", [("bf16", 6e-2, 2e-1), ("fp16c", GATE, 2 * GATE)])
def test_c2_4k_reads_batch_64(sd, built_lib, prec, tol, margin):
    from chimeralm_amd.engine import Engine

    ids, _ = ho.synthetic_batch(7, 64, 4096, seed=61)
    t = torch.from_numpy(ids).cuda()
    e = Engine("cuda:0", precision=prec, chunk_reads=64)
    e.load_state_dict(sd)
    a = e.forward(t).cpu()
    assert a.shape == (64, 2) and torch.isfinite(a).all()
    assert torch.equal(a, e.forward(t).cpu())                                            # bit-identical run to run
    r = e.forward(torch.flip(t, dims=[0]).contiguous()).cpu()                            # other pair partners in the packed FFT
    assert (torch.flip(r, dims=[0]) - a).abs().max() < tol
    pick = [0, 31, 63]
    solo = e.forward(t[pick].contiguous()).cpu()                                         # reads are independent units
    assert (solo - a[pick]).abs().max() < tol
    ref = ho.forward(torch.from_numpy(ids[pick].astype(np.int64)), sd)
    err = float((a[pick] - ref).abs().max())
    assert err <= tol, f"C2 {prec}: max |logit error| {err:.2e} > {tol}"
    decided = (ref[:, 0] - ref[:, 1]).abs() > margin
    assert torch.equal(a[pick].argmax(1)[decided], ref.argmax(1)[decided])
    e.close()


def test_eval_py_hydra_route(sd, tmp_path, golden_dir, built_lib):
    """`python eval.py ckpt_path=... +data.predict_data_path=...` (reference eval.py:33-101): config
    composition -> datamodule / model / PredictionWriter / Trainer from their `_target_`s -> trainer.predict with ckpt loading.
    The files must be the ones the same model gives through the Python API on the same batches."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    from chimeralm_amd import bam, lm, tokenizer as T

    repo = Path(__file__).resolve().parent.parent
    ckpt = tmp_path / "model.ckpt"
    torch.save({"state_dict": sd}, ckpt)                                     # Lightning checkpoint layout
    out = tmp_path / "run"
    env = {**os.environ, "PYTHONPATH": str(repo)}
    r = subprocess.run([sys.executable, str(repo / "eval.py"), f"ckpt_path={ckpt}",
                        f"+data.predict_data_path={golden_dir / 'test_chimric_reads.bam'}", "data.batch_size=10",
                        "+data.max_predict_samples=20", "model.net.precision=fp32", f"hydra.run.dir={out}"],
                       capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    files = sorted((out / "predicts").glob("*.txt"))
    assert [f.name for f in files] == ["0_0.txt", "0_1.txt"]
    model = lm.ChimeraLM.new(precision="fp32").load_reference_checkpoint(ckpt)
    tok = T.load_tokenizer_from_hyena_model("hyenadna-small-32k-seqlen")
    dm = bam.BamDataModule(tokenizer=tok, predict_data_path=golden_dir / "test_chimric_reads.bam", batch_size=10,
                           max_predict_samples=20)
    dm.setup("predict")
    for f, batch in zip(files, dm.predict_dataloader()):
        logits, _ = model.predict_step({**batch, "input_ids": batch["input_ids"].cuda()}, 0)
        assert f.read_text() == "".join(do.prediction_lines(logits.cpu().numpy(), batch["id"].numpy()))
