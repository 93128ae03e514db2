"""GPU (MI355X), DORMANT until the released weights are on the box: the only known answers of the production model that the
reference holds -- the outputs printed in /root/reference/notebooks/attention.ipynb (CPU, torch 2.5.1, `yangliz5/chimeralm`):

    cell 8  (:280)      "ATCGCGTG" -> {'Biological': '0.973', 'Chimeric Artifact': '0.027'}
    cell 8  (:296-297)  its eight pooling weights (the [SEP] position dropped, `weights[:-1]`)
    cell 10 (:507-509)  "AAAAAAAA" -> (0.968, 0.032),  "TTTTTTTT" -> (0.965, 0.035)

They are the only reference-held evidence that can pin the backbone (the
HyenaDNA remote code and the fine-tuned weights are off-box and the reference has no fixture for them).  The test is skipped
unless `model.safetensors` of `yangliz5/chimeralm` is found: $CLM_WEIGHTS (file or directory), or the local Hugging Face cache.
When it is, the SAME checks run on the CPU oracle (fp32) and on the HIP engine (fp32 and fp16c modes):
softmax(logits) to 3 decimals, pooling weights to 1e-6 (oracle, fp32 engine) / 1e-4 (fp16c), and every tensor of the
checkpoint is consumed by `load_reference_checkpoint` (no unexpected, no missing key).
"""
from __future__ import annotations

import os
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

KNOWN = {"ATCGCGTG": (0.973, 0.027), "AAAAAAAA": (0.968, 0.032), "TTTTTTTT": (0.965, 0.035)}
POOLING_ATCGCGTG = np.array([0.10990726, 0.06500999, 0.14795601, 0.11923233, 0.11386099, 0.10395291, 0.12930304, 0.07828572],
                            np.float32)


def _find_weights() -> Path | None:
    env = os.environ.get("CLM_WEIGHTS")
    if env:
        p = Path(env)
        p = p / "model.safetensors" if p.is_dir() else p
        return p if p.exists() else None
    try:
        from huggingface_hub import hf_hub_download

        return Path(hf_hub_download("yangliz5/chimeralm", "model.safetensors", local_files_only=True))
    except Exception:  # noqa: BLE001 - not cached / hub library absent: stay dormant
        return None


WEIGHTS = _find_weights()
needs_weights = pytest.mark.skipif(WEIGHTS is None, reason="released weights yangliz5/chimeralm (model.safetensors) not on this "
                                                           "box: set CLM_WEIGHTS or populate the HF cache")


def _ids(seq: str) -> torch.Tensor:
    from chimeralm_amd import tokenizer as T

    tok = T.load_tokenizer_from_hyena_model("hyenadna-small-32k-seqlen")
    ids = tok(seq, truncation=True, max_length=32768)["input_ids"]
    assert len(ids) == len(seq) + 1 and ids[-1] == 1          # exactly one trailing [SEP] (attention.ipynb:167,296)
    return torch.tensor([ids], dtype=torch.int64)


@needs_weights
def test_checkpoint_keys_are_exactly_the_engines():
    from safetensors import safe_open

    from chimeralm_amd import lm

    model = lm.ChimeraLM.new(precision="fp32")
    own = set(model.state_dict())
    with safe_open(str(WEIGHTS), framework="pt") as f:
        have = set(f.keys())
    aliases = {k for k in own if ".implicit_filter.3.freq" in k or ".implicit_filter.5.freq" in k}   # shared sine module
    assert not (have - own), f"checkpoint tensors the engine would ignore: {sorted(have - own)[:5]}"
    assert not (own - have - aliases), f"engine tensors the checkpoint lacks: {sorted(own - have - aliases)[:5]}"
    model.load_reference_checkpoint(WEIGHTS)


@needs_weights
def test_notebook_known_answers_oracle():
    """The CPU restatement itself against the notebook: this is what pins oracle/hyena_oracle.py's backbone."""
    from safetensors.torch import load_file

    from oracle import hyena_oracle as ho

    sd = {k: v.float() for k, v in load_file(str(WEIGHTS)).items()}
    for k in list(sd):                                         # restore the aliases safetensors dropped
        if k.endswith("implicit_filter.1.freq"):
            sd.setdefault(k.replace(".1.freq", ".3.freq"), sd[k]), sd.setdefault(k.replace(".1.freq", ".5.freq"), sd[k])
    for seq, want in KNOWN.items():
        trace = {}
        p = torch.softmax(ho.forward(_ids(seq), sd, trace=trace), dim=-1)[0].numpy()
        assert [f"{v:.3f}" for v in p] == [f"{v:.3f}" for v in want], f"oracle {seq}: {p}"
        if seq == "ATCGCGTG":
            w = trace["attn_weights"][0, :-1, 0].numpy()
            assert np.abs(w - POOLING_ATCGCGTG).max() <= 1e-6, w


@needs_weights
@pytest.mark.parametrize("prec,wtol", [("fp32", 1e-6), ("fp16c", 1e-4)])
def test_notebook_known_answers_engine(built_lib, prec, wtol):
    from chimeralm_amd import lm

    model = lm.ChimeraLM.from_pretrained(str(WEIGHTS), save_attention=True, precision=prec)
    model.eval()
    for seq, want in KNOWN.items():
        logits = model(_ids(seq).cuda(), None)
        p = torch.softmax(logits, dim=-1)[0].cpu().numpy()
        assert [f"{v:.3f}" for v in p] == [f"{v:.3f}" for v in want], f"{prec} {seq}: {p}"
        if seq == "ATCGCGTG":
            w = model.net.head.attention_weights.squeeze(0).squeeze(-1)[:-1].numpy()
            assert np.abs(w - POOLING_ATCGCGTG).max() <= wtol, w
