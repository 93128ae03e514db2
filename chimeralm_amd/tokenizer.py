"""Token stream of the predict path, mirroring /root/reference/chimeralm/data/tokenizer.py.

* vocabulary                                           tokenizer.py:230-239
* `load_tokenizer_from_hyena_model(model_name)`        tokenizer.py:36-55  (HF remote tokenizer in the reference:
  characters -> ids, ONE trailing [SEP], no [CLS], left padding, model_max_length 32770 -- SURVEY.md section 8(a))
* `tokenize_and_align_labels_and_quals_ids`            tokenizer.py:85-114 (id row = [len] + code points, 256 wide)
* `DataCollator.torch_call`                            tokenizer.py:136-187 (pad to longest with [PAD]=4)

`CharTokenizer(add_cls=True, padding_side="right")` reproduces the in-tree `CharacterTokenizer` (tokenizer.py:190-327),
which is what the reference's own tokenizer tests pin (tests/test_tokenzier.py:11-12).
"""
from __future__ import annotations

import numpy as np
import torch

VOCAB = {"[CLS]": 0, "[SEP]": 1, "[BOS]": 2, "[MASK]": 3, "[PAD]": 4, "[RESERVED]": 5, "[UNK]": 6,
         "A": 7, "C": 8, "G": 9, "T": 10, "N": 11}
CLS_ID, SEP_ID, PAD_ID, UNK_ID = 0, 1, 4, 6
MAX_ID_LENGTH = 256
MODEL_SEQ_INPUT, MODEL_LABEL_INPUT, ID_FEATURE, SEQ_FEATURE = "input_ids", "labels", "id", "seq"

_LUT = np.full(256, UNK_ID, dtype=np.uint8)
for _ch in "ACGTN":
    _LUT[ord(_ch)] = VOCAB[_ch]

_MAX_LENGTHS = {"hyenadna-tiny-1k-seqlen": 1024, "hyenadna-small-32k-seqlen": 32768,
                "hyenadna-medium-160k-seqlen": 160000, "hyenadna-medium-450k-seqlen": 450000,
                "hyenadna-large-1m-seqlen": 1_000_000}


class CharTokenizer:
    """Single-nucleotide tokenizer.  `model_max_length` counts special tokens, as in transformers."""

    def __init__(self, model_max_length: int = 32770, padding_side: str = "left", *, add_cls: bool = False):
        self.model_max_length = model_max_length
        self.padding_side = padding_side
        self.add_cls = add_cls
        self.pad_token_id, self.sep_token_id, self.cls_token_id, self.unk_token_id = PAD_ID, SEP_ID, CLS_ID, UNK_ID

    @property
    def vocab_size(self) -> int:
        return len(VOCAB)

    @property
    def max_len_single_sentence(self) -> int:
        return self.model_max_length - (2 if self.add_cls else 1)

    def encode_array(self, seq: str, max_length: int | None = None) -> np.ndarray:
        """uint8 ids of one read incl. special tokens, truncated to `max_length` tokens in total."""
        n_special = 2 if self.add_cls else 1
        raw = np.frombuffer(seq.encode("latin-1", "replace"), dtype=np.uint8)
        if max_length is not None:
            raw = raw[: max(0, max_length - n_special)]
        body = _LUT[raw]
        parts = ([np.array([CLS_ID], np.uint8)] if self.add_cls else []) + [body, np.array([SEP_ID], np.uint8)]
        return np.concatenate(parts)

    def __call__(self, seq: str, truncation: bool = True, max_length: int | None = None, padding=True) -> dict:
        ml = (max_length if max_length is not None else self.model_max_length) if truncation else None
        return {MODEL_SEQ_INPUT: self.encode_array(seq, ml).tolist()}

    def encode(self, seq: str, truncation: bool = False, max_length: int | None = None) -> list[int]:
        return self(seq, truncation=truncation, max_length=max_length)[MODEL_SEQ_INPUT]


def load_tokenizer_from_hyena_model(model_name: str) -> CharTokenizer:
    if model_name not in _MAX_LENGTHS:
        raise ValueError(f"Model name {model_name} not found in available models.")
    # the HF tokenizer config of the -hf repos carries model_max_length = max_length + 2
    return CharTokenizer(model_max_length=_MAX_LENGTHS[model_name] + 2, padding_side="left", add_cls=False)


def pack_read_name(name: str, max_id_length: int = MAX_ID_LENGTH) -> list[int]:
    row = [len(name)] + [ord(c) for c in name]
    return row[:max_id_length] if len(row) > max_id_length else row + [0] * (max_id_length - len(row))


def tokenize_and_align_labels_and_quals_ids(data: dict, tokenizer: CharTokenizer, max_length: int, *,
                                            seq_feature: str = SEQ_FEATURE, id_feature: str = ID_FEATURE,
                                            max_id_length: int = MAX_ID_LENGTH) -> dict:
    out = tokenizer(data[seq_feature], truncation=True, max_length=max_length, padding=True)
    out.update({"id": pack_read_name(data[id_feature], max_id_length), MODEL_LABEL_INPUT: -1})
    return out


class DataCollator:
    """Pad to the longest read of the batch on the tokenizer's padding side; `id` -> int8 [B, 256]."""

    def __init__(self, tokenizer: CharTokenizer):
        self.tokenizer = tokenizer

    def torch_call(self, features: list[dict]) -> dict[str, torch.Tensor]:
        lens = [len(f[MODEL_SEQ_INPUT]) for f in features]
        longest = max(lens)
        ids = np.full((len(features), longest), self.tokenizer.pad_token_id, dtype=np.int64)
        for i, f in enumerate(features):
            if self.tokenizer.padding_side == "left":
                ids[i, longest - lens[i]:] = f[MODEL_SEQ_INPUT]
            else:
                ids[i, : lens[i]] = f[MODEL_SEQ_INPUT]
        batch = {MODEL_SEQ_INPUT: torch.from_numpy(ids)}
        if "id" in features[0]:
            rows = np.asarray([f["id"] for f in features], dtype=np.int64)
            batch["id"] = torch.from_numpy((rows & 0xFF).astype(np.uint8).view(np.int8))   # see callbacks.py note
        label_name = "label" if "label" in features[0] else "labels"
        if label_name in features[0]:
            batch[label_name] = torch.tensor([f[label_name] for f in features], dtype=torch.int64)
        return batch

    __call__ = torch_call
