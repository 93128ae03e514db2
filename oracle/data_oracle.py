"""CPU restatement of the data side of `chimeralm predict` (test oracle, integer/byte work).

TEST INFRASTRUCTURE -- never imported by the product package `chimeralm_amd`.

Follows, in /root/reference:
  chimeralm/data/tokenizer.py:85-114   tokenize_and_align_labels_and_quals_ids (id packing, labels=-1)
  chimeralm/data/tokenizer.py:136-187  DataCollator.torch_call (pad to longest, id -> int8[B,256])
  chimeralm/data/tokenizer.py:230-239  vocabulary
  chimeralm/data/tokenizer.py:297-306  in-tree CharacterTokenizer special tokens ([CLS] ... [SEP])
  chimeralm/data/bam.py:21-38          is_chimeric / parse_bam_file
  chimeralm/models/callbacks.py:38-63  resume_read_name
  chimeralm/models/callbacks.py:107-142 label = argmax, "{name}\t{label}\n" into {rank}_{batch}.txt
The production tokenizer is HF remote code (tokenizer.py:52-55, not in the tree); evidence in the tree
says it appends one [SEP] and no [CLS] (notebooks/attention.ipynb:167,296) -- `add_cls=False` below.
Pinned by tests/golden/collate_*.json generated from the reference's own functions.
"""
from __future__ import annotations

import gzip
import struct

import numpy as np

VOCAB = {"[CLS]": 0, "[SEP]": 1, "[BOS]": 2, "[MASK]": 3, "[PAD]": 4, "[RESERVED]": 5, "[UNK]": 6,
         "A": 7, "C": 8, "G": 9, "T": 10, "N": 11}
PAD_ID, SEP_ID, CLS_ID, UNK_ID = 4, 1, 0, 6
MAX_ID_LENGTH = 256


def tokenize(seq: str, max_length: int, *, add_cls: bool = False) -> list[int]:
    """Characters -> ids, truncated so that the total (with special tokens) is <= max_length."""
    n_special = 2 if add_cls else 1
    body = [VOCAB.get(ch, UNK_ID) for ch in seq[: max(0, max_length - n_special)]]
    return ([CLS_ID] if add_cls else []) + body + [SEP_ID]


def pack_read_name(name: str) -> list[int]:
    """[len(name)] + code points, cut / zero-padded to 256 entries (tokenizer.py:108-111)."""
    row = [len(name)] + [ord(c) for c in name]
    return row[:MAX_ID_LENGTH] if len(row) > MAX_ID_LENGTH else row + [0] * (MAX_ID_LENGTH - len(row))


def collate(features: list[dict], *, padding_side: str = "left") -> dict[str, np.ndarray]:
    """Pad `input_ids` to the longest in the batch with [PAD]=4; id -> int8 (wraps like torch.int8)."""
    longest = max(len(f["input_ids"]) for f in features)
    ids = np.full((len(features), longest), PAD_ID, np.int64)
    mask = np.zeros((len(features), longest), np.int64)
    for i, f in enumerate(features):
        n = len(f["input_ids"])
        if padding_side == "left":
            ids[i, longest - n:] = f["input_ids"]
            mask[i, longest - n:] = 1
        else:
            ids[i, :n] = f["input_ids"]
            mask[i, :n] = 1
    out = {"input_ids": ids, "attention_mask": mask}
    if "id" in features[0]:
        out["id"] = np.array([f["id"] for f in features], dtype=np.int64).astype(np.int8)
    if "labels" in features[0]:
        out["labels"] = np.array([f["labels"] for f in features], dtype=np.int64)
    return out


def resume_read_name(row) -> str:
    """callbacks.py:38-63.  Raises ValueError for length <= 0 or >= len(row) (e.g. int8 overflow)."""
    row = [int(x) for x in row]
    if not row:
        return ""
    n = row[0]
    if n <= 0 or n >= len(row):
        raise ValueError("Invalid read name data")
    return "".join(chr(b) for b in row[1: 1 + n] if 32 <= b <= 126)


def prediction_lines(logits: np.ndarray, id_rows: np.ndarray) -> list[str]:
    """callbacks.py:107-142: argmax(dim=1) and one "name<TAB>label" line per read."""
    labels = np.argmax(np.asarray(logits), axis=1)
    lines = []
    for i, row in enumerate(id_rows):
        try:
            name = resume_read_name(row) or f"unknown_read_{i}"
        except ValueError:
            name = f"error_read_{i}"
        lines.append(f"{name}\t{int(labels[i])}\n")
    return lines


# ----------------------------------------------------------------------------- BAM (pure python)
_SEQ_CODE = "=ACMGRSVTWYHKDBN"


def iter_bam_records(path):
    """Yield (flag, name, seq, has_SA) for every record of a BAM file (BGZF = concatenated gzip)."""
    with gzip.open(path, "rb") as fh:
        data = fh.read()
    assert data[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<i", data, 4)[0]
    off = 8 + l_text
    n_ref = struct.unpack_from("<i", data, off)[0]
    off += 4
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", data, off)[0]
        off += 4 + l_name + 4
    while off < len(data):
        block_size = struct.unpack_from("<i", data, off)[0]
        rec = data[off + 4: off + 4 + block_size]
        off += 4 + block_size
        _ref, _pos, l_read_name, _mapq, _bin, n_cigar, flag, l_seq = struct.unpack_from("<iiBBHHHi", rec, 0)
        p = 32
        name = rec[p: p + l_read_name - 1].decode()
        p += l_read_name + 4 * n_cigar
        packed = rec[p: p + (l_seq + 1) // 2]
        seq = "".join(_SEQ_CODE[b >> 4] + _SEQ_CODE[b & 15] for b in packed)[:l_seq]
        p += (l_seq + 1) // 2 + l_seq
        has_sa = False
        while p < len(rec):
            tag, typ = rec[p: p + 2], chr(rec[p + 2])
            p += 3
            if tag == b"SA":
                has_sa = True
            if typ in "AcC":
                p += 1
            elif typ in "sS":
                p += 2
            elif typ in "iIf":
                p += 4
            elif typ in "ZH":
                p = rec.index(b"\x00", p) + 1
            elif typ == "B":
                sub, cnt = chr(rec[p]), struct.unpack_from("<i", rec, p + 1)[0]
                p += 5 + cnt * {"c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}[sub]
            else:
                raise ValueError(f"bad aux type {typ}")
        yield flag, name, seq, has_sa


def chimeric_reads(path):
    """bam.py:21-38: keep mapped, primary (not secondary/supplementary) records that carry an SA tag."""
    for flag, name, seq, has_sa in iter_bam_records(path):
        if not (flag & 0x4) and has_sa and not (flag & 0x100) and not (flag & 0x800):
            yield {"id": name, "seq": seq}
