"""BAM input of the predict path, mirroring /root/reference/chimeralm/data/bam.py.

`is_chimeric` (:21-23), `parse_bam_file` (:26-38), `BamDataModule.setup("predict")` (:148-174) and
`predict_dataloader` (:287-299).  The reference reads BAM through pysam and materialises HF `datasets` Arrow caches;
here a small BGZF/BAM decoder (stdlib gzip + struct) streams records, which is a behavioural superset (same reads,
same order, same batches).  A native C++ feeder is the next step of SURVEY.md section 8(f)-2.
"""
from __future__ import annotations

import gzip
import struct
from collections.abc import Iterator
from pathlib import Path

import numpy as np

from .tokenizer import DataCollator, tokenize_and_align_labels_and_quals_ids

_SEQ_LUT = np.frombuffer(b"=ACMGRSVTWYHKDBN", dtype=np.uint8)
_FLAG_UNMAPPED, _FLAG_SECONDARY, _FLAG_SUPPLEMENTARY = 0x4, 0x100, 0x800
_AUX_SIZE = {"A": 1, "c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}


def _has_tag(aux: memoryview | bytes, want: bytes) -> bool:
    p, n = 0, len(aux)
    while p + 3 <= n:
        tag, typ = bytes(aux[p: p + 2]), chr(aux[p + 2])
        if tag == want:
            return True
        p += 3
        if typ in _AUX_SIZE:
            p += _AUX_SIZE[typ]
        elif typ in "ZH":
            while aux[p] != 0:
                p += 1
            p += 1
        elif typ == "B":
            p += 5 + struct.unpack_from("<i", aux, p + 1)[0] * _AUX_SIZE[chr(aux[p])]
        else:
            raise ValueError(f"corrupt BAM auxiliary field type {typ!r}")
    return False


def iter_bam(path: str | Path) -> Iterator[tuple[int, str, str, bytes]]:
    """(flag, query_name, sequence, aux bytes) of every alignment record, in file order."""
    with gzip.open(path, "rb") as fh:          # BGZF is a series of gzip members
        if fh.read(4) != b"BAM\x01":
            raise ValueError(f"{path}: not a BAM file")
        fh.read(struct.unpack("<i", fh.read(4))[0])
        for _ in range(struct.unpack("<i", fh.read(4))[0]):
            fh.read(struct.unpack("<i", fh.read(4))[0] + 4)
        while True:
            head = fh.read(4)
            if len(head) < 4:
                return
            rec = fh.read(struct.unpack("<i", head)[0])
            l_name, n_cigar, flag, l_seq = rec[8], *struct.unpack_from("<HHi", rec, 12)
            p = 32
            name = rec[p: p + l_name - 1].decode()
            p += l_name + 4 * n_cigar
            packed = np.frombuffer(rec, dtype=np.uint8, count=(l_seq + 1) // 2, offset=p)
            codes = np.empty(2 * len(packed), np.uint8)
            codes[0::2], codes[1::2] = packed >> 4, packed & 15
            seq = _SEQ_LUT[codes[:l_seq]].tobytes().decode()
            p += (l_seq + 1) // 2 + l_seq
            yield flag, name, seq, rec[p:]


def is_chimeric(flag: int, aux: bytes) -> bool:
    """mapped, primary (neither secondary nor supplementary) and carrying an SA tag."""
    return not (flag & _FLAG_UNMAPPED) and not (flag & _FLAG_SECONDARY) and not (flag & _FLAG_SUPPLEMENTARY) \
        and _has_tag(aux, b"SA")


def parse_bam_file(file_path: str | Path) -> Iterator[dict]:
    for flag, name, seq, aux in iter_bam(file_path):
        if is_chimeric(flag, aux):
            yield {"id": name, "seq": seq}


class BamDataModule:
    """Predict-stage subset of the reference's LightningDataModule (same constructor arguments)."""

    def __init__(self, tokenizer, train_data_path=None, batch_size: int = 12, val_data_path=None, test_data_path=None,
                 predict_data_path=None, num_workers: int = 0, max_train_samples=None, max_val_samples=None,
                 max_test_samples=None, max_predict_samples: int | None = None, *, pin_memory: bool = False):
        self.tokenizer, self.batch_size, self.predict_data_path = tokenizer, batch_size, predict_data_path
        self.max_predict_samples, self.pin_memory, self.num_workers = max_predict_samples, pin_memory, num_workers
        self.batch_size_per_device = batch_size
        self.data_collator = DataCollator(tokenizer)
        self.data_predict = False
        self.world_size, self.rank = 1, 0

    @property
    def num_classes(self) -> int:
        return 2

    def setup(self, stage: str | None = None, world_size: int = 1, rank: int = 0) -> None:
        if stage != "predict":
            raise NotImplementedError("the MI355X engine covers the predict stage only")
        if not self.predict_data_path:
            raise ValueError("Predict data path is required for prediction stage.")
        if self.batch_size % world_size != 0:
            raise RuntimeError(f"Batch size ({self.batch_size}) is not divisible by the number of devices ({world_size}).")
        self.world_size, self.rank = world_size, rank
        self.batch_size_per_device = self.batch_size // world_size
        if not Path(self.predict_data_path).exists():
            raise FileNotFoundError(f"File not found: {self.predict_data_path}")
        self.data_predict = True               # set up; the file itself is streamed by predict_dataloader

    def predict_dataloader(self) -> Iterator[dict]:
        """Batches in file order; with world_size > 1 rank r takes samples r, r+G, r+2G, ... (the non-shuffling
        distributed sampler Lightning installs -- SURVEY.md Appendix B).  The reference tokenises the whole file into an Arrow
        cache before the first batch (bam.py:153-172); here records are parsed, tokenised and collated as they are consumed,
        so memory holds one batch whatever the size of the BAM (the native feeder, the default of `predict`, does the same
        in C++)."""
        assert self.data_predict, "call setup('predict') first"
        max_length = self.tokenizer.max_len_single_sentence
        batch: list[dict] = []
        for i, rec in enumerate(parse_bam_file(self.predict_data_path)):
            if self.max_predict_samples is not None and i >= self.max_predict_samples:
                break
            if self.world_size > 1 and i % self.world_size != self.rank:
                continue
            batch.append(tokenize_and_align_labels_and_quals_ids(rec, self.tokenizer, max_length))
            if len(batch) == self.batch_size_per_device:
                yield self.data_collator.torch_call(batch)
                batch = []
        if batch:
            yield self.data_collator.torch_call(batch)
