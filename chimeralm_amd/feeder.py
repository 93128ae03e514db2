"""ctypes front of the native BAM feeder (include/chimeralm_feed.h, csrc/bam_feeder.cpp).

Replaces, for `predict`, the reference's Python data path -- `parse_bam_file` (/root/reference/chimeralm/data/bam.py:21-38),
`tokenize_and_align_labels_and_quals_ids` (data/tokenizer.py:85-114), `DataCollator.torch_call` (:136-187) and the
per-device batching of `BamDataModule` (data/bam.py:142-174,287-299) -- with a C++ decoder thread that fills a ring of
page-locked host slots.  `BamFeeder` yields the same batches in the same order as `chimeralm_amd.bam.BamDataModule`.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from pathlib import Path

import numpy as np

from . import _native as N


class FeederError(RuntimeError):
    pass


@dataclass
class FeedBatch:
    slot: int
    n_reads: int
    n_tokens: int
    row_stride: int
    ids_ptr: int            # host address of uint8 [n_reads, row_stride] (page-locked when the feeder is pinned)
    ids: np.ndarray         # zero-copy view of the slot -- valid until `release`
    names: np.ndarray       # int8 [n_reads, 256], a COPY (outlives the slot)
    first_index: int


class BamFeeder:
    def __init__(self, bam_path: str | Path, batch_size: int = 12, *, max_tokens: int = 32769, slots: int = 4, rank: int = 0,
                 world: int = 1, pad_left: bool = True, pinned: bool = True, max_reads: int | None = None,
                 inflate_threads: int = 0):
        self._lib = N.load()
        cfg = N.ClmFeederConfig()
        self._lib.clm_feeder_default_config(C.byref(cfg))
        cfg.batch_size, cfg.max_tokens, cfg.slots, cfg.rank, cfg.world = batch_size, max_tokens, slots, rank, world
        cfg.pad_left, cfg.pinned = int(pad_left), int(pinned)
        cfg.max_reads = -1 if max_reads is None else int(max_reads)
        cfg.inflate_threads = int(inflate_threads)      # 0: chosen by the library from the core count and `world`
        self._h = C.c_void_p()
        rc = self._lib.clm_feeder_open(str(bam_path).encode(), C.byref(cfg), C.byref(self._h))
        if rc != 0:
            self._h = None
            raise FeederError(self._lib.clm_feeder_last_error(None).decode())
        self.batch_size = batch_size

    def next(self) -> FeedBatch | None:
        b = N.ClmFeedBatch()
        rc = self._lib.clm_feeder_next(self._h, C.byref(b))
        if rc < 0:
            raise FeederError(self._lib.clm_feeder_last_error(self._h).decode())
        if rc == 0:
            return None
        ids = np.ctypeslib.as_array(C.cast(b.ids, C.POINTER(C.c_uint8)), shape=(b.n_reads, b.row_stride))
        names = np.ctypeslib.as_array(C.cast(b.names, C.POINTER(C.c_int8)), shape=(b.n_reads, 256)).copy()
        return FeedBatch(b.slot, b.n_reads, b.n_tokens, b.row_stride, int(b.ids), ids, names, b.first_index)

    def release(self, batch: FeedBatch):
        if self._lib.clm_feeder_release(self._h, batch.slot) != 0:
            raise FeederError(self._lib.clm_feeder_last_error(self._h).decode())

    def stats(self) -> dict[str, int]:
        v = [C.c_int64() for _ in range(4)]
        self._lib.clm_feeder_stats(self._h, *[C.byref(x) for x in v])
        return dict(zip(("records", "selected", "delivered", "truncated_bases"), (x.value for x in v)))

    def __iter__(self):
        """Batches as numpy copies, slot released immediately (tests, small tools)."""
        while (b := self.next()) is not None:
            ids = b.ids.copy()
            self.release(b)
            yield ids, b.names

    def close(self):
        if getattr(self, "_h", None):
            self._lib.clm_feeder_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
