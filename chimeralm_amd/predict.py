"""The predict loop that `lightning.Trainer.predict` runs in the reference
(/root/reference/chimeralm/__main__.py:307-317; call stack in SURVEY.md section 3.1), without Lightning:

    for batch in datamodule.predict_dataloader():      # collated on the host (pad-to-longest, file order)
        H2D copy on a side stream (pinned staging, overlapped with the previous batch's compute)
        logits, labels = model.predict_step(batch)      # MI355X engine
        writer.write_on_batch_end(...)                  # {rank}_{batch}.txt, "name<TAB>label"

Results leave the device one batch behind: the logits of batch i are copied to page-locked host memory right behind their
forward, the forward of batch i+1 is enqueued, and only then does the host wait for copy i and write its file -- the GPU never
idles while Python formats read names (the reference syncs on every batch, callbacks.py:107).

`run_predict_native` is the same loop fed by the native BAM feeder (csrc/bam_feeder.cpp): a C++ thread decodes, selects,
tokenises and collates into a ring of page-locked slots; each batch crosses PCIe as uint8 on the engine's copy stream
(`clm_stage_ids`) while the previous batch is computing, and the slot goes back to the ring once its copy has landed.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch

from .distributed import gather_logits


def _to_device(batch: dict, device: torch.device, stream: torch.cuda.Stream) -> dict:
    out = {}
    with torch.cuda.stream(stream):
        for k, v in batch.items():
            if k == "input_ids":
                # ids fit a byte (vocabulary 12): 8x less PCIe traffic than the reference's int64 batch
                v = v.to(torch.uint8).pin_memory().to(device, non_blocking=True)
            out[k] = v
    return out


class _Deferred:
    """The logits of one batch on their way to the host (async copy + event), and what the writer needs with them."""

    def __init__(self, logits: torch.Tensor, labels, batch: dict, batch_idx: int, gathered: torch.Tensor | None):
        self.host = torch.empty(logits.shape, dtype=logits.dtype, pin_memory=True)   # caching host allocator: cheap after the first
        self.host.copy_(logits, non_blocking=True)
        self.gathered = None
        if gathered is not None:
            self.gathered = torch.empty(gathered.shape, dtype=gathered.dtype, pin_memory=True)
            self.gathered.copy_(gathered, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()
        self.labels, self.batch, self.batch_idx = labels, batch, batch_idx

    def flush(self, writer, trainer, model, on_batch) -> None:
        self.event.synchronize()
        if self.gathered is not None and on_batch is not None:
            on_batch(self.batch_idx, self.gathered)
        writer.write_on_batch_end(trainer, model, (self.host, self.labels), None, self.batch, self.batch_idx, 0)


def run_predict(model, datamodule, writer, device: torch.device, *, rank: int = 0, gather: bool = False,
                on_batch=None) -> int:
    """Returns the number of reads this rank classified."""
    model.eval()
    copy_stream = torch.cuda.Stream(device)
    compute = torch.cuda.current_stream(device)
    trainer = SimpleNamespace(global_rank=rank)
    it = iter(datamodule.predict_dataloader())
    nxt = next(it, None)
    staged = _to_device(nxt, device, copy_stream) if nxt is not None else None
    n_reads, batch_idx = 0, 0
    pending: _Deferred | None = None
    with torch.inference_mode():
        while staged is not None:
            compute.wait_stream(copy_stream)
            cur = staged
            nxt = next(it, None)                              # host collation of batch i+1 ...
            staged = _to_device(nxt, device, copy_stream) if nxt is not None else None   # ... and its H2D overlap
            logits, labels = model.predict_step(cur, batch_idx)
            now = _Deferred(logits, labels, cur, batch_idx, gather_logits(logits) if gather else None)
            if pending is not None:
                pending.flush(writer, trainer, model, on_batch)   # batch i-1: its copy finished while batch i was enqueued
            pending = now
            n_reads += logits.shape[0]
            batch_idx += 1
        if pending is not None:
            pending.flush(writer, trainer, model, on_batch)
    return n_reads


def run_predict_native(model, feeder, writer, device: torch.device, *, rank: int = 0, gather: bool = False,
                       on_batch=None) -> int:
    """Predict loop over a `chimeralm_amd.feeder.BamFeeder`; same files as `run_predict` over `BamDataModule`."""
    from ._native import DT_U8

    model.eval()
    eng = model.net.engine(device)
    trainer = SimpleNamespace(global_rank=rank)
    n_reads, batch_idx = 0, 0
    pending: _Deferred | None = None
    cur = feeder.next()
    staged = eng.stage_host_ids(cur.ids_ptr, DT_U8, cur.row_stride, cur.n_reads, cur.n_tokens) if cur is not None else -1
    with torch.inference_mode():
        while cur is not None:
            nxt = feeder.next()                               # already decoded by the feeder thread, normally
            nxt_staged = (eng.stage_host_ids(nxt.ids_ptr, DT_U8, nxt.row_stride, nxt.n_reads, nxt.n_tokens)
                          if nxt is not None else -1)         # H2D of batch i+1 overlaps the forward of batch i
            logits = eng.forward_staged(staged, cur.n_reads)
            eng.stage_wait(staged)                            # the copy has left the slot ...
            feeder.release(cur)                               # ... which goes back to the decoder
            labels = torch.full((cur.n_reads,), -1, dtype=torch.int64)   # tokenizer.py:113: predict labels are all -1
            batch = {"id": torch.from_numpy(cur.names), "labels": labels}
            now = _Deferred(logits, labels, batch, batch_idx, gather_logits(logits) if gather else None)
            if pending is not None:
                pending.flush(writer, trainer, model, on_batch)
            pending = now
            n_reads += cur.n_reads
            batch_idx += 1
            cur, staged = nxt, nxt_staged
        if pending is not None:
            pending.flush(writer, trainer, model, on_batch)
    return n_reads
