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

from .distributed import LogitsGather


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
    """The logits of one batch on their way to the host (async copy + event), and what the writer needs with them.

    With a `LogitsGather` the batch also takes part in the all-gather of this round: every rank contributes a
    `[rows, 3]` tensor (logit0, logit1, valid) -- short or empty batches are padded with valid = 0 -- on the gather's own stream,
    behind this forward only; the gathered `[world * rows, 3]` tensor follows to page-locked host memory on that stream."""

    def __init__(self, logits: torch.Tensor | None, labels, batch: dict | None, batch_idx: int,
                 gather: LogitsGather | None = None, rows: int = 0, device: torch.device | None = None):
        self.host, self.event = None, None
        if logits is not None and logits.is_cuda:
            self.host = torch.empty(logits.shape, dtype=logits.dtype, pin_memory=True)   # caching host allocator: cheap after the first
            self.host.copy_(logits, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record()
        elif logits is not None:                               # host tensors: the CPU rehearsal of the multi-rank protocol (tests)
            self.host = logits
        self.gathered, self.gathered_done = None, None
        if gather is not None:
            mine = torch.zeros((rows, 3), dtype=torch.float32, device=device if logits is None else logits.device)
            if logits is not None:
                mine[: logits.shape[0], :2] = logits
                mine[: logits.shape[0], 2] = 1.0
            full, done = gather.submit(mine)
            if done is not None:
                with torch.cuda.stream(gather.side):
                    self.gathered = torch.empty(full.shape, dtype=full.dtype, pin_memory=True)
                    self.gathered.copy_(full, non_blocking=True)
                    full.record_stream(gather.side)
                    self.gathered_done = torch.cuda.Event()
                    self.gathered_done.record(gather.side)
            else:
                self.gathered = full.cpu()
        self.labels, self.batch, self.batch_idx = labels, batch, batch_idx

    def flush(self, writer, trainer, model, on_batch) -> bool:
        """Write this batch's file; hand the gathered round to `on_batch`.  Returns whether ANY rank had reads in this round."""
        alive = self.host is not None
        if self.gathered is not None:
            if self.gathered_done is not None:
                self.gathered_done.synchronize()
            alive = bool((self.gathered[:, 2] > 0).any())
            if on_batch is not None and alive:
                on_batch(self.batch_idx, self.gathered)
        if self.host is not None:
            if self.event is not None:
                self.event.synchronize()
            writer.write_on_batch_end(trainer, model, (self.host, self.labels), None, self.batch, self.batch_idx, 0)
        return alive


def _drain_gather(pending, gatherer, rows, device, batch_idx, writer, trainer, model, on_batch):
    """Ranks run out of reads at different times, but a collective needs every rank: a rank that is done keeps contributing
    empty rounds until one round has come back with no valid row from anybody.  All ranks see the same gathered tensors one
    round behind, so all of them leave this loop after the same number of rounds."""
    while True:
        now = _Deferred(None, None, None, batch_idx, gatherer, rows, device)
        alive = pending.flush(writer, trainer, model, on_batch) if pending is not None else True
        pending = now
        batch_idx += 1
        if not alive:
            break
    pending.flush(writer, trainer, model, on_batch)


def run_predict(model, datamodule, writer, device: torch.device, *, rank: int = 0, gather: bool = False,
                on_batch=None) -> int:
    """Returns the number of reads this rank classified.  `gather`: every batch's logits are also all-gathered over the process
    group (RCCL over xGMI when the backend is "nccl"), off the compute stream, and handed to `on_batch(batch_idx, tensor)` one
    batch behind as a `[world * rows, 3]` host tensor (logit0, logit1, valid), rank r's rows at [r * rows, (r + 1) * rows)."""
    model.eval()
    copy_stream = torch.cuda.Stream(device)
    compute = torch.cuda.current_stream(device)
    trainer = SimpleNamespace(global_rank=rank)
    gatherer = LogitsGather(device) if gather else None
    rows = getattr(datamodule, "batch_size_per_device", 0)
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
            now = _Deferred(logits, labels, cur, batch_idx, gatherer, rows)
            if pending is not None:
                pending.flush(writer, trainer, model, on_batch)   # batch i-1: its copy finished while batch i was enqueued
            pending = now
            n_reads += logits.shape[0]
            batch_idx += 1
        _check_engine(model, device, batch_idx)
        if gatherer is not None:
            _drain_gather(pending, gatherer, rows, device, batch_idx, writer, trainer, model, on_batch)
        elif pending is not None:
            pending.flush(writer, trainer, model, on_batch)
    return n_reads


def _check_engine(model, device: torch.device, batch_idx: int) -> None:
    """Errors only the device sees (token ids outside the embedding table: the reference raises IndexError inside the forward,
    hyena.py:249) are reported by the NEXT engine call -- after the last batch there is none, so ask once more before its file
    is written.  The message names the batch range it can belong to."""
    net = getattr(model, "net", None)
    eng = getattr(net, "_engine", None)
    if eng is None:
        return
    from .engine import EngineError

    try:
        eng.check()
    except EngineError as e:
        raise EngineError(e.code, f"{e} [detected after batch {batch_idx - 1}, the last of this rank]") from None


def run_predict_native(model, feeder, writer, device: torch.device, *, rank: int = 0, gather: bool = False,
                       on_batch=None) -> int:
    """Predict loop over a `chimeralm_amd.feeder.BamFeeder`; same files as `run_predict` over `BamDataModule`."""
    from ._native import DT_U8

    model.eval()
    eng = model.net.engine(device)
    trainer = SimpleNamespace(global_rank=rank)
    gatherer = LogitsGather(device) if gather else None
    rows = feeder.batch_size
    n_reads, batch_idx = 0, 0
    pending: _Deferred | None = None
    cur = feeder.next()
    staged = eng.stage_host_ids(cur.ids_ptr, DT_U8, cur.row_stride, cur.n_reads, cur.n_tokens) if cur is not None else -1
    with torch.inference_mode():
        while cur is not None:
            nxt = feeder.next()                               # already decoded by the feeder thread, normally
            nxt_staged = (eng.stage_host_ids(nxt.ids_ptr, DT_U8, nxt.row_stride, nxt.n_reads, nxt.n_tokens)
                          if nxt is not None else -1)         # H2D of batch i+1 overlaps the forward of batch i
            # the 16-bit mode against the exact-fp32 kernels on this batch's first reads, where a self-check is due (HyenaDna.guard)
            # (a callable: the host copy + H2D of the sampled rows happens only on the few batches a check is due for)
            model.net.guard(eng, lambda c=cur: torch.from_numpy(c.ids[: c.n_reads, : c.n_tokens][
                model.net._sample_rows(c.n_reads, model.net._BATCH_ROWS)].copy()).to(device), n_tokens=cur.n_tokens, n_reads=cur.n_reads)
            logits = eng.forward_staged(staged, cur.n_reads)
            eng.stage_wait(staged)                            # the copy has left the slot ...
            feeder.release(cur)                               # ... which goes back to the decoder
            labels = torch.full((cur.n_reads,), -1, dtype=torch.int64)   # tokenizer.py:113: predict labels are all -1
            batch = {"id": torch.from_numpy(cur.names), "labels": labels}
            now = _Deferred(logits, labels, batch, batch_idx, gatherer, rows)
            if pending is not None:
                pending.flush(writer, trainer, model, on_batch)
            pending = now
            n_reads += cur.n_reads
            batch_idx += 1
            cur, staged = nxt, nxt_staged
        _check_engine(model, device, batch_idx)
        if gatherer is not None:
            _drain_gather(pending, gatherer, rows, device, batch_idx, writer, trainer, model, on_batch)
        elif pending is not None:
            pending.flush(writer, trainer, model, on_batch)
    return n_reads
