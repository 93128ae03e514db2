"""The predict loop that `lightning.Trainer.predict` runs in the reference
(/root/reference/chimeralm/__main__.py:307-317; call stack in SURVEY.md section 3.1), without Lightning:

    for batch in datamodule.predict_dataloader():      # collated on the host (pad-to-longest, file order)
        H2D copy on a side stream (pinned staging, overlapped with the previous batch's compute)
        logits, labels = model.predict_step(batch)      # MI355X engine
        writer.write_on_batch_end(...)                  # {rank}_{batch}.txt, "name<TAB>label"
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
    with torch.inference_mode():
        while staged is not None:
            compute.wait_stream(copy_stream)
            cur = staged
            nxt = next(it, None)                              # host collation of batch i+1 ...
            staged = _to_device(nxt, device, copy_stream) if nxt is not None else None   # ... and its H2D overlap
            logits, labels = model.predict_step(cur, batch_idx)
            if gather:
                logits_all = gather_logits(logits)
                if on_batch is not None:
                    on_batch(batch_idx, logits_all)
            writer.write_on_batch_end(trainer, model, (logits, labels), None, cur, batch_idx, 0)
            n_reads += logits.shape[0]
            batch_idx += 1
    return n_reads
