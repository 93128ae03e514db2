"""Prediction files, mirroring /root/reference/chimeralm/models/callbacks.py.

`resume_read_name` (:38-63) and `PredictionWriter.write_on_batch_end` (:79-150): one file
`{output_dir}/{global_rank}_{batch_idx}.txt` per batch, one line `name<TAB>label` per read, label = argmax of logits.

Difference kept on purpose: the id row is decoded with its length byte read as UNSIGNED, so read names of
128..255 characters work; the reference's collator raises on them (torch.tensor(..., dtype=int8) overflow,
tokenizer.py:168), so no input the reference accepts is treated differently.
"""
from __future__ import annotations

import logging
from pathlib import Path
from typing import Any

import torch

log = logging.getLogger(__name__)


def resume_read_name(bytes_data) -> str:
    if isinstance(bytes_data, torch.Tensor):
        if bytes_data.numel() == 0:
            return ""
        bytes_data = bytes_data.tolist()
    elif bytes_data is None or len(bytes_data) == 0:
        return ""
    data = [int(b) & 0xFF for b in bytes_data]
    n = data[0]
    if n <= 0 or n >= len(data):
        raise ValueError("Invalid read name data")
    return "".join(chr(b) for b in data[1: 1 + n] if 32 <= b <= 126)


class PredictionWriter:
    """Same constructor and hook signature as the reference (a Lightning BasePredictionWriter there)."""

    def __init__(self, output_dir: str | Path, write_interval: str = "batch") -> None:
        self.output_dir = Path(output_dir)
        self.interval = write_interval

    def write_on_batch_end(self, trainer: Any, pl_module: Any, prediction: Any, batch_indices: Any,
                           batch: dict[str, Any], batch_idx: int, dataloader_idx: int = 0) -> None:
        if prediction is None or "id" not in batch:
            log.error("batch %d: missing prediction or 'id'", batch_idx)
            return
        pred = prediction[0] if isinstance(prediction, (list, tuple)) else prediction
        if pred is None or pred.numel() == 0:
            log.warning("Empty prediction tensor for batch %d", batch_idx)
            return
        labels = pred.argmax(dim=1).cpu().tolist()     # device boundary: D2H + sync, as in the reference (:107)
        ids = batch["id"].cpu() if isinstance(batch["id"], torch.Tensor) else batch["id"]
        if len(labels) != len(ids):
            log.error("Size mismatch: predictions=%d, batch_ids=%d for batch %d", len(labels), len(ids), batch_idx)
            return
        lines = []
        for i, row in enumerate(ids):
            try:
                name = resume_read_name(row) or f"unknown_read_{i}"
            except ValueError:
                name = f"error_read_{i}"
            lines.append(f"{name}\t{labels[i]}\n")
        self.output_dir.mkdir(parents=True, exist_ok=True)
        rank = getattr(trainer, "global_rank", 0) if trainer is not None else 0
        with (self.output_dir / f"{rank}_{batch_idx}.txt").open("w") as f:
            f.writelines(lines)
