"""`ChimeraLM` factory, mirroring /root/reference/chimeralm/models/lm.py:9-61 (same fixed hyper-parameters)."""
from __future__ import annotations

import os
from functools import partial
from pathlib import Path

import torch

from .basic_module import ClassificationLit
from .hyena import BinarySequenceClassifier, HyenaDna


class ChimeraLM:
    @classmethod
    def new(cls, *, save_attention: bool = False, precision: str = "fp16c", chunk_reads: int = 256,
            selfcheck: bool | None = None, selfcheck_tol: float = 5e-4, selfcheck_every: int = 16) -> ClassificationLit:
        """Randomly initialised model of the production architecture (lm.py:39-61).  `precision` / `selfcheck`: see
        `chimeralm_amd.hyena.HyenaDna` -- the default "fp16c" is measured against the exact-fp32 kernels on the loaded weights
        before the first batch and replaced by them if it is more than `selfcheck_tol` off."""
        return ClassificationLit(
            net=HyenaDna(
                number_of_classes=2,
                backbone_name="hyenadna-small-32k-seqlen",
                head=BinarySequenceClassifier(input_dim=256, hidden_dim=512, num_layers=2, dropout=0.1,
                                              pooling_type="attention", activation="gelu", use_residual=True,
                                              save_attention=save_attention),
                precision=precision, chunk_reads=chunk_reads, selfcheck=selfcheck, selfcheck_tol=selfcheck_tol,
                selfcheck_every=selfcheck_every,
            ),
            optimizer=partial(torch.optim.AdamW, lr=0.0001, weight_decay=0.01),
            scheduler=partial(torch.optim.lr_scheduler.ReduceLROnPlateau, mode="min", factor=0.1, patience=10),
            criterion=torch.nn.CrossEntropyLoss(),
            compile=False,
        )

    @classmethod
    def from_pretrained(cls, model_name: str = "yangliz5/chimeralm", *, save_attention: bool = False,
                        precision: str = "fp16c", chunk_reads: int = 256, selfcheck: bool | None = None,
                        selfcheck_tol: float = 5e-4, selfcheck_every: int = 16) -> ClassificationLit:
        """Released weights (lm.py:12-37).  `model_name` is a local directory / file holding `model.safetensors`
        or a Lightning `.ckpt`; a Hub repo id is resolved through the local HF cache only (no network here)."""
        model = cls.new(save_attention=save_attention, precision=precision, chunk_reads=chunk_reads, selfcheck=selfcheck,
                        selfcheck_tol=selfcheck_tol, selfcheck_every=selfcheck_every)
        p = Path(model_name)
        if p.is_dir():
            p = p / "model.safetensors"
        if not p.exists():
            try:
                from huggingface_hub import hf_hub_download

                p = Path(hf_hub_download(model_name, "model.safetensors",
                                         local_files_only=os.environ.get("HF_HUB_OFFLINE", "0") == "1"))
            except Exception as e:  # noqa: BLE001
                raise FileNotFoundError(
                    f"weights {model_name!r} not found locally and not in the Hugging Face cache; pass a directory with "
                    "model.safetensors or use --ckpt") from e
        return model.load_reference_checkpoint(p)
