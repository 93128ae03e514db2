"""`ClassificationLit` for the predict path, mirroring /root/reference/chimeralm/models/basic_module.py.

Same constructor (`net, optimizer, scheduler, criterion, *, compile`), `forward(input_ids, input_quals)` (:67-77) and
`predict_step(batch, batch_idx) -> (logits, labels)` (:177-187).  When `lightning` is importable the class is a
`LightningModule` (so `Trainer.predict` drives it as in the reference); otherwise a plain `nn.Module` -- the training
hooks and torchmetrics of the reference (:43-65, :87-175) are out of scope for an inference engine.
"""
from __future__ import annotations

from typing import Any

import torch
from torch import nn

try:  # optional: not installed in the build image
    from lightning import LightningModule as _Base
except ImportError:  # pragma: no cover - depends on the environment
    _Base = nn.Module


class ClassificationLit(_Base):
    def __init__(self, net: nn.Module, optimizer: Any = None, scheduler: Any = None, criterion: nn.Module | None = None,
                 *, compile: bool = False):  # noqa: A002 - reference keyword
        super().__init__()
        if not hasattr(net, "number_of_classes"):
            raise AttributeError("net must expose `number_of_classes` (basic_module.py:43-58)")
        self.net = net
        self.criterion = criterion
        self.optimizer_factory, self.scheduler_factory, self.compile_flag = optimizer, scheduler, compile

    def forward(self, input_ids: torch.Tensor, input_quals: torch.Tensor | None = None) -> torch.Tensor:
        return self.net(input_ids, input_quals)

    def predict_step(self, batch: dict[str, torch.Tensor], batch_idx: int = 0):
        logits = self.forward(batch["input_ids"], batch.get("input_quals", None))
        return logits, batch["labels"]

    # ---- checkpoint loading (reference: PyTorchModelHubMixin.from_pretrained / Lightning ckpt_path) ----
    def load_reference_checkpoint(self, path) -> "ClassificationLit":
        """Load `model.safetensors` (HF hub layout of `yangliz5/chimeralm`) or a Lightning `.ckpt`."""
        path = str(path)
        if path.endswith(".safetensors"):
            from safetensors.torch import load_file

            sd = load_file(path)
        else:
            # a Lightning .ckpt is a pickle: load tensors only.  Checkpoints that also pickle arbitrary objects (hyper-parameter
            # partials, callbacks) need the full unpickler, which executes code from the file -- opt in explicitly.
            import os
            import pickle

            try:
                obj = torch.load(path, map_location="cpu", weights_only=True)
            except (pickle.UnpicklingError, RuntimeError) as e:
                if os.environ.get("CLM_TRUST_CHECKPOINT") != "1":
                    raise RuntimeError(
                        f"{path} holds pickled Python objects besides tensors; loading it would run code from the file. "
                        "Re-run with CLM_TRUST_CHECKPOINT=1 if you trust its origin, or convert it to safetensors.") from e
                obj = torch.load(path, map_location="cpu", weights_only=False)
            sd = obj.get("state_dict", obj)
        own = self.state_dict()
        # safetensors drops the aliases of the shared sine module; restore them from `.1.freq`
        for k in list(own):
            if k not in sd and (".implicit_filter.3.freq" in k or ".implicit_filter.5.freq" in k):
                src = k.replace(".3.freq", ".1.freq").replace(".5.freq", ".1.freq")
                if src in sd:
                    sd[k] = sd[src]
        missing = [k for k in own if k not in sd]
        if missing:
            raise KeyError(f"checkpoint lacks {len(missing)} tensors, e.g. {missing[:3]}")
        self.load_state_dict({k: sd[k] for k in own}, strict=True)
        return self
