"""`Trainer` for the predict stage: what `lightning.pytorch.trainer.Trainer` does for the reference's
`trainer.predict(model=model, dataloaders=datamodule, ckpt_path=cfg.ckpt_path, return_predictions=False)`
(/root/reference/eval.py:74-80; configs/trainer/{default,gpu,ddp}.yaml for the constructor keywords), without Lightning:
one process per GPU (torchrun environment), the datamodule set up for this rank, the MI355X predict loop
(`chimeralm_amd.predict.run_predict`), the prediction-writer callbacks.  Training keywords (`min_epochs`, `max_epochs`,
`check_val_every_n_epoch`, ...) are accepted and ignored: this engine has no training path.
"""
from __future__ import annotations

import logging
from pathlib import Path

import torch

from . import distributed
from .predict import run_predict

log = logging.getLogger(__name__)


class Trainer:
    def __init__(self, accelerator: str = "gpu", devices: int | str = 1, callbacks: list | None = None, logger=None,
                 default_root_dir: str | None = None, deterministic: bool = False, strategy: str | None = None,
                 num_nodes: int = 1, **training_only):
        if accelerator not in ("gpu", "cuda", "auto"):
            raise ValueError(f"accelerator={accelerator!r}: this engine runs on MI355X GPUs only (no CPU path exists)")
        if num_nodes != 1:
            raise ValueError("one node (up to 8 GPUs over xGMI) is what the predict path shards over")
        self.devices, self.callbacks, self.logger = devices, list(callbacks or []), logger
        self.default_root_dir, self.deterministic, self.strategy = default_root_dir, deterministic, strategy
        self.callback_metrics: dict = {}
        self.global_rank, self.local_rank, self.world_size = distributed.env_world()

    def predict(self, model, dataloaders=None, datamodule=None, ckpt_path: str | Path | None = None,
                return_predictions: bool = False):
        dm = datamodule if datamodule is not None else dataloaders
        if dm is None or not hasattr(dm, "predict_dataloader"):
            raise ValueError("Trainer.predict needs a datamodule with predict_dataloader()")
        rank, local_rank, world = distributed.init_process_group()
        self.global_rank, self.local_rank, self.world_size = rank, local_rank, world
        if self.devices not in (-1, "auto") and int(self.devices) != world:
            log.warning("trainer.devices=%s but WORLD_SIZE=%d: one process per GPU is launched by torchrun "
                        "(python -m torch.distributed.run --nproc-per-node N eval.py ...)", self.devices, world)
        device = torch.device("cuda", local_rank)
        torch.cuda.set_device(device)
        if ckpt_path is not None:
            log.info("Loading checkpoint %s", ckpt_path)
            model.load_reference_checkpoint(ckpt_path)
        dm.setup("predict", world_size=world, rank=rank)
        writers = [cb for cb in self.callbacks if hasattr(cb, "write_on_batch_end")]
        if not writers:
            raise ValueError("no prediction-writer callback configured (configs/callbacks/write.yaml)")
        n = run_predict(model, dm, writers[0], device, rank=rank)
        distributed.barrier()
        log.info("[rank %d] %d reads predicted", rank, n)
        return None
