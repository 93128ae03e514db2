"""`python eval.py ckpt_path=<ckpt|safetensors> +data.predict_data_path=<reads.bam> [data.batch_size=12] [hydra.run.dir=<out>]`

The Hydra entry route of the reference's predict path (/root/reference/eval.py:33-101, configs/eval.yaml): compose
configs/eval.yaml, instantiate datamodule / model / callbacks / trainer from their `_target_`s and run
`trainer.predict(model=model, dataloaders=datamodule, ckpt_path=cfg.ckpt_path, return_predictions=False)`.
Prediction files land in `${paths.output_dir}/predicts/{rank}_{batch}.txt` (configs/callbacks/write.yaml).
The built-in composer (chimeralm_amd/config.py) reads the files -- whether or not hydra-core is installed: one behaviour everywhere.
Multi-GPU: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 eval.py trainer=ddp ...`.
"""
from __future__ import annotations

import logging
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

log = logging.getLogger("eval")


def evaluate(cfg):
    from chimeralm_amd.config import instantiate, instantiate_callbacks

    assert cfg.ckpt_path
    log.info(f"Instantiating datamodule <{cfg.data._target_}>")
    datamodule = instantiate(cfg.data)
    log.info(f"Instantiating model <{cfg.model._target_}>")
    model = instantiate(cfg.model)
    log.info("Instantiating callbacks...")
    callbacks = instantiate_callbacks(cfg.get("callbacks"))
    log.info(f"Instantiating trainer <{cfg.trainer._target_}>")
    trainer = instantiate(cfg.trainer, callbacks=callbacks, logger=[])
    object_dict = {"cfg": cfg, "datamodule": datamodule, "model": model, "logger": [], "trainer": trainer}
    if getattr(datamodule, "predict_data_path", None) is None:
        raise NotImplementedError("trainer.test: this build covers the predict stage only; pass +data.predict_data_path=<bam>")
    trainer.predict(model=model, dataloaders=datamodule, ckpt_path=cfg.ckpt_path, return_predictions=False)
    return trainer.callback_metrics, object_dict


def main(argv: list[str] | None = None):
    logging.basicConfig(level=logging.INFO, format="%(message)s")
    argv = list(sys.argv[1:] if argv is None else argv)
    # One composer everywhere: chimeralm_amd/config.py reads the same files and override grammar.  (`hydra.compose` outside
    # `@hydra.main` sets no HydraConfig, so configs/paths/default.yaml's ${hydra:runtime.output_dir} could not resolve, and
    # `hydra.run.dir=` would be taken as a plain override: an installed hydra-core would have crashed this route, not helped it.)
    from chimeralm_amd.config import compose

    out = next((a.split("=", 1)[1] for a in argv if a.startswith("hydra.run.dir=")), None)
    cfg = compose(ROOT / "configs", "eval.yaml", [a for a in argv if not a.startswith("hydra.")], output_dir=out)
    if cfg.get("extras", {}).get("print_config"):
        import yaml

        from chimeralm_amd.config import to_container

        log.info(yaml.safe_dump(to_container(cfg), sort_keys=False))
    return evaluate(cfg)


if __name__ == "__main__":
    main()
