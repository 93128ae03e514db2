"""`chimeralm predict` for the MI355X engine -- same options as /root/reference/chimeralm/__main__.py:248-259:

    python -m chimeralm_amd predict DATA_PATH [-g GPUS] [-o OUTPUT] [-b BATCH] [-w WORKERS] [-c CKPT] [-r] [-v]

plus engine-only options `--weights` (directory/file with the released `model.safetensors`; the reference downloads
`yangliz5/chimeralm` from the Hub) and `--precision`.  `--gpus 0` (the reference's CPU mode) is refused: this engine
has no CPU path.  `filter` (reference :320-331) is the host-only step after predict; `web` is not built.
"""
from __future__ import annotations

import logging
import os
import subprocess
import sys
from pathlib import Path

import torch
import typer

app = typer.Typer(context_settings={"help_option_names": ["-h", "--help"]},
                  help="ChimeraLM predict on AMD MI355X: flag chimera artifacts from whole genome amplification.")
log = logging.getLogger("chimeralm_amd")


@app.callback()
def main():
    """ChimeraLM (MI355X engine)."""


@app.command()
def predict(
    data_path: Path = typer.Argument(..., help="Path to the dataset (BAM)"),
    gpus: int = typer.Option(1, "--gpus", "-g", help="Number of GPUs to use"),
    output_path: Path | None = typer.Option(None, "--output", "-o", help="Output path for predictions"),
    batch_size: int = typer.Option(12, "--batch-size", "-b", help="Batch size"),
    num_workers: int = typer.Option(0, "--workers", "-w", help="Number of workers"),
    ckpt_path: Path | None = typer.Option(None, "--ckpt", "-c", hidden=True, help="Path to the checkpoint file"),
    weights: str = typer.Option("yangliz5/chimeralm", "--weights", help="Directory/file with model.safetensors"),
    precision: str = typer.Option("fp16c", "--precision", help="arithmetic of the dense projections: fp16c (default: fp16 activations x fp16 hi + fp8 lo weights -- MLP weights plain fp16 -- at 16-bit MFMA rate, checked against the exact-fp32 kernels on the loaded weights before the first batch, replaced by them if more than --selfcheck-tol off) | fp32 (exact, the reference's) | fp16x3 (every operand as two halfs, three fp16 MFMAs per product: fp32-class accuracy at twice the exact rate, no check needed) | fp16 | bf16 (reduced precision)"),
    selfcheck_tol: float = typer.Option(5e-4, "--selfcheck-tol", help="largest |logit difference| from exact fp32 the fp16c mode may show in its self-check (0 disables the check)"),
    feeder: str = typer.Option("native", "--feeder", help="BAM input: native (C++ decoder thread, pinned ring) | python"),
    random: bool = typer.Option(False, "--random", "-r", help="Make the prediction not deterministic"),
    verbose: bool = typer.Option(False, "--verbose", "-v", help="Enable verbose output"),
    gather_logits: bool = typer.Option(False, "--gather-logits", help="multi-GPU: all-gather every batch's logits (RCCL, side "
                                       "stream) and let rank 0 also write them to logits.tsv (batch, rank, row, logit0, logit1)"),
):
    """Predict the given dataset using ChimeraLM."""
    logging.basicConfig(level=logging.DEBUG if verbose else logging.INFO, format="%(message)s")
    if gpus < 1:
        raise typer.BadParameter("this engine runs on MI355X GPUs only; use --gpus >= 1 (no CPU path exists)")
    if output_path is None:                       # README-documented default (the reference crashes here)
        output_path = data_path.with_suffix(".predictions")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if gpus > 1 and world == 1:                   # one process per GPU, launched before anything touches HIP
        from .distributed import free_port

        port = os.environ.get("MASTER_PORT") or str(free_port())      # a port that is free now, not a fixed one
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
               "--master-addr", "127.0.0.1", "--master-port", port, "-m", "chimeralm_amd", *sys.argv[1:]]
        raise typer.Exit(subprocess.call(cmd))

    from . import bam, callbacks, distributed, lm, predict as loop, tokenizer

    # CLM_DIST_BACKEND=gloo with CLM_RANKS_SHARE_GPU=1 is the one-GPU rehearsal of a multi-GPU run (tests/test_gpu_multirank.py):
    # RCCL refuses two ranks per device, everything else is the real path
    backend = os.environ.get("CLM_DIST_BACKEND")
    rank, local_rank, world = distributed.init_process_group(backend)
    if not random:
        torch.manual_seed(42)
    device = torch.device("cuda", local_rank % torch.cuda.device_count() if os.environ.get("CLM_RANKS_SHARE_GPU") == "1" else local_rank)
    torch.cuda.set_device(device)
    tok = tokenizer.load_tokenizer_from_hyena_model("hyenadna-small-32k-seqlen")
    if feeder not in ("native", "python"):
        raise typer.BadParameter("--feeder must be native or python")
    if batch_size % world != 0:                   # bam.py:142-146
        raise RuntimeError(f"Batch size ({batch_size}) is not divisible by the number of devices ({world}).")
    if ckpt_path is not None:
        log.info(f"Loading model from {ckpt_path}")
        model = lm.ChimeraLM.new(precision=precision, selfcheck=None if selfcheck_tol > 0 else False,
                                 selfcheck_tol=selfcheck_tol).load_reference_checkpoint(ckpt_path)
    else:
        log.info(f"Loading model weights {weights}")
        model = lm.ChimeraLM.from_pretrained(weights, precision=precision, selfcheck=None if selfcheck_tol > 0 else False,
                                             selfcheck_tol=selfcheck_tol)
    output_path.mkdir(parents=True, exist_ok=True)
    writer = callbacks.PredictionWriter(output_dir=output_path, write_interval="batch")
    if feeder == "native":
        from .feeder import BamFeeder

        with BamFeeder(data_path, batch_size=batch_size // world, max_tokens=tok.max_len_single_sentence, rank=rank,
                       world=world, pad_left=tok.padding_side == "left") as fd:
            n = loop.run_predict_native(model, fd, writer, device, rank=rank, gather=world > 1 and gather_logits,
                                        on_batch=_gathered_sink(output_path, rank) if gather_logits else None)
            log.info(f"[rank {rank}] feeder: {fd.stats()}")
    else:
        dm = bam.BamDataModule(tokenizer=tok, train_data_path=Path("dummy.bam"), predict_data_path=data_path,
                               batch_size=batch_size, num_workers=num_workers)
        dm.setup("predict", world_size=world, rank=rank)
        n = loop.run_predict(model, dm, writer, device, rank=rank, gather=world > 1 and gather_logits,
                             on_batch=_gathered_sink(output_path, rank) if gather_logits else None)
    distributed.barrier()
    rep = getattr(model.net, "selfcheck_report", None)
    if rep:
        log.info(f"[rank {rank}] precision {precision}: self-check against exact fp32 max |dlogit| {rep.get('max_abs_dlogit', 0.0):.2e} "
                 f"(threshold {rep.get('tol')}) -> {('FELL BACK to ' + rep.get('fallback_precision', 'fp16x3')) if rep.get('fallback') else 'kept'}")
    log.info(f"[rank {rank}] {n} reads; predictions saved to {output_path}")


def _gathered_sink(output_path: Path, rank: int):
    """Rank 0 appends the gathered [B, 2] logits of every batch to <output>/logits.tsv; rows are in rank order (rank r's
    shard of batch b are rows r*B/G .. (r+1)*B/G - 1)."""
    if rank != 0:
        return lambda batch_idx, gathered: None
    f = (output_path / "logits.tsv").open("w")

    def sink(batch_idx: int, gathered):
        rows = gathered.shape[0] // max(1, int(os.environ.get("WORLD_SIZE", "1")))
        for i, row in enumerate(gathered.tolist()):
            if row[2] > 0:                                    # rows of short / empty batches are padding
                f.write(f"{batch_idx}\t{i // rows}\t{i % rows}\t{row[0]:.7g}\t{row[1]:.7g}\n")
        f.flush()
    return sink


@app.command()
def filter(  # noqa: A001 - reference command name
    bam_path: Path = typer.Argument(..., help="Path to the BAM file"),
    predictions_path: Path = typer.Argument(..., help="Path to the predictions file"),
    output_prediction: bool = typer.Option(False, "--output-prediction", "-p", help="write summary of the predictions"),
    verbose: bool = typer.Option(False, "--verbose", "-v", help="Enable verbose output"),
):
    """Filter the BAM file by predictions."""
    logging.basicConfig(level=logging.DEBUG if verbose else logging.INFO, format="%(message)s")
    from .filter import filter_bam_by_predcition

    log.info(f"Filtering {bam_path} by predictions from {predictions_path}")
    res = filter_bam_by_predcition(bam_path, predictions_path, index=True, output_prediction=output_prediction)
    if res:
        log.info(f"kept {res['kept']} records, dropped {res['dropped']}: {res['sorted'] or res['filtered']}")


if __name__ == "__main__":
    app()
