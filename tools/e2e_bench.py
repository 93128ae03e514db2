"""End-to-end `predict` throughput on a synthetic BAM: native feeder -> staged H2D -> engine -> prediction files, one GPU.

    python tools/e2e_bench.py [--reads 6000] [--bases 8192] [--batch 256] [--precision fp16c]

Same loop as `python -m chimeralm_amd predict` (chimeralm_amd.predict.run_predict_native) with seeded random weights; the
clock starts after the first batch (filters / workspace for the length are built on it) and stops when the last prediction
file is written."""
from __future__ import annotations

import argparse
import sys
import tempfile
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=6000)
    ap.add_argument("--bases", type=int, default=8192)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--precision", default="fp16c")
    ap.add_argument("--min-bases", type=int, default=None, help="ragged file: read lengths uniform in [min-bases, bases]")
    a = ap.parse_args()
    from feeder_bench import write_bam

    from chimeralm_amd import lm
    from chimeralm_amd.callbacks import PredictionWriter
    from chimeralm_amd.feeder import BamFeeder
    from chimeralm_amd.predict import run_predict_native

    device = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = lm.ChimeraLM.new(precision=a.precision)
    with tempfile.TemporaryDirectory() as td:
        path = Path(td) / "synthetic.bam"
        write_bam(path, a.reads, a.bases, min_bases=a.min_bases)
        for label, n in (("warm-up", 2 * a.batch), ("timed", None)):
            out = Path(td) / f"pred_{label}"
            t0 = time.perf_counter()
            with BamFeeder(path, batch_size=a.batch, max_reads=n) as f:
                done = run_predict_native(model, f, PredictionWriter(out), device)
            dt = time.perf_counter() - t0
            files = len(list(out.glob("*.txt")))
            print(f"{label}: {done} reads, {files} prediction files, {dt:.2f} s -> {done / dt:,.0f} reads/s end to end")


if __name__ == "__main__":
    main()
