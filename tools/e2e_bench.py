"""End-to-end `predict` throughput on a synthetic BAM: native feeder -> staged H2D -> engine -> prediction files, one GPU.

    python tools/e2e_bench.py [--reads 6000] [--bases 8192] [--batch 256] [--precision fp16c]
    python tools/e2e_bench.py --bam tests/golden/test_chimric_reads.bam --batch 12 --repeat 5     (a real, ragged file: the reference's)

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
    ap.add_argument("--bam", type=Path, default=None, help="a real BAM instead of the synthetic one (batches padded on the left to their longest read)")
    ap.add_argument("--repeat", type=int, default=3, help="--bam: timed passes over the file after one warm-up pass")
    a = ap.parse_args()
    from feeder_bench import write_bam

    from chimeralm_amd import lm
    from chimeralm_amd.callbacks import PredictionWriter
    from chimeralm_amd.feeder import BamFeeder
    from chimeralm_amd.predict import run_predict_native

    device = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = lm.ChimeraLM.new(precision=a.precision)
    if a.bam is not None:
        # one warm-up pass (filters, workspace, the guard's first hearing, the [PAD] tables), then `repeat` timed passes; with the
        # share of tail tiles that lie wholly inside a [PAD] prefix (what csrc/pad_prefix.hip does not compute)
        import os

        import numpy as np

        with tempfile.TemporaryDirectory() as td:
            tiles = pad_tiles = toks = pads = 0
            with BamFeeder(a.bam, batch_size=a.batch) as f:
                while True:
                    fb = f.next()
                    if fb is None:
                        break
                    ids = np.asarray(fb.ids[:, : fb.n_tokens])
                    lead = (ids == 4).cumprod(axis=1).sum(axis=1)
                    tiles += ids.shape[0] * ((ids.shape[1] + 127) // 128)
                    pad_tiles += int((lead // 128).sum())
                    toks += ids.size
                    pads += int(lead.sum())
                    f.release(fb)
            print(f"{a.bam.name} at batch {a.batch}: {toks:,} tokens, {pads / toks:.1%} of them leading [PAD]; {tiles:,} tail tiles, "
                  f"{pad_tiles / tiles:.1%} wholly inside a [PAD] prefix; CLM_DEBUG={os.environ.get('CLM_DEBUG', '')!r}")
            for label, reps in (("warm-up", 1), ("timed", a.repeat)):
                t0, done = time.perf_counter(), 0
                for r in range(reps):
                    with BamFeeder(a.bam, batch_size=a.batch) as f:
                        done += run_predict_native(model, f, PredictionWriter(Path(td) / f"pred_{label}_{r}"), device)
                dt = time.perf_counter() - t0
                print(f"{label}: {done} reads in {reps} pass(es), {dt:.2f} s -> {done / dt:,.0f} reads/s end to end")
            print("guard:", {k: v for k, v in model.net.selfcheck_report.items() if k != "samples"})
        return
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
