"""Stage-by-stage comparison of the HIP engine with the CPU oracle (developer diagnostic, run on the GPU box):

    python tests/gpu_diag.py [--precision fp32] [--B 3] [--L 300] [--seed 0]

For every stage of block 0 the forward is stopped there (clm_debug_stop_after) and the raw buffer is compared with the
oracle's trace; then the full forward is compared (hidden, scores, pooled, logits).  Prints max-abs / relative errors.
"""
from __future__ import annotations

import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from chimeralm_amd import _native as N  # noqa: E402
from chimeralm_amd.engine import Engine  # noqa: E402
from oracle import hyena_oracle as ho  # noqa: E402


def decode(raw: np.ndarray, precision: str) -> np.ndarray:
    if precision == "fp32":
        return raw.view(np.float32)
    if precision in ("fp16", "fp16c"):
        return raw.view(np.float16).astype(np.float32)
    return (raw.view(np.uint16).astype(np.uint32) << 16).view(np.float32)


def report(name, got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    err = np.abs(got - ref)
    scale = np.abs(ref).max() + 1e-30
    bad = int((~np.isfinite(got)).sum())
    idx = np.unravel_index(int(np.nanargmax(err)), err.shape) if err.size else ()
    print(f"{name:28s} max_abs={np.nanmax(err):.3e} rel_to_max={np.nanmax(err) / scale:.3e} "
          f"rms_rel={np.sqrt(np.nanmean(err ** 2)) / (np.sqrt(np.mean(ref ** 2)) + 1e-30):.3e} nonfinite={bad} at={idx}",
          flush=True)
    return np.nanmax(err) / scale


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--B", type=int, default=3)
    ap.add_argument("--L", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--stages", type=int, default=1)
    a = ap.parse_args()
    prec, B, L = a.precision, a.B, a.L
    es = 4 if prec == "fp32" else 2
    Lp = (L + 63) // 64 * 64
    sd = ho.make_state_dict(a.seed, head_scale=3.0)
    ids_np, _ = ho.synthetic_batch(0, B, L - 1, seed=77)
    ids_np[:, :5] = 4  # a few [PAD] tokens on the left, like a collated batch
    ids = torch.from_numpy(ids_np.astype(np.int64))
    t0 = time.time()
    trace = {}
    ref_logits = ho.forward(ids, sd, trace=trace)
    ref64 = ho.forward(ids, sd, dt=torch.float64)
    print(f"oracle fp32 {time.time() - t0:.1f}s; |fp32-fp64| logits = {(ref_logits.double() - ref64).abs().max():.2e}")

    eng = Engine("cuda:0", precision=prec, chunk_reads=32)
    eng.load_state_dict(sd)
    dev_ids = ids.cuda()

    def run(layer=-1, stage=-1):
        eng.debug_stop_after(layer, stage)
        out = eng.forward(dev_ids)
        torch.cuda.synchronize()
        return out

    S = {n: i for i, n in enumerate(N.STAGES)}
    if a.stages:
        run(-1, S["embed"])
        report("embed", eng.debug_fetch("h", (B, L, 256)), trace["embed"])
        for i in range(4):
            report(f"filter.{i}", eng.debug_fetch(f"filter.{i}", (L, 256)), trace[f"l{i}.filter"].T)
        for lay in range(2):
            run(lay, S["ln1_in_proj"])
            z = decode(eng.debug_fetch("z", (B, 768, Lp * es), np.uint8), prec).reshape(B, 768, Lp)[:, :, :L]
            report(f"l{lay}.in_proj z", z, trace[f"l{lay}.in_proj"])
            run(lay, S["short_long_conv"])
            y = decode(eng.debug_fetch("y", (B, 256, Lp * es), np.uint8), prec).reshape(B, 256, Lp)[:, :, :L]
            report(f"l{lay}.conv y", y, trace[f"l{lay}.gated"])
            run(lay, S["out_proj"])
            report(f"l{lay}.out_proj h", eng.debug_fetch("h", (B, L, 256)), trace[f"l{lay}.mixer_out"])
            run(lay, S["ln2_fc1_gelu"])
            u = decode(eng.debug_fetch("u", (B, L, 1024 * es), np.uint8), prec).reshape(B, L, 1024)
            p = f"{ho.BB}layers.{lay}."
            u_ref = torch.nn.functional.gelu(ho._lin(ho._ln(trace[f"l{lay}.mixer_out"], sd, p + "norm2", torch.float32),
                                                     sd, p + "mlp.fc1", torch.float32), approximate="tanh")
            report(f"l{lay}.fc1 u", u, u_ref)
            run(lay, S["fc2"])
            report(f"l{lay}.fc2 h", eng.debug_fetch("h", (B, L, 256)), trace[f"l{lay}.out"])
    logits = run().cpu().numpy()
    report("hidden (l3.out)", eng.debug_fetch("hidden", (B, L, 256)), trace["l3.out"])
    report("scores", eng.debug_fetch("scores", (B, L)), trace["scores"])
    report("pooled", eng.debug_fetch("pooled", (B, 256)), trace["pooled"])
    report("logits vs fp32 oracle", logits, ref_logits)
    report("logits vs fp64 oracle", logits, ref64)
    print("logits gpu   ", logits.tolist())
    print("logits oracle", ref_logits.tolist())
    print("labels equal:", (logits.argmax(1) == ref_logits.numpy().argmax(1)).all())


if __name__ == "__main__":
    main()
