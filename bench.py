"""Headline benchmark of the ChimeraLM `predict` hot path on MI355X (contract: see the task description).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 256] [--bases 8192] [--precision P]     (N > 1: starts its own ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A step = one forward of the whole path (embedding -> 4 Hyena blocks -> ln_f -> attention pooling -> classifier) over
one global batch of synthetic reads (A/C/G/T uniform, N with p=0.001, [SEP] appended), token ids
already resident in HBM.  With N > 1 the batch is read-sharded (B/N contiguous reads per rank, weights replicated)
and the per-read logits are all-gathered over RCCL inside the step; the global batch stays fixed (strong scaling),
as BASELINE.json's configs[2]/[3] state (8k-bp reads, batch = 256 at 1/2/4/8 GPUs).

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (longest accumulated device time over the timed
region, measured with HIP events on the launch stream by the engine's own taps); `cpu_baseline` is the CPU oracle
(oracle/hyena_oracle.py, a port of the reference path) timed on the host cores of this box on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent))

D, DI, NLAYER = 256, 1024, 4
PEAK_TFLOPS = {"fp32": 157.3, "bf16": 2500.0, "fp16": 2500.0, "fp16c": 2500.0, "fp16x3": 2500.0}      # dense MFMA peaks, MI355X_MICROARCH.md
SUSTAINED_MFMA16_TFLOPS = 1630.0   # measured, see roofline["peak_sustained_measured"]
SCLK_UNDER_TAIL_HZ = 2.15e9        # shader clock during the forward loop (profiles/r03_power.txt: 2.15 GHz at 1.26 kW)
# MFMA instructions issued per algorithmic product: fp16c multiplies every activation fragment with the hi AND the lo half of
# the weight pair (include/chimeralm_hip.h CLM_PREC_F16C)
# fp16c: in_proj + out_proj (a third of a block's products) run hi on fp16 MFMAs + lo on fp8 MFMAs at half their cycles; the MLP is plain
# round 4: + the activations' lo term (a second fp8 MFMA per row tile and 64-deep group) in the same products: out_proj (y's lo
# plane) and in_proj (the LayerNorm-1 lo tile), or out_proj and the score layer in the last block's kernel
MFMA_ISSUE_FACTOR = {"fp32": 1.0, "bf16": 1.0, "fp16": 1.0, "fp16c": 1.0 + 0.5 / 3 + 0.5 / 3, "fp16x3": 3.0}
MLP_LO_ISSUE = 0.5 * 2 / 3          # fc1 / fc2 (2/3 of the products) with their lo half on the fp8 MFMA too (the guard's second level)
# arithmetic behind each --precision, as the JSON line's "dtype" words it
DTYPE_NOTE = {"fp32": "fp32 (v_mfma_f32_32x32x2_f32, exact)",
              "fp16x3": "every operand of a dense projection as two halfs (hi = fp16(x), lo = fp16(x - hi): ~21 bits), a product as three fp16 MFMAs into the fp32 accumulator; fp32 everywhere else, z / y / residual stream fp32 in HBM: logits within ~1e-5 of exact fp32", "fp16": "fp16 MFMA inputs, fp32 accumulate (reduced precision: outside the 1e-3 gate)",
              "bf16": "bf16 MFMA inputs, fp32 accumulate (reduced precision: outside the 1e-3 gate)",
              "fp16c": "fp16 activations with e5m2 lo bytes for y, the gated z rows and the ln_f tile (~15 bits); in_proj / out_proj / score weights as fp16 hi + fp8 lo (fp16 MFMA + block-scaled fp8 MFMAs into one fp32 accumulator), MLP weights plain fp16 unless the guard switches them to hi + lo; fp32 LayerNorm / FFT / softmax; the module measures the mode against the exact-fp32 kernels of the same engine on the loaded weights (`guard`) and falls back to them above its threshold"}
PEAK_HBM_GBS = 8000.0
# algorithmic FLOPs per token of each GEMM stage
STAGE_FLOPS_PER_TOKEN = {"ln1_in_proj": 2 * D * 3 * D, "out_proj": 2 * D * D, "ln2_fc1_gelu": 2 * D * DI,
                         "fc2": 2 * D * DI, "lnf_pool_score": 2 * D * D + 2 * D,
                         "out_proj_ln2_mlp": 2 * D * D + 4 * D * DI, "ln2_mlp": 4 * D * DI}


# algorithmic HBM bytes per token of the GEMM stages, by activation element size (h is fp32 in every mode)
TAIL_BYTES_PER_TOKEN = {"out_proj_ln2_mlp": {2: D * 2 + 2 * D * 4, 4: D * 4 + 2 * D * 4},
                        "ln2_mlp": {2: 2 * D * 4, 4: 2 * D * 4},
                        "ln1_in_proj": {2: D * 4 + 3 * D * 2, 4: D * 4 + 3 * D * 4}}
Z_ROWS = 3 if "raw_z" in os.environ.get("CLM_DEBUG", "").split(",") else 2   # rows per channel the fused in_proj stage hands the convolution (gated: x0f, g)


def stage_bytes_per_token(stage: str, es: int) -> float:
    """Algorithmic HBM bytes per token and launch of the bandwidth-bound stages (es = activation element size)."""
    # read z (blocks 1-3: x0f, g in the gated hand-over, x0 | x1 | v before round 3; block 0: one id byte), write y
    return {"short_long_conv": ((NLAYER - 1) * Z_ROWS * D * es + 1) / NLAYER + D * es,
            "embed": 1 + D * 4, "softmax_pool": D * 4 + 8}.get(stage, 0.0)


def kernel_sources_sha() -> str:
    """Identity of the kernel sources a PMC digest belongs to: sha1 over chimeralm_amd/csrc/*.{hip,h} (names + contents)."""
    import hashlib

    h = hashlib.sha1()
    csrc = Path(__file__).resolve().parent / "chimeralm_amd" / "csrc"
    for f in sorted(list(csrc.glob("*.hip")) + list(csrc.glob("*.h"))):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def measured_traffic(stage: str, config: dict, dtype: str):
    """HBM bytes per launch of `stage` from the newest committed PMC digest (profiles/rNN_traffic.json, written by
    tools/profile_round.sh + tools/profile_digest.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes with the
    gfx950 FETCH_SIZE x2 correction) -- only if it was collected on exactly this workload AND on exactly these kernel
    sources (the digest records kernel_sources_sha(); a kernel edited since makes it stale: then None, never an old number)."""
    for f in sorted(Path(__file__).resolve().parent.glob("profiles/r*_traffic.json"), reverse=True):
        try:
            d = json.loads(f.read_text())
        except (OSError, ValueError):
            continue
        if d.get("config") == config and d.get("dtype") == dtype and stage in d.get("stages", {}):
            if d.get("kernel_sources_sha") != kernel_sources_sha():
                return {"stale": f"profiles/{f.name} was collected on other kernel sources"}
            e = d["stages"][stage]
            return {"hbm_bytes_per_launch": e["hbm_bytes_per_dispatch"], "source": f"profiles/{f.name}",
                    "mfma_busy": (e.get("mfma_busy_cycles_per_dispatch"), e.get("pmc_pass_avg_us"))}
    return None


def synthetic_ids(batch_index: int, batch: int, bases: int) -> np.ndarray:
    rng = np.random.default_rng(1234 + batch_index)
    ids = rng.integers(7, 11, size=(batch, bases), dtype=np.uint8)
    ids[rng.random((batch, bases)) < 0.001] = 11
    return np.concatenate([ids, np.ones((batch, 1), np.uint8)], axis=1)


def cpu_baseline(bases: int, budget_s: float = 20.0) -> dict:
    from oracle import hyena_oracle as ho

    # the threads this process may really use: its affinity mask, capped at the GPU box's CPU share (16 per GPU)
    cores = int(os.environ.get("CLM_CPU_THREADS", min(len(os.sched_getaffinity(0)), os.cpu_count() or 1, 16)))
    torch.set_num_threads(cores)
    sd = ho.make_state_dict(0)
    b = 4
    ids = torch.from_numpy(synthetic_ids(10_000, b, bases).astype(np.int64))
    ho.forward(ids[:1], sd)                       # warm-up (thread pool, FFT plans)
    t0, n = time.perf_counter(), 0
    while True:
        ho.forward(ids, sd)
        n += b
        if time.perf_counter() - t0 > budget_s or n >= 64:
            break
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "reads/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} synthetic {bases}-bp reads in batches of {b}, fp32 torch-CPU oracle (oracle/hyena_oracle.py)"}


def bench_transformer(a):
    """Same step / timing contract for the reference's second net (configs/model/transformer.yaml):
    one forward of SequenceCNNTransformer over the global batch of synthetic reads, ids resident in HBM."""
    from chimeralm_amd import distributed as cdist
    from chimeralm_amd.transformer import SequenceCNNTransformer

    rank, local_rank, world = cdist.init_process_group("nccl")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    lo, hi = cdist.shard_bounds(a.batch, rank, world)
    L = a.bases + 1
    torch.manual_seed(0)
    net = SequenceCNNTransformer(vocab_size=12, max_len=32768, num_encoder_layers=12, precision=a.precision)
    n_data = max(1, min(4, a.steps))
    batches = [torch.from_numpy(synthetic_ids(i, a.batch, a.bases)[lo:hi]).to(device) for i in range(n_data)]

    def step(i):
        out = net(batches[i % n_data])
        return cdist.gather_logits(out) if world > 1 else out

    for i in range(a.warmup):
        step(i)
    cdist.barrier()
    torch.cuda.synchronize(device)
    if a.precision != "fp32":
        net.profile_read(reset=True)
        net.profile_enable(True)
    t0 = time.perf_counter()
    for i in range(a.steps):
        out = step(i)
    torch.cuda.synchronize(device)
    cdist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    assert out.shape[0] == a.batch and bool(torch.isfinite(out).all())
    prof = net.profile_read(reset=True) if a.precision != "fp32" else {}
    if rank == 0:
        L3 = L // 8
        # dense FLOPs per read: conv stack (K = 768 GEMMs at L, L/2, L/4 positions) + 12 x (QKV, out, FFN) + attention products
        conv = 2 * 768 * 256 * ((L // 2) * 2 + (L // 4) * 2 + L3 * 2)
        enc = 12 * L3 * 2 * 256 * (768 + 256 + 2 * 1024)
        att = 12 * 2 * 2 * L3 * L3 * 256
        flops = (conv + enc + att) * a.batch * a.steps / elapsed / 1e12
        res = {"metric": f"reads/sec, SequenceCNNTransformer, {a.bases}-bp reads batch={a.batch}", "value": a.batch * a.steps / elapsed,
               "unit": "reads/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": a.precision,
               "data": "synthetic reads (seeded), seeded random-init weights of the configured architecture",
               "config": {"workload": f"synthetic {a.bases}-bp reads, global batch {a.batch}, 1 forward per step, "
                                      "SequenceCNNTransformer (12 layers)", "global_batch": a.batch, "tokens_per_read": L,
                          "reads_per_gpu": hi - lo},
               "selfcheck": {k: v for k, v in net.selfcheck_report.items() if k != "samples"} or None,
               "roofline": {"bound": "mfma", "kernel": "whole forward (conv stack + encoder + attention)", "achieved": flops,
                            "peak": PEAK_TFLOPS[a.precision], "unit": "TFLOP/s", "frac": flops / PEAK_TFLOPS[a.precision],
                            "traffic": None}}
        if prof:
            # per-kernel rooflines from the engine's HIP-event taps (launch stream), averaged over the timed launches:
            # algorithmic dense FLOPs of the stage for this rank's reads / its accumulated time
            reads = (hi - lo) * a.steps
            per = {"conv_stack_pe_ln": conv, "attention": att, "encoder_layer": enc}
            kr = {}
            for k, fl in per.items():
                ms, n = prof[k]
                if n:
                    ach = fl * reads / (ms * 1e-3) / 1e12
                    kr[k] = {"bound": "mfma", "achieved": ach, "peak": PEAK_TFLOPS[a.precision], "unit": "TFLOP/s",
                             "frac": ach / PEAK_TFLOPS[a.precision], "avg_launch_ms": ms / n, "launches": n,
                             "share_of_device_time": ms / max(1e-9, sum(v[0] for v in prof.values()))}
            tr = None
            for f in sorted(Path(__file__).resolve().parent.glob("profiles/r*_tf_traffic.json"), reverse=True):
                try:
                    tj = json.loads(f.read_text())
                except (OSError, ValueError):
                    continue
                if tj.get("config") == res["config"] and tj.get("dtype") == a.precision:
                    tr = (tj, f.name)
                    break
            if tr:
                fresh = tr[0].get("kernel_sources_sha") == kernel_sources_sha()
                for k, stg in (("attention", "attention"), ("encoder_layer", "encoder_layer")):
                    e = tr[0].get("stages", {}).get(stg)
                    if e and k in kr:
                        kr[k]["traffic" if fresh else "traffic_stale"] = e["hbm_bytes_per_dispatch"] if fresh else f"profiles/{tr[1]} is of other kernel sources"
            res["roofline_kernels"] = kr
            dom = max(kr, key=lambda k: kr[k]["share_of_device_time"]) if kr else None
            if dom:
                res["roofline"] = {"bound": "mfma", "kernel": dom, **{k: v for k, v in kr[dom].items()}, "traffic": kr[dom].get("traffic")}
                res["roofline_whole_forward"] = {"achieved": flops, "peak": PEAK_TFLOPS[a.precision], "unit": "TFLOP/s",
                                                 "frac": flops / PEAK_TFLOPS[a.precision]}
        print(json.dumps(res), flush=True)
    cdist.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="global batch (reads per step)")
    ap.add_argument("--bases", type=int, default=8192)
    ap.add_argument("--precision", default=os.environ.get("CLM_PRECISION", "fp16c"),
                    help="fp16c (default: the 16-bit-rate mode inside the reference's 1e-3 tolerance) | fp32 (exact) | "
                         "fp16 | bf16 (reduced precision, outside the tolerance)")
    ap.add_argument("--chunk-reads", type=int, default=256)
    ap.add_argument("--head-scale", type=float, default=3.0, help="factor on every head layer of the default-init weights")
    ap.add_argument("--logit-gap", type=float, default=3.6, help="mean |logit0 - logit1| the output layer is rescaled to (0: off)")
    ap.add_argument("--selfcheck-tol", type=float, default=5e-4, help="the guard's threshold (half the reference's tolerance)")
    ap.add_argument("--no-guard", action="store_true", help="developer A/B runs (timing-only library builds give wrong logits): selfcheck off")
    ap.add_argument("--mlp-lo", action="store_true", help="fp16c: start at the guard's second level (fc1 / fc2 on hi + lo weights) -- "
                    "what the product runs on weights whose MLP rounding shows; to price that level")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp32-leg", action="store_true", help="skip the extra exact-fp32 timing (fp32_exact_reads_per_s)")
    ap.add_argument("--net", default="hyena", choices=["hyena", "transformer"],
                    help="hyena = the production predict path (the headline metric); transformer = SequenceCNNTransformer")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` by itself: one rank per GPU, started here -- BEFORE anything in this process touches HIP -- under
        # torch.distributed.run on a port that is free now (as `chimeralm_amd predict -g N` does).  Rank 0's JSON line is the
        # children's stdout, relayed as it is; this process exits with their status.  The driver's own torchrun form is unchanged.
        import subprocess

        from chimeralm_amd.distributed import free_port

        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", os.environ.get("MASTER_PORT") or str(free_port()), str(Path(__file__).resolve()), *sys.argv[1:]]
        raise SystemExit(subprocess.call(cmd))
    if a.net == "transformer":
        return bench_transformer(a)

    from chimeralm_amd import distributed as cdist, lm
    from chimeralm_amd.engine import Engine
    import warnings

    # CLM_DIST_BACKEND=gloo is the rehearsal switch of tests/test_gpu_multirank.py: several ranks on ONE GPU (RCCL refuses two
    # ranks per device), everything else -- sharding, gather, max-over-ranks timing, the JSON line -- as in the real run
    backend = os.environ.get("CLM_DIST_BACKEND", "nccl")
    rank, local_rank, world = cdist.init_process_group(backend)
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    device = torch.device("cuda", local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank)
    torch.cuda.set_device(device)
    lo, hi = cdist.shard_bounds(a.batch, rank, world)
    L = a.bases + 1

    # The PRODUCT is what is benched: `HyenaDna(precision=...)` with its guard on, on seeded weights from the
    # product's own module (the same on every rank; the oracle package is only touched by the cpu_baseline leg) brought to a
    # REALISTIC logit scale -- default-init heads give logits of +-0.03, against which any self-check is vacuous; the released model
    # answers its notebook example with P = 0.973 / 0.027 (reference notebooks/attention.ipynb:280), a logit gap of 3.6.  Every head
    # layer x `--head-scale` (3, as the parity tests' weight draws), then the output layer rescaled so that the mean gap over 16
    # seeded reads -- measured with the exact-fp32 engine -- is 3.6.
    torch.manual_seed(0)
    model = lm.ChimeraLM.new(precision=a.precision, selfcheck_tol=a.selfcheck_tol, selfcheck=False if a.no_guard else None)
    net = model.net
    n_data = max(1, min(4, a.steps))                 # a few distinct resident batches, cycled
    batches = [torch.from_numpy(synthetic_ids(i, a.batch, a.bases)[lo:hi]).to(device) for i in range(n_data)]
    weights_note = {"init": "chimeralm_amd.lm.ChimeraLM.new(), torch seed 0", "head_scale": a.head_scale}
    with torch.no_grad():
        for prm in net.head.parameters():
            prm.mul_(a.head_scale)
        if a.logit_gap > 0:
            cal = Engine(device, precision="fp32", chunk_reads=a.chunk_reads)
            cal.load_state_dict(model.state_dict())
            cal_ids = torch.from_numpy(synthetic_ids(20_000, 16, a.bases)).to(device)
            lg = cal.forward(cal_ids).float().cpu()
            gap0 = float((lg[:, 0] - lg[:, 1]).abs().mean())
            cal.close()
            k = a.logit_gap / max(gap0, 1e-9)
            net.head.output_layer.weight.mul_(k)
            net.head.output_layer.bias.mul_(k)
            weights_note.update(logit_gap_target=a.logit_gap, logit_gap_before=gap0, output_layer_rescale=k,
                                max_abs_logit=float(lg.abs().max()) * k)
    with warnings.catch_warnings():                  # (a fallback is REPORTED in the line, not printed over it)
        warnings.simplefilter("ignore", RuntimeWarning)
        eng = net.engine(device)                     # loads the weights; the guard has its first hearing on the first batch
        eng.reserve(hi - lo, L)
        if a.mlp_lo and a.precision == "fp16c":
            eng.set_mlp_compensation(True)
            net._mlp_lo = True
            net.selfcheck_report["mlp_compensation"] = True
        net.guard(eng, batches[0])
    rep = net.selfcheck_report
    logits2 = [torch.empty((hi - lo, 2), dtype=torch.float32, device=device) for _ in range(2)]
    logits = logits2[0]
    gather = cdist.LogitsGather(device, timed=True)   # N > 1: the all-gather runs on its own stream, behind the forward it belongs to
    dev_ids = cdist.assert_distinct_devices(device)   # every rank's PCI identity; raises under nccl if two ranks share a GPU
    gather_done = [None, None]               # `done` event of the gather that last read logits2[k]

    def step(i):
        buf = logits2[i & 1]
        if gather_done[i & 1] is not None:   # forward i overwrites the buffer gather i - 2 read on the side stream: order them
            torch.cuda.current_stream(device).wait_event(gather_done[i & 1])
        # exactly HyenaDna.forward: the guard (a self-check where one is due: every `selfcheck_every`-th batch) + the engine
        net.guard(eng, batches[i % n_data])
        eng.forward(batches[i % n_data], out=buf)
        if world == 1:
            return buf
        full, gather_done[i & 1] = gather.submit(buf)   # forward i + 1 is enqueued while gather i is in flight
        return full

    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize(device)
    gather.spans_ms(reset=True)
    eng.profile_read(reset=True)
    eng.profile_enable(True)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    checks0 = rep.get("checks", 0)
    cdist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for i in range(a.steps):
        ev[i][0].record()
        out = step(i)
        ev[i][1].record()
    torch.cuda.synchronize(device)
    cdist.barrier()
    elapsed = time.perf_counter() - t0
    eng.profile_enable(False)
    rank_ms = [elapsed / a.steps * 1e3]
    if world > 1:
        # every rank's own time per step travels with the line: the first real multi-GPU run shows skew without a second run
        allms = [None] * world
        torch.distributed.all_gather_object(allms, rank_ms[0])
        rank_ms = [float(x) for x in allms]
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    prof = eng.profile_read(reset=True)
    eff = eng.effective_precision(L)                 # what reads of this length ran in, after the guard's verdict(s)
    guard = None
    if a.precision != "fp32":
        guard = {"verdict": (f"fell back to {rep.get('fallback_precision', 'fp16x3')}" if rep.get("fallback") else
                             "kept, MLP weights switched to hi + lo" if rep.get("mlp_compensation") else
                             "kept" if net.selfcheck else "not guarded (selfcheck off for this mode)"),
                 "effective_precision": eff, "max_abs_dlogit_vs_exact_fp32": rep.get("max_abs_dlogit"), "tol": rep.get("tol"),
                 "f16c_min_len": rep.get("f16c_min_len"), "selfcheck_every": net.selfcheck_every,
                 "checks_inside_timed_region": rep.get("checks", 0) - checks0,
                 "samples": [(x["sample"], x["max_abs_dlogit"]) for x in rep.get("samples", [])]}
    if guard is not None:
        # the guard's verdict is per RANK (each engine hears its own shard): every rank's (arithmetic, level, fall-back) travels with
        # the line, and a disagreement -- a run priced as uniform that was not -- is flagged
        mine = {"rank": rank, "effective_precision": eff, "fallback": bool(rep.get("fallback")), "mlp_compensation": bool(rep.get("mlp_compensation")),
                "max_abs_dlogit": rep.get("max_abs_dlogit"), "checks_inside_timed_region": rep.get("checks", 0) - checks0}
        by_rank = [mine]
        if world > 1:
            by_rank = [None] * world
            torch.distributed.all_gather_object(by_rank, mine)
        guard["by_rank"] = by_rank
        guard["ranks_agree"] = len({(g["effective_precision"], g["fallback"], g["mlp_compensation"]) for g in by_rank}) == 1
    gather_ms = sorted(gather.spans_ms()) if world > 1 else []
    if world > 1:                            # the collective's result, checked once: this rank's rows of the gathered batch
        assert torch.equal(out[lo:hi], logits2[(a.steps - 1) & 1]), "all-gather returned other logits than this rank computed"
    lat = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
    dev_names = [f"rank {rank}: cuda:{device.index} {torch.cuda.get_device_name(device)}"]
    if world > 1:
        allnames = [None] * world
        torch.distributed.all_gather_object(allnames, dev_names[0])
        dev_names = allnames

    # PCIe-inclusive rate (reported beside `value`, never as it): the same batches start in page-locked HOST memory, are
    # staged as uint8 on the engine's copy stream (clm_stage_ids) one batch ahead of the forward that consumes them
    host_rate = None
    if world == 1:
        from chimeralm_amd._native import DT_U8
        hb = [b.cpu().pin_memory() for b in batches]
        k = max(2, min(a.steps, 6))
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        st = eng.stage_host_ids(hb[0].data_ptr(), DT_U8, hb[0].stride(0), hi - lo, L)
        for i in range(k):
            nxt = eng.stage_host_ids(hb[(i + 1) % n_data].data_ptr(), DT_U8, hb[(i + 1) % n_data].stride(0), hi - lo, L) \
                if i + 1 < k else -1
            eng.forward_staged(st, hi - lo, out=logits)
            st = nxt
        torch.cuda.synchronize(device)
        host_rate = a.batch * k / (time.perf_counter() - t1)
    # (CLM_TIMING_ONLY=1: developer runs of deliberately wrong timing-only library builds, tools/build_variant.sh)
    assert out.shape[0] == a.batch and (bool(torch.isfinite(out).all()) or os.environ.get("CLM_TIMING_ONLY") == "1")
    # latency as BASELINE.json words it: enqueue of a ready (device-resident) batch -> logits visible on the host, one batch in
    # flight (the throughput loop above keeps the queue full instead)
    host_logits = torch.empty((hi - lo, 2), dtype=torch.float32).pin_memory()
    host_lat = []
    for i in range(max(3, min(a.steps, 10))):
        torch.cuda.synchronize(device)
        t2 = time.perf_counter()
        eng.forward(batches[i % n_data], out=logits)
        host_logits.copy_(logits, non_blocking=True)
        torch.cuda.synchronize(device)
        host_lat.append((time.perf_counter() - t2) * 1e3)
    host_lat.sort()

    if rank == 0:
        total_ms = sum(ms for ms, _ in prof.values()) or 1.0
        config = {"workload": f"synthetic {a.bases}-bp reads, global batch {a.batch}, 1 forward per step",
                  "global_batch": a.batch, "tokens_per_read": L, "reads_per_gpu": hi - lo, "chunk_reads": a.chunk_reads,
                  "parallelism": f"read-sharded x{world}, logits all-gather" if world > 1 else "single GPU"}
        # (everything below `value` is reporting: an exception in it must not cost the line -- a NameError here once would have)
        try:
            es = 4 if eff in ("fp32", "fp16x3") else 2
            peak_key = eff                               # a guard that fell back runs -- and is priced against -- the fp32 MFMA
            dom = max(prof, key=lambda k: prof[k][0])
            ms, launches = prof[dom]
            tokens_per_launch = (hi - lo) * L * a.steps * (NLAYER if dom not in ("embed", "lnf_pool_score", "softmax_pool", "head_mlp") else 1) / max(1, launches)
            flops_per_token = STAGE_FLOPS_PER_TOKEN.get(dom, 0)
            fused_next = False
            if dom == "out_proj_ln2_mlp":
                # the tail kernel of block i also runs LN1 + in_proj of block i + 1 (every in_proj that has no launch of its own: 3 of a
                # forward's 4 in the 16-bit modes and in fused exact fp32, whose block 0 keeps its separate kernel) and, in the 16-bit
                # modes, ln_f + the pooling-score GEMM in the last block's launch: average over the launches
                n_in = prof.get("ln1_in_proj", (0.0, 0))[1]
                fused_in = max(launches - n_in - (launches // NLAYER if n_in == 0 else 0), 0)
                fused_score = launches // NLAYER if prof.get("lnf_pool_score", (0.0, 0))[1] == 0 else 0
                flops_per_token += (fused_in * STAGE_FLOPS_PER_TOKEN["ln1_in_proj"] + fused_score * STAGE_FLOPS_PER_TOKEN["lnf_pool_score"]) / max(1, launches)
                fused_next = fused_in > 0
            if dom in STAGE_FLOPS_PER_TOKEN:
                achieved = flops_per_token * tokens_per_launch / (ms / launches * 1e-3) / 1e12
                roof = {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": PEAK_TFLOPS[peak_key],
                        "unit": "TFLOP/s", "frac": achieved / PEAK_TFLOPS[peak_key], "traffic": None}
            else:
                achieved = stage_bytes_per_token(dom, es) * tokens_per_launch / (ms / launches * 1e-3) / 1e9
                roof = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": achieved / PEAK_HBM_GBS, "traffic": None}
            if roof["bound"] == "mfma" and eff != "fp32":
                # what the pool's MI355X sustains with every MFMA pipe busy (tools/micro/mfma_probe.cpp, profiles/r01_mfma_probe.txt:
                # 32 cycles per 32x32x16 MFMA per SIMD at the 1.55 GHz the chip holds under that load, against 2.4 GHz nominal)
                roof["peak_sustained_measured"] = SUSTAINED_MFMA16_TFLOPS
                roof["frac_of_sustained"] = achieved / SUSTAINED_MFMA16_TFLOPS
            roof["flops_per_token_per_launch"] = flops_per_token if dom in STAGE_FLOPS_PER_TOKEN else None
            roof["avg_launch_ms"] = ms / max(1, launches)
            roof["launches"] = launches
            tr = measured_traffic(dom, config, a.precision)
            if tr and "stale" in tr:
                roof["traffic_stale"] = tr["stale"]
            elif tr:
                roof["traffic"], roof["traffic_unit"], roof["traffic_source"] = tr["hbm_bytes_per_launch"], "bytes/launch", tr["source"]
                if tr.get("mfma_busy") and tr["mfma_busy"][0]:
                    # MFMA pipe occupancy from the same committed counter digest: SQ_VALU_MFMA_BUSY_CYCLES (summed over 1,024 SIMDs) over
                    # SIMDs x the kernel's duration in that counter pass x the shader clock the power log shows under this kernel
                    cyc, us = tr["mfma_busy"]
                    roof["mfma_busy_frac"] = cyc / (1024 * us * 1e-6 * SCLK_UNDER_TAIL_HZ)
                    roof["mfma_busy_note"] = f"SQ_VALU_MFMA_BUSY_CYCLES {cyc:.3g} per dispatch / (1,024 SIMDs x {us:.0f} us x {SCLK_UNDER_TAIL_HZ / 1e9:.2f} GHz), {tr['source']}"
                alg = stage_bytes_per_token(dom, es) or TAIL_BYTES_PER_TOKEN.get(dom, {}).get(es, 0.0)
                if fused_next:   # + z of the next block (3 of 4 launches); block 0 reads ids instead of its residual rows
                    alg += Z_ROWS * D * es * (NLAYER - 1) / NLAYER - D * 4 / NLAYER
                roof["algorithmic_hbm_bytes_per_launch"] = alg * tokens_per_launch
            if roof["bound"] == "mfma":
                # MFMA work actually issued (fp16c: two instructions per product) against the same peaks: pipe occupancy
                factor = MFMA_ISSUE_FACTOR[eff] + (MLP_LO_ISSUE if (eff == "fp16c" and rep.get("mlp_compensation")) else 0.0)
                roof["mfma_issue_factor"] = factor
                roof["issued_tflops"] = achieved * factor
                roof["issued_frac"] = roof["issued_tflops"] / PEAK_TFLOPS[peak_key]
        except Exception as exc:  # noqa: BLE001
            roof = {"bound": "mfma", "kernel": None, "achieved": None, "peak": None, "unit": "TFLOP/s", "frac": None,
                    "traffic": None, "error": f"roofline accounting failed: {exc!r}"}
        fp32_rate = None
        fp32_leg_error = None
        try:
            if world == 1 and eff != "fp32" and not a.no_fp32_leg:
                # the exact-fp32 engine on the same batches, so that the driver's run also times the mode that is bit-for-bit the
                # reference's arithmetic (a few steps: it is ~5x slower)
                e32 = Engine(device, precision="fp32", chunk_reads=a.chunk_reads)
                e32.load_state_dict(model.state_dict())
                e32.forward(batches[0], out=logits)
                torch.cuda.synchronize(device)
                k32 = max(2, min(3, a.steps))
                t3 = time.perf_counter()
                for i in range(k32):
                    e32.forward(batches[i % n_data], out=logits)
                torch.cuda.synchronize(device)
                fp32_rate = a.batch * k32 / (time.perf_counter() - t3)
                if guard is not None and guard.get("max_abs_dlogit_vs_exact_fp32") is None:
                    # a mode without a self-check of its own (fp16x3, fp16, bf16): the whole last batch against the exact engine, here
                    ref32 = e32.forward(batches[(k32 - 1) % n_data]).float().cpu()
                    mine = eng.forward(batches[(k32 - 1) % n_data]).float().cpu()
                    guard["max_abs_dlogit_vs_exact_fp32"] = float((ref32 - mine).abs().max())
                    guard["samples"] = [[f"the whole batch x {L} (outside the timed region)", guard["max_abs_dlogit_vs_exact_fp32"]]]
                e32.close()
        except Exception as exc:  # noqa: BLE001
            fp32_rate, fp32_leg_error = None, repr(exc)
        res = {
            "metric": f"reads/sec (whole node), {a.bases}-bp reads batch={a.batch}", "value": a.batch * a.steps / elapsed,
            "unit": "reads/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "ms_per_step_by_rank": rank_ms, "p50_batch_latency_ms": lat[len(lat) // 2],
            "p50_host_visible_latency_ms": host_lat[len(host_lat) // 2],   # this rank's shard: enqueue -> logits on the host
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": eff,
            "dtype_note": DTYPE_NOTE[eff] + ("" if eff == a.precision else f" [requested {a.precision}: the guard fell back]"),
            "data": "synthetic reads (seeded); seeded weights of the production architecture at a realistic logit scale (see `weights`)",
            "weights": weights_note,
            "product": "chimeralm_amd.hyena.HyenaDna with its guard on (HyenaDna.forward = guard + engine): `value` is what the module delivers",
            "config": config,
            "distributed": {"backend": torch.distributed.get_backend() if world > 1 else None,
                            "world_size": torch.distributed.get_world_size() if world > 1 else 1,
                            "devices": dev_names, "device_ids": dev_ids, "distinct_devices": len(set(dev_ids)) == len(dev_ids),
                            "rccl_version": ".".join(map(str, torch.cuda.nccl.version())) if world > 1 and backend == "nccl" else None,
                            "gather": "side stream, one step behind" if world > 1 else None,
                            # HIP-event time of the [B/N, 2] all-gather on its side stream, per step of the timed region (rank 0)
                            "gather_ms_p50": gather_ms[len(gather_ms) // 2] if gather_ms else None,
                            "gather_ms_max": gather_ms[-1] if gather_ms else None},
            "guard": guard,
            "fp32_exact_reads_per_s": fp32_rate,
            **({"fp32_leg_error": fp32_leg_error} if fp32_leg_error else {}),
            "pcie_inclusive_reads_per_s": host_rate,
            "dense_tflops_per_gpu": 6_423_040 * L * (hi - lo) * a.steps / elapsed / 1e12,
            "stage_ms_share": {k: round(v[0] / total_ms, 4) for k, v in prof.items() if v[1]},
            "roofline": roof,
        }
        if world == 1 and not a.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(a.bases)
            except Exception as exc:  # noqa: BLE001
                res["cpu_baseline"] = {"value": None, "unit": "reads/s", "cores": None, "kind": "port", "sample": None, "error": repr(exc)}
        print(json.dumps(res), flush=True)
    cdist.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
