"""GPU (one MI355X): rehearsal of the multi-GPU bench launch.  The driver's exact command line for N = 2
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ...`)
runs with both ranks on the one GPU of the box over gloo (RCCL refuses two ranks per device; `CLM_DIST_BACKEND=gloo` is the only
difference from the real run): read shards, the all-gather of the logits, max-over-ranks timing and the JSON contract."""
from __future__ import annotations

import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_two_ranks_on_one_gpu(built_lib):
    env = dict(os.environ, CLM_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(REPO / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--batch", "8", "--bases", "2100", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                       # rank 0 prints ONE JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1
    assert d["value"] > 0 and d["unit"] == "reads/s" and d["scaling"] == "strong" and d["higher_is_better"] is True
    assert d["config"]["reads_per_gpu"] == 4 and "x2" in d["config"]["parallelism"]
    assert abs(d["value"] - 8 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]   # whole-job reads / max-over-ranks time
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(d["roofline"])
    assert d["distributed"]["backend"] == "gloo" and d["distributed"]["world_size"] == 2 and len(d["distributed"]["devices"]) == 2


def test_bare_bench_gpus_2_launches_its_own_ranks(built_lib):
    """`python bench.py --gpus 2` with no launcher around it starts its two ranks itself (torch.distributed.run
    on a free port, before anything touches HIP), relays rank 0's ONE JSON line and exits with the children's status."""
    env = dict(os.environ, CLM_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8", "--bases", "2100",
           "--no-cpu-baseline", "--no-fp32-leg"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["distributed"]["world_size"] == 2 and d["value"] > 0
    assert len(d["guard"]["by_rank"]) == 2 and d["guard"]["ranks_agree"] is True   # every rank's verdict travels with the line
    bad = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "8",
                          "--bases", "2100", "--precision", "no-such-mode", "--no-cpu-baseline"], capture_output=True, text=True,
                         timeout=600, env=env, cwd=REPO)
    assert bad.returncode != 0                                 # the children's failure is this process's exit status


def test_predict_two_ranks_on_one_gpu_union_equals_single_rank(tmp_path, golden_dir, built_lib):
    """`python -m chimeralm_amd predict BAM -g 2` (one process per rank through torchrun, a free rendezvous port) with both
    ranks on this box's one GPU over gloo: rank r classifies reads r, r+2, ... (the non-shuffling distributed sampler of the
    reference's Lightning run) and writes `{rank}_{batch}.txt`; `--gather-logits` exercises the side-stream all-gather and its
    drain protocol for real.  Every read must appear exactly once in the union of the files, and a single-rank run fed the same
    per-rank batches must give the same labels (batches are compared like for like: logits depend on the batch's padding)."""
    import shutil

    from safetensors.torch import save_file

    from oracle import hyena_oracle as ho

    sd = ho.make_state_dict(0, head_scale=3.0)
    wdir = tmp_path / "weights"
    wdir.mkdir()
    save_file({k: v.contiguous() for k, v in sd.items() if not (k.endswith(".3.freq") or k.endswith(".5.freq"))},
              str(wdir / "model.safetensors"))
    bam = tmp_path / "reads.bam"
    shutil.copyfile(golden_dir / "test_chimric_reads.bam", bam)
    env = dict(os.environ, PYTHONPATH=str(REPO), CLM_DIST_BACKEND="gloo", CLM_RANKS_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("MASTER_PORT", None)
    out2 = tmp_path / "two"
    r = subprocess.run([sys.executable, "-m", "chimeralm_amd", "predict", str(bam), "-g", "2", "-b", "24", "-o", str(out2),
                        "--weights", str(wdir), "--precision", "fp32", "--gather-logits"],
                       capture_output=True, text=True, env=env, cwd=str(REPO), timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    files = sorted(p.name for p in out2.glob("*_*.txt"))
    # 100 selected reads, 12 per rank and batch: 50 reads per rank = 5 batches each (12, 12, 12, 12, 2)
    assert files == sorted(f"{rk}_{b}.txt" for rk in (0, 1) for b in range(5)), files
    two = {}
    for rk in (0, 1):
        for b in range(5):
            for ln in (out2 / f"{rk}_{b}.txt").read_text().splitlines():
                name, label = ln.split("\t")
                assert name not in two, f"{name} classified twice"
                two[name] = (rk, b, int(label))
    assert len(two) == 100
    # gathered logits: rank 0 wrote one row per read, rows of rank r in [r*12, (r+1)*12)
    rows = [ln.split("\t") for ln in (out2 / "logits.tsv").read_text().splitlines()]
    assert len(rows) == 100 and {int(x[1]) for x in rows} == {0, 1}
    # same batches through one process: the Python data module with world_size 2 yields exactly the per-rank batches
    import torch

    from chimeralm_amd import bam as bam_mod, lm, tokenizer as T
    from oracle import data_oracle as do

    model = lm.ChimeraLM.new(precision="fp32")
    model.load_state_dict(sd, strict=True)
    tok = T.load_tokenizer_from_hyena_model("hyenadna-small-32k-seqlen")
    for rk in (0, 1):
        dm = bam_mod.BamDataModule(tokenizer=tok, predict_data_path=bam, batch_size=24)
        dm.setup("predict", world_size=2, rank=rk)
        for b, batch in enumerate(dm.predict_dataloader()):
            logits, _ = model.predict_step({**batch, "input_ids": batch["input_ids"].cuda()}, b)
            want = "".join(do.prediction_lines(logits.cpu().numpy(), batch["id"].numpy()))
            assert (out2 / f"{rk}_{b}.txt").read_text() == want, f"rank {rk} batch {b}"
            mine = [x for x in rows if int(x[0]) == b and int(x[1]) == rk]
            got = torch.tensor([[float(x[3]), float(x[4])] for x in mine])
            assert got.shape == logits.shape and (got - logits.cpu()).abs().max() < 1e-5
