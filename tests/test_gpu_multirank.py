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
