"""CPU, world_size 2 over gloo: the N > 1 path -- contiguous read shards, all-gather of the per-read logits restoring
the global order, per-rank prediction files, and the strong-scaling bookkeeping bench.py uses."""
from __future__ import annotations

import os
import socket
from pathlib import Path

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, tmp: str):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from types import SimpleNamespace

    from chimeralm_amd import distributed as cd
    from chimeralm_amd.callbacks import PredictionWriter
    from chimeralm_amd.tokenizer import pack_read_name

    r, lr, w = cd.init_process_group("gloo")
    assert (r, w) == (rank, world)
    B = 8
    lo, hi = cd.shard_bounds(B, r, w)
    assert hi - lo == B // world
    full = torch.arange(B * 2, dtype=torch.float32).reshape(B, 2)          # stand-in for per-read logits
    gathered = cd.gather_logits(full[lo:hi].clone())
    assert torch.equal(gathered, full)                                      # global read order restored on every rank
    ids = torch.tensor([pack_read_name(f"read{i}") for i in range(lo, hi)], dtype=torch.int64).to(torch.int8)
    PredictionWriter(Path(tmp), "batch").write_on_batch_end(SimpleNamespace(global_rank=r), None,
                                                            (full[lo:hi], None), None, {"id": ids}, 0, 0)
    cd.barrier()
    try:
        cd.shard_bounds(7, r, w)
    except RuntimeError as e:
        assert "not divisible" in str(e)
    else:
        raise AssertionError("indivisible batch accepted")
    t = torch.tensor([float(r + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                                # max-over-ranks timing, as in bench.py
    assert t.item() == world
    dist.destroy_process_group()


def test_two_ranks_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    files = sorted(p.name for p in tmp_path.glob("*.txt"))
    assert files == ["0_0.txt", "1_0.txt"]                                  # {rank}_{batch}.txt, callbacks.py:134
    names = [ln.split("\t")[0] for f in files for ln in (tmp_path / f).read_text().splitlines()]
    assert names == [f"read{i}" for i in range(8)]
