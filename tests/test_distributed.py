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


def _gather_worker(rank: int, world: int, port: int, tmp: str):
    """Ranks with DIFFERENT numbers of batches through the gathering predict protocol: nobody may hang, every rank sees the
    same gathered rounds, padding rows are marked invalid."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from types import SimpleNamespace

    from chimeralm_amd import distributed as cd
    from chimeralm_amd.callbacks import PredictionWriter
    from chimeralm_amd.predict import _Deferred, _drain_gather
    from chimeralm_amd.tokenizer import pack_read_name

    cd.init_process_group("gloo")
    rows, device = 3, torch.device("cpu")
    gatherer = cd.LogitsGather(device)
    my_batches = [3, 3, 2] if rank == 0 else [3]                            # rank 1 runs out two rounds earlier; one short batch
    seen = []
    writer = PredictionWriter(Path(tmp), "batch")
    trainer = SimpleNamespace(global_rank=rank)
    pending, batch_idx = None, 0
    for n in my_batches:
        logits = torch.full((n, 2), float(10 * rank + batch_idx))
        ids = torch.tensor([pack_read_name(f"r{rank}b{batch_idx}i{i}") for i in range(n)], dtype=torch.int64).to(torch.int8)
        now = _Deferred(logits, torch.full((n,), -1), {"id": ids}, batch_idx, gatherer, rows)
        if pending is not None:
            pending.flush(writer, trainer, None, lambda b, g: seen.append((b, g.clone())))
        pending = now
        batch_idx += 1
    _drain_gather(pending, gatherer, rows, device, batch_idx, writer, trainer, None, lambda b, g: seen.append((b, g.clone())))
    assert [b for b, _ in seen] == [0, 1, 2], seen                          # three rounds had reads somewhere, on BOTH ranks
    g0, g1, g2 = (g for _, g in seen)
    assert g0.shape == (world * rows, 3)
    assert g0[:, 2].tolist() == [1, 1, 1, 1, 1, 1] and g1[:, 2].tolist() == [1, 1, 1, 0, 0, 0] and g2[:, 2].tolist() == [1, 1, 0, 0, 0, 0]
    assert g0[:3, 0].tolist() == [0.0] * 3 and g0[3:, 0].tolist() == [10.0] * 3 and g2[:2, 0].tolist() == [2.0, 2.0]
    cd.barrier()
    dist.destroy_process_group()


def test_gathering_predict_protocol_with_unequal_batch_counts(tmp_path):
    world = 2
    mp.spawn(_gather_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    files = sorted(p.name for p in tmp_path.glob("*.txt"))
    assert files == ["0_0.txt", "0_1.txt", "0_2.txt", "1_0.txt"]            # empty rounds write no file
    assert (tmp_path / "0_2.txt").read_text().count("\n") == 2
