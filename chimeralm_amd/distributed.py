"""Read-parallel multi-GPU support: one process per GPU, weights replicated, reads sharded, and ONE small collective
per batch -- an all-gather of the per-read logits (RCCL over xGMI when the backend is "nccl" on ROCm).

The reference has no collective on the predict path (Lightning DDP; each rank writes `{rank}_{batch}.txt`,
/root/reference/chimeralm/models/callbacks.py:134; batch_size // world_size per device, data/bam.py:142-146).
The gather exists so rank 0 can own the whole batch's logits (single writer / aggregated `predictions.txt`,
SURVEY.md section 8(e)); it moves B*2 floats, so it is latency- not bandwidth-bound.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_world() -> tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1-process defaults)."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_process_group(backend: str | None = None) -> tuple[int, int, int]:
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_bounds(n_reads: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous shard [lo, hi) of a global batch; the reference requires divisibility (bam.py:143-145)."""
    if n_reads % world != 0:
        raise RuntimeError(f"Batch size ({n_reads}) is not divisible by the number of devices ({world}).")
    per = n_reads // world
    return rank * per, (rank + 1) * per


def gather_logits(local_logits: torch.Tensor, world: int | None = None) -> torch.Tensor:
    """All-gather [B/G, C] -> [B, C] in rank order (contiguous shards, so this restores the global read order)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local_logits
    world = dist.get_world_size() if world is None else world
    out = torch.empty((world * local_logits.shape[0], local_logits.shape[1]), dtype=local_logits.dtype,
                      device=local_logits.device)
    dist.all_gather_into_tensor(out, local_logits.contiguous())
    return out


class LogitsGather:
    """The per-batch all-gather of `[B/G, C]` logits, kept OFF the compute stream ("on a side
    stream"): `submit` orders the collective after the forward that produced `local_logits` through an event and issues it on
    this object's own stream, so the next batch's kernels are enqueued and run while the 2 KiB exchange is in flight; the
    caller picks the result up one batch later (`done.synchronize()` / `wait`).  RCCL when the backend is "nccl"."""

    def __init__(self, device: torch.device, timed: bool = False):
        self.device = device
        self.side = torch.cuda.Stream(device) if device.type == "cuda" else None
        self.timed = timed                 # keep (start, end) events of every collective on the side stream (bench.py)
        self._spans: list = []

    def submit(self, local_logits: torch.Tensor):
        """-> (gathered [B, C] tensor, event that fires when it is complete)."""
        if not dist.is_initialized() or dist.get_world_size() == 1:
            return local_logits, None
        world = dist.get_world_size()
        out = torch.empty((world * local_logits.shape[0], local_logits.shape[1]), dtype=local_logits.dtype,
                          device=local_logits.device)
        if self.side is None:
            dist.all_gather_into_tensor(out, local_logits.contiguous())
            return out, None
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.side):
            self.side.wait_event(ready)
            if self.timed:
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record(self.side)
            dist.all_gather_into_tensor(out, local_logits.contiguous())
            if self.timed:
                t1.record(self.side)
                self._spans.append((t0, t1))
            done = torch.cuda.Event()
            done.record(self.side)
        local_logits.record_stream(self.side)
        out.record_stream(self.side)
        return out, done

    def wait(self):
        if self.side is not None:
            self.side.synchronize()

    def spans_ms(self, reset: bool = True) -> list[float]:
        """Device time of every timed collective so far (HIP events on the side stream); synchronises that stream."""
        self.wait()
        out = [a.elapsed_time(b) for a, b in self._spans]
        if reset:
            self._spans = []
        return out


def device_identity(device: torch.device) -> str:
    """What tells two GPUs of a node apart: PCI bus id (or the UUID) of the device this rank computes on."""
    p = torch.cuda.get_device_properties(device)
    for attr in ("pci_bus_id", "uuid"):
        v = getattr(p, attr, None)
        if v not in (None, ""):
            dom = getattr(p, "pci_domain_id", 0)
            dev = getattr(p, "pci_device_id", 0)
            return f"{dom:04x}:{int(v):02x}:{dev:02x}" if attr == "pci_bus_id" and isinstance(v, int) else str(v)
    return f"cuda:{device.index}"


def assert_distinct_devices(device: torch.device) -> list[str]:
    """Every rank's device identity, gathered; under "nccl" (RCCL) two ranks on ONE device cannot form a communicator that
    works -- fail fast with a message that says which ranks collide instead of hanging in the first collective."""
    me = device_identity(device)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [me]
    ids: list = [None] * dist.get_world_size()
    dist.all_gather_object(ids, me)
    if dist.get_backend() == "nccl" and len(set(ids)) != len(ids):
        dup = {i: [r for r, v in enumerate(ids) if v == i] for i in set(ids) if ids.count(i) > 1}
        raise RuntimeError(f"ranks share a GPU under the nccl (RCCL) backend: {dup}; launch one process per GPU "
                           "(torch.distributed.run --nproc-per-node N sets LOCAL_RANK = device index) or rehearse with "
                           "CLM_DIST_BACKEND=gloo")
    return ids


def free_port() -> int:
    """A TCP port that is free right now on 127.0.0.1 (rendezvous of the ranks `predict -g N` spawns)."""
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
