"""Multi-GPU plumbing: replicas only.

One recurrent stream cannot be sharded (frame t needs the output of frame t-1,
reference scripts/training/models.py:752-764, 821-823), so N GPUs run N
independent streams, one process per GPU.  The only collective is a start-up
broadcast of the model container from rank 0 (RCCL over xGMI when the backend is
"nccl"; gloo in the CPU tests) so that only rank 0 touches the file system.
There is no per-frame communication.
"""

from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend: str) -> None:
    if dist.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    dist.init_process_group(backend=backend)


def broadcast_model(blob: Optional[bytes], device: torch.device, src: int = 0) -> bytes:
    """Broadcast the model container bytes from ``src`` to every rank.

    Two collectives: the length (one int64), then the payload as uint8 on
    ``device`` (for "nccl" a GPU tensor, i.e. RCCL moves it over xGMI)."""
    if not dist.is_initialized():
        if blob is None:
            raise ValueError("single process needs the model bytes")
        return blob
    rank = dist.get_rank()
    n = torch.tensor([len(blob) if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src=src)
    size = int(n.item())
    if rank == src:
        payload = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
    else:
        payload = torch.empty(size, dtype=torch.uint8, device=device)
    dist.broadcast(payload, src=src)
    return payload.cpu().numpy().tobytes()


def assign_streams(num_streams: int, world_size: int) -> list:
    """Stream i runs on GPU i mod N."""
    return [[s for s in range(num_streams) if s % world_size == r]
            for r in range(world_size)]


def max_over_ranks(value: float, device: torch.device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier() -> None:
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
