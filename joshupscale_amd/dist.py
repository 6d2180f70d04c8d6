"""Multi-GPU plumbing: replicas only.

One recurrent stream cannot be sharded (frame t needs the output of frame t-1,
reference scripts/training/models.py:752-764, 821-823), so N GPUs run N
independent streams, one process per GPU.  The only collective is a start-up
broadcast of the model container from rank 0 so that only rank 0 touches the file
system.  There is no per-frame communication.

On GPUs the broadcast is the C layer's own (``ju_comm_*`` in libJoshUpscale.so:
``ncclBroadcast`` of uint8 through librccl over xGMI, csrc/comm.cpp);
``torch.distributed`` only carries the 128-byte communicator id from rank 0 to the other
ranks (it is the launcher's rendezvous).  The pure ``torch.distributed`` broadcast below
is the test double: the CPU tests run it on gloo with world size 2.
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


# how the last broadcast_model() moved the bytes (bench.py reports it)
LAST_BROADCAST = {"how": "none", "ranks_seen": 1, "bytes": 0}
_COMM = None  # the C layer's communicator, kept for max_over_ranks()


def native_comm():
    return _COMM


def broadcast_model_native(blob: Optional[bytes], device: torch.device, src: int = 0) -> bytes:
    """The product path on GPUs: ``ju_comm_broadcast`` (RCCL in the C layer).  The process
    group is used for exactly two small object broadcasts: the communicator id and the
    payload length."""
    global _COMM
    from . import runtime as R
    rank, world = dist.get_rank(), dist.get_world_size()
    meta = [R.comm_unique_id(), len(blob)] if rank == src else [None, None]
    dist.broadcast_object_list(meta, src=src)
    comm = R.Comm(meta[0], rank, world, device.index if device.index is not None else 0)
    out = comm.broadcast(blob if rank == src else None, int(meta[1]), root=src)
    if _COMM is not None:  # a repeated broadcast: the previous communicator is not leaked
        _COMM.close()
    _COMM = comm
    LAST_BROADCAST.update(how="rccl (ju_comm_broadcast, C layer)", ranks_seen=comm.count(), bytes=int(meta[1]))
    return out


def broadcast_model(blob: Optional[bytes], device: torch.device, src: int = 0,
                    native: Optional[bool] = None) -> bytes:
    """Broadcast the model container bytes from ``src`` to every rank.

    ``native`` (default: on for the "nccl" backend): the C layer's RCCL broadcast.
    Otherwise two torch.distributed collectives: the length (one int64), then the payload
    as uint8 on ``device`` -- the test double used on gloo."""
    if not dist.is_initialized():
        if blob is None:
            raise ValueError("single process needs the model bytes")
        return blob
    if native is None:
        native = dist.get_backend() == "nccl"
    if native:
        return broadcast_model_native(blob, device, src)
    rank = dist.get_rank()
    n = torch.tensor([len(blob) if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src=src)
    size = int(n.item())
    if rank == src:
        payload = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
    else:
        payload = torch.empty(size, dtype=torch.uint8, device=device)
    dist.broadcast(payload, src=src)
    LAST_BROADCAST.update(how=f"torch.distributed ({dist.get_backend()})", ranks_seen=dist.get_world_size(),
                          bytes=size)
    return payload.cpu().numpy().tobytes()


def assign_streams(num_streams: int, world_size: int) -> list:
    """Stream i runs on GPU i mod N."""
    return [[s for s in range(num_streams) if s % world_size == r]
            for r in range(world_size)]


def max_over_ranks(value: float, device: torch.device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    if _COMM is not None:  # the C layer's communicator (RCCL all-reduce of one double)
        return _COMM.allreduce_max(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(value: float, device: torch.device) -> list:
    """One float per rank, in rank order, on every rank (the bench's per-rank frame rates:
    outside the timed region, so the launcher's process group carries it)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(value)]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, float(value))
    return [float(v) for v in out]


def barrier() -> None:
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
