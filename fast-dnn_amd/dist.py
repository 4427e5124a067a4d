"""Multi-GPU use of the scorer: one process per GPU, frames sharded, weights replicated.

Frames are independent (no recurrence, no cross-frame state in ``calculate``), so
the path shards by contiguous frame ranges with NO collective in steady state.
The only communication is at load time: rank 0 parses + quantizes the ``.bin``
once and the packed weight blob (~45 MB for 7x2048 -> 8000) is broadcast over
RCCL/xGMI, which also guarantees bit-identical weights on every rank.  The
reference has no counterpart (single process); its concurrency model is
caller-side threads over independent utterances (MultiThreadedStressTest.java).
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np


def frame_shards(n_frames: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous [start, stop) frame ranges, sizes differing by at most one."""
    base, extra = divmod(n_frames, world)
    out, start = [], 0
    for r in range(world):
        size = base + (1 if r < extra else 0)
        out.append((start, start + size))
        start += size
    return out


def broadcast_blob(blob, rank: int, world: int, device, src: int = 0, force: bool = False):
    """Broadcast a uint8 tensor whose length only ``src`` knows.  ``blob`` is the
    payload tensor on ``src`` (any value elsewhere); returns the payload on every
    rank.  Works on whatever backend the default process group uses (nccl = RCCL
    on GPUs, gloo in the CPU tests)."""
    import torch
    import torch.distributed as dist

    if world == 1 and not force:  # (force: run the collectives on a one-rank group -- proves the backend loads and moves device tensors)
        return blob
    size = torch.zeros(1, dtype=torch.int64, device=device)
    if rank == src:
        size[0] = blob.numel()
    dist.broadcast(size, src=src)
    if rank != src:
        blob = torch.empty(int(size.item()), dtype=torch.uint8, device=device)
    dist.broadcast(blob, src=src)
    return blob


def load_replicated(model_path: str, device_index: int, rank: int, world: int, host_broadcast: bool = False, force_collective: bool = False):
    """QuantizedDnn.loadFromFile on rank 0 + RCCL broadcast of the packed weights.

    ``host_broadcast``: the default process group is gloo (several ranks sharing one device, where
    RCCL refuses to start): the blob makes the trip through host memory instead."""
    import torch

    from . import api

    dev = torch.device("cuda", device_index)
    if world == 1 and not force_collective:
        return api.QuantizedDnn.loadFromFile(model_path, device=device_index)
    blob = None
    dnn = None
    if rank == 0:
        dnn = api.QuantizedDnn.loadFromFile(model_path, device=device_index)
        nbytes = dnn.blobSize()
        blob = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        dnn.exportBlob(blob.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    if host_broadcast:
        blob = broadcast_blob(blob.cpu() if rank == 0 else None, rank, world, torch.device("cpu")).to(dev)
    else:
        blob = broadcast_blob(blob, rank, world, dev, force=force_collective)
    if rank != 0 or force_collective:
        # (force_collective on one rank: the model that scores is the one IMPORTED from the broadcast buffer, as on ranks > 0)
        torch.cuda.synchronize()
        if dnn is not None:
            dnn.delete()
        dnn = api.QuantizedDnn.fromDeviceBlob(blob.data_ptr(), blob.numel(), device_index)
    return dnn


def calculate_sharded(dnn, frames: np.ndarray, rank: int, world: int) -> Tuple[np.ndarray, Tuple[int, int]]:
    """Each rank scores its own contiguous frame range; no data-path collective."""
    lo, hi = frame_shards(frames.shape[0], world)[rank]
    return dnn.calculate(frames[lo:hi]), (lo, hi)
