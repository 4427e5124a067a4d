"""The N>1 path on CPU: world_size 2, gloo.  What the multi-GPU bench does at load
time and per step, minus the device: rank 0 quantizes + packs, the blob is
broadcast with the same helper the RCCL path uses, every rank validates it,
scores its own contiguous frame shard (the oracle stands in for the kernels
here -- this test is about the sharding/broadcast protocol), and the gathered
result equals the single-process answer."""
import hashlib
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, golden
from fast_dnn_amd import dist as fd


def test_frame_shards_cover_and_balance():
    for n in (0, 1, 7, 100, 1000, 1_000_003):
        for w in (1, 2, 4, 8):
            sh = fd.frame_shards(n, w)
            assert sh[0][0] == 0 and sh[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(sh, sh[1:]))
            sizes = [b - a for a, b in sh]
            assert max(sizes) - min(sizes) <= 1
    assert fd.frame_shards(1_000_000, 8)[3] == (375_000, 500_000)  # BASELINE configs[4]: 125k frames per GPU


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, model_path, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    from fast_dnn_amd import api, dist as fd, formats as F
    from oracle.oracle import Oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    blob = None
    if rank == 0:
        blob = torch.from_numpy(api.HostModel(model_path).blob())
    blob = fd.broadcast_blob(blob, rank, world, torch.device("cpu"))
    info = api.host_blob_check(blob.numpy())
    digest = hashlib.sha256(blob.numpy().tobytes()).hexdigest()
    x = F.synth_features(101, seed=4)
    lo, hi = fd.frame_shards(x.shape[0], world)[rank]
    part = Oracle(model_path).calculate(x[lo:hi])
    parts = [None] * world
    dist.all_gather_object(parts, (lo, hi, part, digest, info))
    if rank == 0:
        np.savez(os.path.join(out_dir, "gathered.npz"), probs=np.concatenate([p[2] for p in parts]),
                 ranges=np.array([[p[0], p[1]] for p in parts]))
        assert len({p[3] for p in parts}) == 1, "ranks hold different weight blobs"
        assert all(p[4] == parts[0][4] for p in parts)
    dist.barrier()
    dist.destroy_process_group()


def test_world2_broadcast_and_shard(tiny_model_path, tmp_path):
    import torch.multiprocessing as mp

    from fast_dnn_amd import formats as F
    from oracle.oracle import Oracle

    port = _free_port()
    mp.spawn(_worker, args=(2, port, tiny_model_path, str(tmp_path)), nprocs=2, join=True)
    g = np.load(str(tmp_path / "gathered.npz"))
    assert g["ranges"].tolist() == [[0, 51], [51, 101]]
    want = Oracle(tiny_model_path).calculate(F.synth_features(101, seed=4))
    assert (g["probs"] == want).all()


def test_blob_check_rejects_garbage(tiny_model_path):
    from fast_dnn_amd import api

    blob = api.HostModel(tiny_model_path).blob()
    assert api.host_blob_check(blob) == {"input_dim": 432, "hidden_dim": 64, "output_dim": 100, "n_affine": 4}
    bad = blob.copy()
    bad[0] ^= 0xFF
    with pytest.raises(api.FdnnError):
        api.host_blob_check(bad)
    with pytest.raises(api.FdnnError):
        api.host_blob_check(blob[:-256])


def test_c_abi_shards_equal_the_python_shards():
    """fdnn_group_shard (host C++, one process / N devices) and dist.frame_shards (one process per
    GPU) cut a batch the same way: contiguous, sizes differing by at most one, covering [0, n)."""
    from fast_dnn_amd import api
    from fast_dnn_amd.dist import frame_shards

    for n in (0, 1, 7, 8, 9, 1000, 10000, 1_000_000):
        for world in (1, 2, 3, 8):
            got = [api.group_shard(n, world, r) for r in range(world)]
            assert got == frame_shards(n, world), (n, world)
            assert got[0][0] == 0 and got[-1][1] == n
