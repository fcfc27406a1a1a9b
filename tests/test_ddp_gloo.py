"""N>1 path on CPU: 2 gloo ranks.  Slice sharding is disjoint and complete; the flat-gradient all-reduce sums over ranks
(what RCCL does on the GPUs).  The rendezvous uses 127.0.0.1."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from mslesseg_amd.train import allreduce_gradients, shard_indices

    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    allreduce_gradients(g)
    mine = shard_indices(53, 3, 0, rank, world)
    torch.save({"g": g, "mine": torch.from_numpy(mine)}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_flat_allreduce(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"r{k}.pt", weights_only=True) for k in range(world)]
    want = torch.arange(1000, dtype=torch.float32) * 3  # (1 + 2) * arange
    assert torch.equal(r[0]["g"], want) and torch.equal(r[1]["g"], want)
    a, b = set(r[0]["mine"].tolist()), set(r[1]["mine"].tolist())
    assert not (a & b) and a | b == set(range(53)) and abs(len(a) - len(b)) <= 1
