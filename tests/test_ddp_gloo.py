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

    from mslesseg_amd.train import allreduce_gradients, collective_selfcheck, shard_indices

    dist.init_process_group("gloo", rank=rank, world_size=world)
    check = collective_selfcheck("cpu", n=4096)  # the start-up self-check every data-parallel Trainer runs on its real backend (here: gloo)
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    allreduce_gradients(g)
    mine = shard_indices(53, 3, 0, rank, world)
    torch.save({"g": g, "mine": torch.from_numpy(mine), "check_ok": torch.tensor(check["ok"] and check["world"] == world and check["rank"] == rank)},
               os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_flat_allreduce(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"r{k}.pt", weights_only=True) for k in range(world)]
    want = torch.arange(1000, dtype=torch.float32) * 3  # (1 + 2) * arange
    assert torch.equal(r[0]["g"], want) and torch.equal(r[1]["g"], want)
    assert bool(r[0]["check_ok"]) and bool(r[1]["check_ok"])
    a, b = set(r[0]["mine"].tolist()), set(r[1]["mine"].tolist())
    assert not (a & b) and a | b == set(range(53)) and abs(len(a) - len(b)) <= 1


def test_gradient_buckets_partition_the_flat_buffer():
    """The two all-reduce buckets of the data-parallel step (head + neck = layers model.11.., backbone = the rest; trainprog.ParamStore.bucket_ranges) are
    disjoint ranges that cover the flat gradient buffer exactly, and every parameter lies entirely in the bucket of its layer."""
    from mslesseg_amd import params

    class Store:  # the address book of ParamStore without its device buffers
        pass

    from mslesseg_amd.trainprog import ParamStore, _align4
    import math

    st = Store()
    st.specs = params.param_specs("n", 1)
    st.entries = {}
    off = 0
    for name, s in st.specs.items():
        shp = (s["cin"], 2, 2, s["cout"]) if s["kind"] == "convT" else ((3, 3, 3, s["cout"]) if s.get("stem") else ((3, 3, s["cout"]) if s["groups"] > 1 else (s["cout"], s["k"], s["k"], s["cin"])))
        st.entries[name + ".w"] = (off, shp)
        off = _align4(off + math.prod(shp))
    st.n_decay = off
    for name, s in st.specs.items():
        for t in (("gamma", "beta") if (s["kind"] == "conv" and s["bn"]) else ("bias",)):
            st.entries[f"{name}.{t}"] = (off, (s["cout"],))
            off = _align4(off + s["cout"])
    st.n = off
    head, back = ParamStore.bucket_ranges(st, 11)
    spans = sorted(head + back)
    assert spans[0][0] == 0 and spans[-1][1] == st.n and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    for key, (o, shp) in st.entries.items():
        mine = head if int(key.split(".")[1]) >= 11 else back
        assert any(lo <= o and o + math.prod(shp) <= hi for lo, hi in mine), key


def _shard_worker(rank, world, port, root, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from mslesseg_amd import data as D

    dist.init_process_group("gloo", rank=rank, world_size=world)
    ds = D.SegDataset(root, 64, shard=(rank, world))
    decoded = sum(r is not None for r in ds.raw)
    ds.exchange()
    torch.save({"decoded": decoded, "imgs": [torch.from_numpy(ds.raw[i][0].copy()) for i in range(len(ds.raw))],
                "labels": [[torch.from_numpy(p) for _, p in ds.raw[i][1]] for i in range(len(ds.raw))]}, os.path.join(out_dir, f"s{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_dataset_load_equals_the_full_load(tmp_path):
    """Data-parallel start-up: every rank decodes 1 / world of the fold's PNG files and one all-gather completes the list (data.SegDataset(shard=...).exchange())
    — the same slices and labels, in the same order, as one process loading everything."""
    from mslesseg_amd import data as D
    from mslesseg_amd import pngio

    rng = np.random.default_rng(0)
    root = tmp_path / "ds"
    (root / "images").mkdir(parents=True)
    (root / "labels").mkdir()
    for k in range(7):
        h, w = [(40, 33), (33, 40), (40, 40)][k % 3]
        pngio.write_png(root / "images" / f"s{k:02d}.png", rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8))
        (root / "labels" / f"s{k:02d}.txt").write_text("".join(f"0 {0.1 + 0.05 * j:.3f} 0.2 0.6 0.25 0.5 {0.7 + 0.02 * j:.3f}\n" for j in range(k % 3 + 1)))
    full = D.SegDataset(root, 64)
    world, port = 2, _free_port()
    out = tmp_path / "out"
    out.mkdir()
    mp.spawn(_shard_worker, args=(world, port, str(root), str(out)), nprocs=world, join=True)
    r = [torch.load(out / f"s{k}.pt", weights_only=True) for k in range(world)]
    assert r[0]["decoded"] == 4 and r[1]["decoded"] == 3
    for rk in r:
        assert len(rk["imgs"]) == len(full.raw) == 7
        for i, (img, inst) in enumerate(full.raw):
            assert np.array_equal(rk["imgs"][i].numpy(), img)
            assert len(rk["labels"][i]) == len(inst) and all(np.array_equal(a.numpy(), p) for a, (_, p) in zip(rk["labels"][i], inst))
