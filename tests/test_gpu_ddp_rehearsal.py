"""Data-parallel training rehearsed on one GPU (-m gpu): two real `Trainer` ranks (processes) under a gloo group share the box's GPU, run two
epochs of `fit()` — per-step flat-gradient all-reduce, rank-0-only validation + checkpoint writing, final barrier — and must end with bit-identical
parameters and EMA; a single-rank run of the same schedule on rank 0's shard alone must differ (the other rank's gradient took part).
The production collective is RCCL over xGMI, one rank per GPU (`torch.distributed` backend "nccl"): same code path, `allreduce_gradients`."""
import os
import subprocess
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu

WORKER = Path(__file__).with_name("ddp_worker.py")


def _launch(out, world, port):
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(WORKER), str(out)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors="replace"))
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{logs[r][-3000:]}"


def test_two_trainer_ranks_stay_identical_and_both_contribute(tmp_path):
    _launch(tmp_path, 2, 29531)
    a, b = (torch.load(tmp_path / f"rank{r}_of2.pt", weights_only=True) for r in (0, 1))
    assert a["steps"] == b["steps"] == 4
    assert torch.equal(a["p"], b["p"]) and torch.equal(a["ema"], b["ema"]), "ranks diverged: the gradient exchange is not symmetric"
    assert bool(torch.isfinite(a["p"]).all())
    run = tmp_path / "w2"
    for f in ("weights/best.pt", "weights/last.pt", "results.csv", "args.yaml"):  # written by rank 0 only
        assert (run / f).exists() and (run / f).stat().st_size > 0
    assert len((run / "results.csv").read_text().strip().splitlines()) == 3
    _launch(tmp_path, 1, 29532)
    solo = torch.load(tmp_path / "rank0_of1.pt", weights_only=True)
    assert float((solo["p"] - a["p"]).abs().max()) > 0
