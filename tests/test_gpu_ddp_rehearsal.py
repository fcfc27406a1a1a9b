"""Data-parallel training rehearsed on one GPU (-m gpu): two real `Trainer` ranks (processes) under a gloo group share the box's GPU, run two
epochs of `fit()` — per-step flat-gradient all-reduce, validation sharded over the ranks (one all-gather of the match records), checkpoint files
from rank 0's writer thread, final barrier — and must end with bit-identical parameters and EMA; the sharded validation must give the metrics
and val losses one rank computes alone on the same weights; a single-rank run of the same schedule on rank 0's shard alone must differ (the other rank's gradient took part).
The production collective is RCCL over xGMI, one rank per GPU (`torch.distributed` backend "nccl"): same code path, `allreduce_gradients`."""
import os
import subprocess
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu

WORKER = Path(__file__).with_name("ddp_worker.py")


def _launch(out, world, port, mode=None):
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(WORKER), str(out)] + ([mode] if mode else []), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
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
    for r in (a, b):
        sh, si = r["val_sharded"], r["val_single"]
        assert torch.allclose(sh["losses"], si["losses"], rtol=2e-4, atol=1e-6), (sh["losses"], si["losses"])  # the loss op sums with fp32 atomics: 1e-5 run to run
        for k, v in si["metrics"].items():
            assert abs(sh["metrics"][k] - v) <= 1e-6, (k, sh["metrics"][k], v)
        assert r["model_device"] == r["trainer_device"] == r["predict_device"]
    assert a["val_sharded"]["metrics"] == b["val_sharded"]["metrics"]  # every rank rebuilds the same merged lists
    _launch(tmp_path, 1, 29532)
    solo = torch.load(tmp_path / "rank0_of1.pt", weights_only=True)
    assert float((solo["p"] - a["p"]).abs().max()) > 0


def test_data_parallel_step_equals_one_rank_accumulating_the_union_batch(tmp_path):
    """Equivalence, not only symmetry: two ranks x 4 slices per step against ONE rank that takes the same 8 slices per step as two micro-batches of
    4 with gradient accumulation (the ranks' batches, re-created through the trainer's own dealing: tests/ddp_worker.py union_run).  BatchNorm
    statistics are per rank in the data-parallel run and per micro-batch in the other — the same numbers — so the summed gradient, the AdamW step and
    the EMA must agree to fp32 rounding (the loss op and the weight-gradient partial sums add in a different order)."""
    _launch(tmp_path, 2, 29541, "union")
    _launch(tmp_path, 1, 29542, "union")
    dp = torch.load(tmp_path / "union_rank0_of2.pt", weights_only=True)
    one = torch.load(tmp_path / "union_rank0_of1.pt", weights_only=True)
    assert dp["steps"] == one["steps"] == 2
    assert torch.equal(dp["p0"], one["p0"])
    moved = float((dp["p"] - dp["p0"]).abs().max())
    assert moved > 1e-3, "the two steps must have changed the parameters for the comparison to mean anything"
    err = float((dp["p"] - one["p"]).abs().max())
    print(f"union-batch equivalence: max |dp - accumulate| {err:.3e} for a largest parameter change of {moved:.3e}")
    assert err <= 2e-3 * moved, (err, moved)
    assert float((dp["ema"] - one["ema"]).abs().max()) <= 2e-3 * moved
