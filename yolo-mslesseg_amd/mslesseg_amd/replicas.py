"""Zero-communication replica mode: the reference trains one model per fold — `for fold_test in range(1, k_folds + 1)`
[REF yolo_mslesseg/ejecutar_pipeline.py:174-184] — and repeats the pipeline per plane (axial / coronal / sagital experiments are separate
`Modelo` runs [REF utils/Modelo.py:86-100]): 5 folds x 3 planes = 15 trainings that share nothing.  On a node they are independent jobs, one
process per GPU, no collective on the data path; this is the trivially linear counterpart of data-parallel training and is reported beside it.

    jobs = fold_plane_jobs()                                     # 15 (plano, fold) pairs in the reference's loop order
    mine = jobs_of_rank(jobs, costs, rank, world)                # longest-processing-time-first assignment, identical on every rank
    for job in mine: train_one(job)                              # `Trainer(..., replica=True)`: rank-local, no process group

`bench.py --mode replicas` measures it; `skip_if_done` mirrors the reference's stage-level idempotence (`existe_modelo_entrenado`
[REF utils/utils.py:240-251]: best.pt exists and is non-empty)."""
from __future__ import annotations

import os
from pathlib import Path
from typing import Callable, List, Optional, Sequence, Tuple

PLANES = ("axial", "coronal", "sagital")


def fold_plane_jobs(k_folds: int = 5, planes: Sequence[str] = PLANES) -> List[Tuple[str, int]]:
    return [(pl, fold) for pl in planes for fold in range(1, k_folds + 1)]


def schedule(costs: Sequence[float], workers: int) -> List[List[int]]:
    """Longest-processing-time-first: jobs by decreasing cost, each to the least-loaded worker (ties → lowest worker, lowest job index).
    Deterministic, so every rank derives the same assignment without talking to the others."""
    assert workers >= 1
    load = [0.0] * workers
    out: List[List[int]] = [[] for _ in range(workers)]
    for j in sorted(range(len(costs)), key=lambda j: (-float(costs[j]), j)):
        w = min(range(workers), key=lambda w: (load[w], w))
        out[w].append(j)
        load[w] += float(costs[j])
    return out


def jobs_of_rank(jobs: Sequence, costs: Optional[Sequence[float]] = None, rank: Optional[int] = None, world: Optional[int] = None) -> List:
    rank = int(os.environ.get("RANK", "0")) if rank is None else rank
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
    costs = [1.0] * len(jobs) if costs is None else costs
    return [jobs[j] for j in schedule(costs, world)[rank]]


def existe_modelo_entrenado(run_dir) -> bool:
    """[REF utils/utils.py:240-251]"""
    p = Path(run_dir) / "weights" / "best.pt"
    return p.exists() and p.stat().st_size > 0


def run_replicas(jobs: Sequence, train_one: Callable, costs: Optional[Sequence[float]] = None, run_dir_of: Optional[Callable] = None,
                 rank: Optional[int] = None, world: Optional[int] = None) -> List:
    """This rank's share of `jobs`, one after the other; jobs whose run directory already holds a trained model are skipped."""
    done = []
    for job in jobs_of_rank(jobs, costs, rank, world):
        if run_dir_of is not None and existe_modelo_entrenado(run_dir_of(job)):
            continue
        train_one(job)
        done.append(job)
    return done
