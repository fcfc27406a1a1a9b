"""Host side of the training leg on CPU: the LR/warm-up schedule against the reference's 25 results.csv (KAT #1), the data
pipeline (scan-line fill, affine warp, mosaic, collation), label files."""
import json

import numpy as np
import pytest
import torch

from mslesseg_amd import data as D
from mslesseg_amd import labels as LB
from mslesseg_amd.train import RESULT_COLUMNS, Schedule


def test_trainer_schedule_reproduces_all_25_results_csv(golden_dir):
    runs = json.loads((golden_dir / "lr_kat.json").read_text())
    worst = 0.0
    for r in runs:
        nb = r["nb_from_jpg"]
        s = Schedule(nb, 50, round(0.002 * 5 / (4 + 1), 6), 0.01, 3.0, 64, 64)
        for e, want in enumerate(r["lr_pg0"]):
            got = s.lr((e + 1) * nb - 1, e)  # the LR in force at the last iteration of the epoch is what gets logged
            worst = max(worst, abs(got - want) / want)
    assert worst <= 3.5e-6, worst


def test_results_csv_header_matches_reference(golden_dir):
    assert len(RESULT_COLUMNS) == 21 and RESULT_COLUMNS[0] == "epoch" and RESULT_COLUMNS[-3:] == ["lr/pg0", "lr/pg1", "lr/pg2"]
    assert RESULT_COLUMNS[2:6] == ["train/box_loss", "train/seg_loss", "train/cls_loss", "train/dfl_loss"]


def test_accumulate_ramp():
    s = Schedule(100, 50, 0.002, 0.01, 3.0, 16, 64)
    assert s.accumulate(0) == 1 and s.accumulate(10**6) == 4 and 1 <= s.accumulate(150) <= 4


def test_fill_polygon_matches_point_in_polygon():
    rng = np.random.default_rng(0)
    for _ in range(5):
        ang = np.sort(rng.uniform(0, 2 * np.pi, 7))
        poly = np.stack([20 + 12 * np.cos(ang), 18 + 9 * np.sin(ang)], 1)
        m = np.zeros((40, 44), np.uint8)
        D.fill_polygon(m, poly, 3)
        ys, xs = np.mgrid[0:40, 0:44]
        px, py = xs + 0.5, ys + 0.5
        inside = np.zeros_like(m, bool)
        x, y = poly[:, 0], poly[:, 1]
        for i in range(len(poly)):  # crossing-number test at pixel centres
            j = (i + 1) % len(poly)
            c = ((y[i] <= py) & (y[j] > py)) | ((y[j] <= py) & (y[i] > py))
            xi = x[i] + (py - y[i]) * (x[j] - x[i]) / (y[j] - y[i] + 1e-30)
            inside ^= c & (px < xi)
        assert (m > 0).sum() > 50 and ((m > 0) != inside).sum() <= 2 and set(np.unique(m)) <= {0, 3}


def test_warp_affine_identity_and_shift():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)
    assert np.array_equal(D.warp_affine(img, np.array([[1, 0, 0], [0, 1, 0]], float), (20, 30)), img)
    sh = D.warp_affine(img, np.array([[1, 0, 3], [0, 1, 2]], float), (20, 30))
    assert np.array_equal(sh[2:, 3:], img[:-2, :-3]) and (sh[:2] == D.PAD).all() and (sh[:, :3] == D.PAD).all()


def test_synthetic_dataset_augment_and_collate():
    ds = D.SyntheticSegDataset(6, 128, seed=0)
    rng = np.random.default_rng(0)
    for mosaic in (True, False):
        samples = [D.augment(ds, i, rng, mosaic, 128) for i in range(4)]
        b = D.collate(samples, 128)
        assert b["img"].shape == (4, 128, 128, 3) and b["img"].dtype == np.uint8 and b["masks"].shape == (4, 32, 32)
        T = len(b["cls"])
        assert b["bboxes"].shape == (T, 4) and b["batch_idx"].shape == (T,) and (b["bboxes"] >= 0).all() and (b["bboxes"] <= 1).all()
        for i in range(4):
            assert b["masks"][i].max() <= int((b["batch_idx"] == i).sum())
    a = D.collate([D.augment(ds, 0, np.random.default_rng(5), True, 128)], 128)
    c = D.collate([D.augment(ds, 0, np.random.default_rng(5), True, 128)], 128)
    assert all(np.array_equal(a[k], c[k]) for k in a)  # seeded ⇒ deterministic


def test_label_file_roundtrip(tmp_path):
    inst = [(0, np.array([[0.1, 0.2], [0.5, 0.25], [0.3, 0.8]], np.float32)), (0, np.array([[0.6, 0.6], [0.9, 0.6], [0.9, 0.9], [0.6, 0.9]], np.float32))]
    LB.write_label_file(tmp_path / "P1_FLAIR_3.txt", inst)
    back = LB.read_label_file(tmp_path / "P1_FLAIR_3.txt")
    assert len(back) == 2 and all(np.allclose(a[1], b[1], atol=1e-6) and a[0] == b[0] for a, b in zip(inst, back))
    (tmp_path / "bad.txt").write_text("0 0.1 0.2\n0 0.1 0.2 0.3\n")
    assert LB.read_label_file(tmp_path / "bad.txt") == [] and LB.read_label_file(tmp_path / "missing.txt") == []


def test_replica_schedule_is_deterministic_balanced_and_complete():
    """Zero-communication mode: the reference's 15 fold x plane trainings [REF ejecutar_pipeline.py:174-184] dealt over the GPUs of a node."""
    from mslesseg_amd import replicas as R

    jobs = R.fold_plane_jobs()
    assert len(jobs) == 15 and jobs[0] == ("axial", 1) and jobs[-1] == ("sagital", 5)
    # iterations per epoch of the reference's runs (112 axial / 163 coronal / 117 sagittal): coronal jobs cost ~1.45x
    costs = [{"axial": 112, "coronal": 163, "sagital": 117}[pl] for pl, _ in jobs]
    for world in (1, 2, 4, 8):
        parts = R.schedule(costs, world)
        flat = sorted(j for p in parts for j in p)
        assert flat == list(range(15))
        loads = [sum(costs[j] for j in p) for p in parts]
        assert max(loads) <= sum(costs) / world + max(costs)  # LPT bound
        assert [R.jobs_of_rank(jobs, costs, r, world) for r in range(world)] == [[jobs[j] for j in p] for p in parts]
    assert max(sum(costs[j] for j in p) for p in R.schedule(costs, 8)) == 275  # 8 GPUs: the makespan is one coronal + one axial job (lower bound sum/8 = 245)
    seen = []
    done = R.run_replicas(jobs, seen.append, costs, rank=3, world=8)
    assert done == seen == R.jobs_of_rank(jobs, costs, 3, 8) and len(done) in (1, 2)


def test_replica_skips_finished_runs(tmp_path):
    from mslesseg_amd import replicas as R

    jobs = R.fold_plane_jobs(2, ("axial",))
    d = tmp_path / "axial" / "fold1" / "weights"
    d.mkdir(parents=True)
    (d / "best.pt").write_bytes(b"x")
    seen = []
    R.run_replicas(jobs, seen.append, run_dir_of=lambda job: tmp_path / job[0] / f"fold{job[1]}", rank=0, world=1)
    assert seen == [("axial", 2)]


def test_validation_batches_are_dealt_over_ranks_without_changing_them():
    """val_batches: one rank → batches of min(batch, 128) in fold order; several ranks → the same boundaries (every rank takes each world-th batch) while
    every rank gets at least one, smaller equal batches for a fold too short for that."""
    from mslesseg_amd.train import val_batches

    one = val_batches(584, 256, 1)
    assert one == [(0, 128), (128, 256), (256, 384), (384, 512), (512, 584)]
    assert val_batches(584, 256, 2) == one and val_batches(584, 256, 4) == one
    eight = val_batches(584, 256, 8)
    assert len(eight) == 8 and eight[0] == (0, 73) and eight[-1][1] == 584 and all(b[1] - b[0] <= 73 for b in eight)
    for world in (1, 2, 3, 8):
        bs = val_batches(361, 64, world)
        dealt = sorted(b for r in range(world) for b in bs[r::world])
        assert dealt == bs and bs[0][0] == 0 and bs[-1][1] == 361 and all(a[1] == b[0] for a, b in zip(bs, bs[1:]))
    assert val_batches(3, 128, 8) == [(0, 1), (1, 2), (2, 3)]  # fewer slices than ranks: some ranks score nothing


def test_segstats_parts_of_several_ranks_merge_into_the_single_rank_lists():
    """SegStats.export / merged: per-image rows carry the image index, so the union of what the ranks collected over disjoint batches is, row for
    row, what one rank collects walking the fold in order (sharded validation: train._validate)."""
    from mslesseg_amd import metrics as MT

    rng = np.random.default_rng(0)
    B, P, G = 12, 7, 3

    def batch(lo, hi):
        n = hi - lo
        r = np.random.default_rng([1, lo])
        n_pred = torch.from_numpy(r.integers(0, P + 1, n))
        n_gt = torch.from_numpy(r.integers(0, G + 1, n))
        gt_xy = torch.from_numpy(r.uniform(0, 60, (n, G, 2)).astype(np.float32))
        gt = torch.cat([gt_xy, gt_xy + torch.from_numpy(r.uniform(8, 30, (n, G, 2)).astype(np.float32))], 2)
        pr = gt[:, r.integers(0, G, P)] + torch.from_numpy(r.normal(0, 2.0, (n, P, 4)).astype(np.float32))
        conf = torch.from_numpy(r.uniform(0.01, 1, (n, P)).astype(np.float32))
        pm = torch.from_numpy((r.random((n, P, 64)) < 0.4).astype(np.float32))
        gm = torch.from_numpy((r.random((n, G, 64)) < 0.4).astype(np.float32))
        return pr, conf, torch.zeros(n, P), pm, n_pred, gt, torch.zeros(n, G), gm, n_gt

    bounds = [(0, 4), (4, 8), (8, 12)]
    single = MT.SegStats()
    for lo, hi in bounds:
        single.add_batch(*batch(lo, hi), first_id=lo)
    ranks = [MT.SegStats(), MT.SegStats()]
    for bi, (lo, hi) in enumerate(bounds):
        ranks[bi % 2].add_batch(*batch(lo, hi), first_id=lo)
    merged = MT.SegStats.merged([ranks[1].export(), ranks[0].export()])  # part order must not matter
    assert merged.pids == single.pids and merged.tids == single.tids and len(single.tids) == B
    for f in ("tp_b", "tp_m", "conf", "pcls", "tcls"):
        a, b = getattr(merged, f), getattr(single, f)
        assert len(a) == len(b) and all(np.array_equal(x, y) for x, y in zip(a, b)), f
    assert merged.result() == single.result() and single.result()["metrics/mAP50(B)"] > 0
    del rng
