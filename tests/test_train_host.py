"""Host side of the training leg on CPU: the LR/warm-up schedule against the reference's 25 results.csv (KAT #1), the data
pipeline (scan-line fill, affine warp, mosaic, collation), label files."""
import json

import numpy as np
import pytest

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
