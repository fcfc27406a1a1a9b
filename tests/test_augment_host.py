"""Host half of the device data feeder (augment.DeviceAugmenter.prepare: random draws + batched label geometry) against the per-sample
restatement data.augment + data.collate on the same generator state: same kept instances in the same order, same boxes and classes, the same
vertices handed to the raster op.  (The image warp and the mask raster are device ops: tests/test_gpu_augment.py.)"""
import numpy as np
import pytest

from mslesseg_amd import augment as A
from mslesseg_amd import data as D


class ShapesDataset:
    """Slices of the three shapes the planes produce (640x534, 534x640, 640x640) with random polygons, some degenerate."""

    def __init__(self, n, seed, size=640):
        rng = np.random.default_rng(seed)
        self.items = []
        for i in range(n):
            h, w = [(size, size * 534 // 640), (size * 534 // 640, size), (size, size)][i % 3]
            img = rng.integers(0, 256, size=(h, w, 1), dtype=np.uint8).repeat(3, 2)
            inst = []
            for _ in range(int(rng.integers(0, 9))):
                c = rng.uniform(0.05, 0.95, 2) * [w, h]
                rad = rng.uniform(0.004, 0.12) * size
                k = int(rng.integers(2, 12))  # k = 2: not a polygon, must never become a label
                ang = np.sort(rng.uniform(0, 2 * np.pi, k))
                p = np.stack([c[0] + rad * np.cos(ang), c[1] + rad * np.sin(ang)], 1)
                inst.append((0, np.clip(p, 0, [w - 1, h - 1]).astype(np.float32)))
            if i % 7 == 3:
                inst += [(0, inst[0][1].copy())] if inst else []  # two instances of exactly equal area: the order must still agree
            self.items.append((img, inst))

    def __len__(self):
        return len(self.items)

    def get(self, i):
        return self.items[i]


def test_ragged_arange():
    assert A.ragged_arange(np.array([5, 0, 9]), np.array([2, 0, 3])).tolist() == [5, 6, 9, 10, 11]
    assert A.ragged_arange(np.array([], np.int64), np.array([], np.int64)).size == 0


@pytest.mark.parametrize("mosaic,augment,size", [(True, True, 640), (False, True, 640), (False, False, 640), (True, True, 320)])
def test_prepare_labels_equal_the_per_sample_path(mosaic, augment, size):
    ds = ShapesDataset(24, seed=1, size=size)
    aug = A.DeviceAugmenter(A.SliceCache(ds), size)
    idx = [3, 0, 7, 7, 12, 21, 5, 10]
    for seed in range(4):
        draws = D.draw_params(np.random.default_rng([seed, 9]), len(idx), len(ds), mosaic, size)
        if augment:
            samples = [D.augment(ds, i, None, mosaic, size, draws=D.draw_row(draws, b)) for b, i in enumerate(idx)]
        else:
            samples = [D.plain(ds, i, size) for i in idx]
        want = D.collate(samples, size)
        got = aug.prepare(idx, None, mosaic, augment, draws=draws)
        assert np.array_equal(got["batch_idx"], want["batch_idx"])
        assert np.array_equal(got["cls"], want["cls"])
        assert np.array_equal(got["bboxes"], want["bboxes"])  # bit-equal float32
        assert got["n_max"] == want["n_max"]
        # vertices for the raster op, polygon by polygon in overlap order, are the collate polygons / mask_ratio
        k = 0
        for b, (_, inst) in enumerate(samples):
            polys = [np.asarray(p, np.float32) for _, p in inst if len(p) >= 3]
            _, fp, fo = D.flatten_instances([(0, p) for p in polys])
            order = np.argsort(-D.poly_areas(fp, fo), kind="stable") if polys else []
            first, cnt = got["ranges"][b]
            assert cnt == len(polys)
            for j, o in enumerate(order):
                v0, nv, val, _ = got["poly"][first + j]
                assert val == j + 1 and np.array_equal(got["pts"][v0 : v0 + nv], polys[o] / 4)
                k += 1
        assert k == len(want["cls"])
    if augment:  # a generator instead of a record: the batch draw is what the per-sample path draws for a batch of one
        r1, r2 = np.random.default_rng(5), np.random.default_rng(5)
        a = D.collate([D.augment(ds, 3, r1, mosaic, size)], size)
        b = aug.prepare([3], r2, mosaic, True)
        assert np.array_equal(a["bboxes"], b["bboxes"]) and r1.random() == r2.random()
