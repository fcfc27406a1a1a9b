"""Trainer behind ``YOLO.train(data=, epochs=, batch=-1, cache=True, project=, name=, verbose=False)`` (boundary B2)
[REF yolo_mslesseg/scripts/train.py:358-366].  Blocking; creates ``<project>/<name>/weights/{best,last}.pt`` (non-empty),
``results.csv`` (the reference's 21 columns) and ``args.yaml`` — the files `entrenamiento_exitoso` checks
[REF train.py:105-116].

Resolved hyper-parameters are those frozen in the reference's args.yaml [REF trains/Base/FLAIR_P50c_5folds_50epochs/axial/
fold1/args.yaml]: imgsz 640, nbs 64, seed 0, warmup 3 epochs, close_mosaic 10, weight_decay 5e-4, optimizer 'auto' → AdamW
lr0 = round(0.002*5/(4+nc), 6), beta1 0.9, warm-up from 0 for every group — the schedule the 25 results.csv files pin
(tests/test_oracle_pins.py KAT #1; beyond 10 000 iterations 'auto' takes SGD + Nesterov, lr 0.01, which no run of the reference reaches).
Gradient clip 10, ModelEMA(0.9999, tau 2000) [UPSTREAM engine/trainer.py].

Data: the slice cache of `cache=True` lives in HBM and every batch is augmented there (augment.py: mosaic, scale / translate warp, value gain,
flip, mask rasterisation — two launches, one batch ahead on a side stream); a host thread only draws the random numbers and does the label
geometry.  Validation: once per epoch, eval mode, EMA weights, the whole held-out fold — val/* losses and box / mask P, R, mAP from one pass.

One process per GPU.  Data parallelism = ONE RCCL all-reduce(SUM) of the flat gradient buffer per optimizer step
(ultralytics scales the loss by world_size and lets DDP average: the same sum); BatchNorm statistics stay per-rank
(no SyncBatchNorm: a documented deviation from ultralytics' DDP path, which the reference never exercises — SURVEY §7.3 #6).
"""
from __future__ import annotations

import csv
import datetime
import math
import os
import shutil
import struct
import threading
import time
from pathlib import Path
from queue import Full, Queue
from typing import Optional

import numpy as np
import torch
import yaml

from . import data as D
from . import hiplib, params
from .hiplib import MSL_BF16, MSL_F32
from .loss import GAIN_BOX, GAIN_CLS, GAIN_DFL, segmentation_loss
from .segloss import SegLossOp, device_targets, pack_targets
from .trainprog import ParamStore, TrainPlan

RESULT_COLUMNS = ["epoch", "time", "train/box_loss", "train/seg_loss", "train/cls_loss", "train/dfl_loss", "metrics/precision(B)", "metrics/recall(B)",
                  "metrics/mAP50(B)", "metrics/mAP50-95(B)", "metrics/precision(M)", "metrics/recall(M)", "metrics/mAP50(M)", "metrics/mAP50-95(M)",
                  "val/box_loss", "val/seg_loss", "val/cls_loss", "val/dfl_loss", "lr/pg0", "lr/pg1", "lr/pg2"]  # [REF trains/…/results.csv:1]

DEFAULTS = dict(imgsz=640, nbs=64, seed=0, lrf=0.01, warmup_epochs=3.0, warmup_bias_lr=0.0, weight_decay=0.0005, close_mosaic=10,
                beta1=0.9, beta2=0.999, eps=1e-8, clip=10.0, ema_decay=0.9999, ema_tau=2000.0, auto_batch=None, augment=True, val_max=None,
                optimizer=None, momentum=0.937, warmup_momentum=0.8, device_augment=True)


def _fbits(x: float) -> int:
    return struct.unpack("<i", struct.pack("<f", float(x)))[0]


def auto_batch_size(device, imgsz: int, fraction: float = 0.6, cap: int = 256) -> int:
    """Per-GPU batch for `batch=-1`: the largest multiple of 32 whose train plan fits `fraction` of the device memory [UPSTREAM
    utils/autobatch.py check_train_batch_size(batch=0.6)], capped at 256 slices to bound the step latency.  Memory per slice is the
    measured footprint of the train plan (activations + their gradients, bf16): 36.8 GB at batch 256 and 640x640 → 0.145 GB per slice."""
    total = torch.cuda.get_device_properties(device).total_memory
    per_slice = 0.145e9 * (imgsz / 640.0) ** 2
    b = int(fraction * total / per_slice) // 32 * 32
    return int(max(16, min(cap, b)))


class Schedule:
    """LR per iteration: lr0 * lf(epoch) * warm-up ramp; `accumulate` ramps 1 → nbs/batch during warm-up  [UPSTREAM BaseTrainer]."""

    def __init__(self, nb: int, epochs: int, lr0: float, lrf: float, warmup_epochs: float, batch: int, nbs: int):
        self.nb, self.epochs, self.lr0, self.lrf, self.batch, self.nbs = nb, epochs, lr0, lrf, batch, nbs
        self.nw = max(round(warmup_epochs * nb), 100) if warmup_epochs > 0 else -1

    def lf(self, epoch: int) -> float:
        return max(1 - epoch / self.epochs, 0) * (1.0 - self.lrf) + self.lrf

    def lr(self, ni: int, epoch: int) -> float:
        target = self.lr0 * self.lf(epoch)
        if ni <= self.nw:
            return float(np.interp(ni, [0, self.nw], [0.0, target]))
        return target

    def accumulate(self, ni: int) -> int:
        full = max(round(self.nbs / self.batch), 1)
        if ni <= self.nw:
            return max(1, int(np.interp(ni, [0, self.nw], [1, self.nbs / self.batch]).round()))
        return full


def shard_indices(n: int, epoch: int, seed: int, rank: int, world: int) -> np.ndarray:
    """Slices seen by `rank` in `epoch`: a seeded permutation dealt round-robin — disjoint across ranks, identical on every rank."""
    return np.random.default_rng([seed, epoch]).permutation(n)[rank::world]


def collective_selfcheck(device, n: int = 1 << 18) -> dict:
    """Start-up check of the gradient collective on the REAL backend (RCCL over xGMI on a GPU node): every rank contributes (rank + 1) x a known ramp, the
    all-reduce must return sum(1..world) x the ramp on every element.  → {rank, world, backend, device, ok, ms}; raises when the sum is wrong — a
    mis-wired communicator (wrong device per rank, a peer on another job's port) then stops the run at once instead of training on garbage."""
    dist = torch.distributed
    world, rank = dist.get_world_size(), dist.get_rank()
    backend = dist.get_backend()
    on_dev = backend != "gloo"
    ramp = torch.arange(n, dtype=torch.float32, device=device if on_dev else "cpu") % 251
    buf = ramp * float(rank + 1)
    if on_dev:
        torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    if on_dev:
        torch.cuda.synchronize(device)
    ms = (time.perf_counter() - t0) * 1e3
    ok = bool(torch.equal(buf, ramp * float(world * (world + 1) // 2)))
    rec = {"rank": rank, "world": world, "backend": backend, "device": str(device), "ok": ok, "first_allreduce_ms": round(ms, 3)}
    if not ok:
        raise RuntimeError(f"gradient collective self-check failed: {rec}")
    return rec


def allreduce_gradients(flat: torch.Tensor) -> torch.Tensor:
    """The one collective of the training data path: SUM of the flat gradient buffer over ranks (RCCL on GPUs, gloo in tests).  A gloo group
    over device tensors (two ranks sharing one GPU in the single-box rehearsal test) stages through host memory."""
    dist = torch.distributed
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if flat.is_cuda and dist.get_backend() == "gloo":
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            flat.copy_(host)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def gather_objects(obj, world: int, group=None, to_rank0: bool = False):
    """[obj of rank 0, …, obj of rank world-1] on every rank — or, with `to_rank0`, on rank 0 only (None elsewhere).  Small host-side records
    (validation statistics), not on the training data path; `group`: the process group to use (the trainer's validation group, so that the
    writer threads' gathers never interleave with the main threads' gradient all-reduces)."""
    if world <= 1:
        return [obj]
    dist = torch.distributed
    if to_rank0:
        out = [None] * world if dist.get_rank() == 0 else None
        dist.gather_object(obj, out, dst=0, group=group)
        return out
    out = [None] * world
    dist.all_gather_object(out, obj, group=group)
    return out


def val_batches(n: int, batch: int, world: int):
    """[(first, last+1)] slice ranges of the validation batches.  One rank: batches of min(batch, 128) in fold order.  Several ranks: the SAME
    boundaries while there are at least `world` of them (rank r takes every world-th batch, so every per-batch loss is the number a single rank
    computes), smaller equal batches once the fold is too short to give every rank one."""
    vb = min(batch, 128)
    if world > 1 and math.ceil(n / vb) < world:
        vb = max(math.ceil(n / world), 1)
    return [(b0, min(b0 + vb, n)) for b0 in range(0, n, vb)]


class CheckpointWriter:
    """last.pt / best.pt off the training stream: the EMA buffers are copied into pinned host memory by an asynchronous copy on the caller's
    stream, a thread waits for that copy and writes the files while the next epoch's steps are already queued.  One write in flight; `wait()`
    before the files are read (end of fit) — it re-raises what the thread caught."""

    def __init__(self, trainer):
        self.tr = trainer
        st = trainer.store
        if trainer.rank == 0:  # only rank 0 writes files
            self.hp = torch.empty(st.p.numel(), dtype=torch.float32).pin_memory()
            self.hb = torch.empty(st.b.numel(), dtype=torch.float32).pin_memory()
        self.thread, self.error = None, None

    def wait(self):
        if self.thread is not None:
            self.thread.join()
            self.thread = None
        if self.error is not None:
            e, self.error = self.error, None
            raise RuntimeError(f"checkpoint writer failed: {e!r}") from e

    def submit(self, job, snapshot: bool = True):
        """Snapshot the EMA buffers (asynchronous copy on the caller's stream; `snapshot=False`: a rank that writes no files) and run `job(self)`
        in the writer thread once the copy — and everything queued before it, e.g. the validation pass of the epoch — has finished.  `job`
        decides what to write and calls `write`."""
        self.wait()
        tr = self.tr
        if snapshot:
            self.hp.copy_(tr.ema_p, non_blocking=True)
            self.hb.copy_(tr.ema_b, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(tr.device))

        def work():
            try:
                torch.cuda.set_device(tr.device)  # a new thread starts on cuda:0 whatever the trainer's device is: events and pinned copies belong to the rank's own GPU
                ev.synchronize()
                job(self)
            except BaseException as e:  # surfaced by wait() in the main thread: the rank then leaves fit() with an exception and the launcher ends the job
                self.error = e

        self.thread = threading.Thread(target=work, daemon=True)
        self.thread.start()

    def write(self, extra: dict, is_best: bool) -> None:
        """(writer thread) last.pt from the snapshot; best.pt = the same bytes when the fitness improved."""
        tr = self.tr
        sd = tr.store.state_dict(p=self.hp, b=self.hb)
        last, best = tr.wdir / "last.pt", tr.wdir / "best.pt"
        # written beside the target and renamed into place: a reader (or a crash of this process) never sees a truncated checkpoint
        tmp = last.with_name(last.name + ".tmp")
        params.save_checkpoint(tmp, sd, tr.store.scale, tr.nc, tr.names, extra)
        os.replace(tmp, last)
        if is_best:
            tmpb = best.with_name(best.name + ".tmp")
            shutil.copyfile(last, tmpb)  # the same bytes: one serialisation per epoch
            os.replace(tmpb, best)


class Trainer:
    def __init__(self, yolo, data=None, epochs: int = 100, batch: int = -1, cache: bool = True, project=None, name: str = "train",
                 verbose: bool = False, dataset=None, val_dataset=None, max_iters: Optional[int] = None, replica: bool = False, **overrides):
        """`replica=True`: an independent training on this process's GPU even under a multi-process launch — the zero-communication mode of
        SURVEY §8e (the reference's 15 fold x plane trainings are independent jobs [REF yolo_mslesseg/ejecutar_pipeline.py:174-184]); no process
        group is joined and no gradient is exchanged (replicas.py schedules such jobs over the GPUs of a node)."""
        self.yolo, self.epochs, self.verbose = yolo, int(epochs), verbose
        self.hyp = {**DEFAULTS, **{k: v for k, v in overrides.items() if k in DEFAULTS}}
        self.max_iters = max_iters
        self.rank = 0 if replica else int(os.environ.get("RANK", "0"))
        self.world = 1 if replica else int(os.environ.get("WORLD_SIZE", "1"))
        self.local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)  # more ranks than GPUs only in the single-box rehearsal test
        self.device = torch.device(f"cuda:{self.local}" if (self.world > 1 or (replica and "LOCAL_RANK" in os.environ)) else yolo.device)
        if not torch.cuda.is_available():
            raise hiplib.MslError("no GPU: training runs only on the HIP kernels (no CPU fallback)")
        torch.cuda.set_device(self.device)
        if self.world > 1 and not torch.distributed.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            torch.distributed.init_process_group("nccl", device_id=self.device)
        self.collective = collective_selfcheck(self.device) if self.world > 1 else None  # first use of the communicator: a known sum, checked
        self.yolo.device = str(self.device)  # predict()/val() after fit() run where the training ran (one process per GPU: never all on cuda:0)
        # a process group of its own for the per-epoch gather of validation records: it is issued from the writer threads (one gather per epoch on
        # every rank, in epoch order), so it must not share a communicator with the main threads' gradient all-reduces.  Host objects: gloo.
        # (created at the start of fit(): gloo announces its connections on stdout, and a Trainer that only steps — bench.py — must stay silent)
        self._val_group = None
        # ---- data
        self.names, self.nc = {0: "lesion"}, 1
        if dataset is None:
            cfg = yaml.safe_load(Path(data).read_text())
            self.nc = int(cfg.get("nc", len(cfg.get("names", [1]))))
            nm = cfg.get("names", ["lesion"])
            self.names = dict(enumerate(nm)) if isinstance(nm, (list, tuple)) else {int(k): v for k, v in nm.items()}
            root = Path(cfg.get("path", "."))
            tr, va = Path(cfg["train"]), Path(cfg["val"])
            # data parallel: the ranks of the node split the PNG decoding of the fold and exchange the decoded slices (data.SegDataset.exchange)
            shard = (self.rank, self.world) if self.world > 1 else None
            dataset = D.SegDataset(tr if tr.is_absolute() else root / tr, self.hyp["imgsz"], shard=shard)
            val_dataset = D.SegDataset(va if va.is_absolute() else root / va, self.hyp["imgsz"], shard=shard)
            dataset.exchange()
            val_dataset.exchange()
        self.ds, self.val_ds = dataset, val_dataset
        # the slice cache of `cache=True` lives in HBM and batches are augmented there (augment.py); device_augment=False keeps the NumPy path of data.py
        self.aug = self.val_aug = None
        self._aug_stream = None
        self._val_engine, self._val_loss_ops, self._ckpt, self._pin = None, None, None, {}
        if self.hyp["device_augment"]:
            from .augment import DeviceAugmenter, SliceCache

            self.aug = DeviceAugmenter(SliceCache(self.ds, self.device), self.hyp["imgsz"])
            if self.val_ds is not None and len(self.val_ds):
                self.val_aug = DeviceAugmenter(SliceCache(self.val_ds, self.device), self.hyp["imgsz"])
        self.data_path = str(data) if data is not None else "synthetic"
        # batch=-1: "choose for me" [REF train.py:361] — upstream's autobatch fills 60 % of the device memory; same rule here (see auto_batch_size)
        self.batch = int(batch) if batch and batch > 0 else int(self.hyp["auto_batch"] or auto_batch_size(self.device, self.hyp["imgsz"]))
        self.save_dir = Path(project or "runs/segment") / name
        self.wdir = self.save_dir / "weights"
        if self.rank == 0:
            self.wdir.mkdir(parents=True, exist_ok=True)
        # ---- model state → flat master buffers
        scale = yolo.scale
        state = yolo.state
        if state is None or int(yolo.nc) != self.nc:  # like ultralytics: rebuild the head for the dataset's nc, keep what matches
            fresh = params.init_state(scale, self.nc, seed=self.hyp["seed"])
            if state is not None:
                for k, v in state.items():
                    if k in fresh and tuple(fresh[k].shape) == tuple(v.shape):
                        fresh[k] = v.float() if v.is_floating_point() else v
            state = fresh
        state = {k: (v.float() if v.is_floating_point() else v) for k, v in state.items()}
        self.store = ParamStore(scale, self.nc, self.device)
        self.store.load_state(state)
        self.dtype = getattr(yolo, "train_dtype", yolo.dtype)  # training arithmetic (bf16 by default: the counterpart of the reference's amp=True)
        S = self.hyp["imgsz"]
        # data parallel: the backward program is cut behind the head + neck layers (model.11..) so that their gradient bucket is reduced beside the backbone's
        # backward (forward_backward(reduce_now=True)); MSLESSEG_GRAD_BUCKETS=1 keeps one program and one all-reduce behind it
        cut = 11 if ((self.world > 1 and os.environ.get("MSLESSEG_GRAD_BUCKETS", "2") != "1") or os.environ.get("MSLESSEG_FORCE_CUT") == "1") else None  # FORCE_CUT: measure the cut's own cost on one GPU
        self.plan = TrainPlan(self.store, self.batch, S, S, self.dtype, bucket_cut=cut)
        self._buckets = self.store.bucket_ranges(cut) if cut is not None else None
        self._reduce_works, self._reduced = [], False
        lv = [self.plan.levels[i] for i in sorted(self.plan.levels)]
        self.loss_op = SegLossOp(lv, [tuple(self.plan.G(v) for v in l) for l in lv], self.plan.proto_view, self.plan.G(self.plan.proto_view), self.nc, S, S,
                                 self.dtype, self.device)
        self.torch_loss = False  # True: the autograd tensor-op loss of loss.py (the fp32 reference of the HIP loss op) — tests only
        # ---- optimizer ('auto' rule) and schedule
        per_rank = math.ceil(len(self.ds) / self.world)
        self.nb = max(math.ceil(per_rank / self.batch), 1)
        total_batch = self.batch * self.world
        iterations = math.ceil(len(self.ds) / max(total_batch, self.hyp["nbs"])) * self.epochs
        # optimizer='auto' [UPSTREAM build_optimizer]: AdamW(lr 0.002*5/(4+nc)) up to 10 000 iterations, SGD(lr 0.01, Nesterov) beyond; every run of
        # the reference takes the AdamW branch (<= 8 700 iterations, pinned by its 25 results.csv — tests/test_oracle_pins.py)
        self.optimizer = self.hyp.get("optimizer") or ("AdamW" if iterations <= 10000 else "SGD")
        if self.optimizer not in ("AdamW", "SGD"):
            raise ValueError(f"optimizer {self.optimizer!r}: only the two branches of 'auto' (AdamW, SGD) exist here")
        self.lr0 = round(0.002 * 5 / (4 + self.nc), 6) if self.optimizer == "AdamW" else 0.01
        self.sched = Schedule(self.nb, self.epochs, self.lr0, self.hyp["lrf"], self.hyp["warmup_epochs"], total_batch, self.hyp["nbs"])
        self.wd = self.hyp["weight_decay"] * total_batch * max(round(self.hyp["nbs"] / total_batch), 1) / self.hyp["nbs"]
        self.opt_steps = 0
        self.ema_p, self.ema_b, self.ema_updates = self.store.p.clone(), self.store.b.clone(), 0
        self.gscale = torch.ones(1, dtype=torch.float32, device=self.device)
        self.best_fitness, self.t0 = None, None
        self.ni = 0  # iterations seen (the momentum warm-up of the SGD branch reads it)

    # ------------------------------------------------------------------ data feeding
    def _batches(self, epoch: int):
        n = len(self.ds)
        rng = np.random.default_rng([self.hyp["seed"], epoch, self.rank])
        mine = shard_indices(n, epoch, self.hyp["seed"], self.rank, self.world)
        mosaic = epoch < self.epochs - self.hyp["close_mosaic"]
        q: Queue = Queue(maxsize=4)
        stop = threading.Event()

        def put(item) -> bool:
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.2)
                    return True
                except Full:
                    continue
            return False

        def work():
            try:
                for b in range(self.nb):
                    idx = [int(mine[(b * self.batch + j) % len(mine)]) for j in range(self.batch)]
                    if self.aug is not None:  # host part only (random draws + label geometry); the consumer launches the two device ops
                        if not put(self.aug.prepare(idx, rng, mosaic, augment=bool(self.hyp["augment"]))):
                            return
                        continue
                    if self.hyp["augment"]:
                        samples = [D.augment(self.ds, i, rng, mosaic, self.hyp["imgsz"]) for i in idx]
                    else:
                        samples = [D.plain(self.ds, i, self.hyp["imgsz"]) for i in idx]
                    if not put(D.collate(samples, self.hyp["imgsz"])):
                        return
                put(None)
            except BaseException as e:  # a bad label file, a shape mismatch …: hand it to the training loop instead of dying silently
                put(e)

        threading.Thread(target=work, daemon=True).start()

        def render(item):
            """Device half of the feeder, one batch AHEAD and on a side stream: its two launches and small uploads overlap the previous step's
            kernels instead of sitting between two steps (measured: 1.3 ms per 128-slice batch when serialised)."""
            if self._aug_stream is None:
                self._aug_stream = torch.cuda.Stream(self.device)
            with torch.cuda.stream(self._aug_stream):
                out = self.aug.render(item)
                ev = torch.cuda.Event()
                ev.record(self._aug_stream)
            return out, ev

        try:
            pending = None
            while True:
                item = q.get()
                if isinstance(item, BaseException):
                    raise RuntimeError(f"data feeder failed: {item!r}") from item
                if self.aug is None:
                    if item is None:
                        return
                    yield item
                    continue
                nxt = render(item) if item is not None else None
                if pending is not None:
                    out, ev = pending
                    cur = torch.cuda.current_stream(self.device)
                    cur.wait_event(ev)
                    for k in ("img", "masks", "gt"):
                        out[k].record_stream(cur)  # allocated on the side stream, consumed here
                    yield out
                pending = nxt
                if item is None:
                    return
        finally:
            stop.set()  # early exit (max_iters, an exception in the step): the feeder stops waiting on the queue

    # ------------------------------------------------------------------ one optimisation step
    def to_device(self, batch):
        """numpy batch (data.collate) → device tensors; the bench keeps one such batch resident in HBM."""
        out = {k: (torch.from_numpy(v).to(self.device, non_blocking=True) if isinstance(v, np.ndarray) else v) for k, v in batch.items()}
        S = self.hyp["imgsz"]
        gt, _ = pack_targets(batch["batch_idx"], batch["cls"], batch["bboxes"], len(batch["masks"]), S, S)
        out["gt"] = torch.from_numpy(gt).to(self.device, non_blocking=True)
        return out

    def _reduce_bucket(self, k: int) -> None:
        """All-reduce (SUM) of gradient bucket k's flat ranges, issued now: RCCL runs it on its own stream behind what this stream holds so far, so bucket 0
        (head + neck) overlaps the backbone's backward program.  gloo (the single-box rehearsal): staged through host memory, synchronously."""
        dist = torch.distributed
        for lo, hi in self._buckets[k]:
            if hi <= lo:
                continue
            view = self.store.g[lo:hi]
            if dist.get_backend() == "gloo":
                host = view.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM)
                view.copy_(host)
            else:
                self._reduce_works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True))

    def forward_backward(self, batch, reduce_now: bool = False) -> torch.Tensor:
        """HIP forward → HIP loss op (value + gradient of the head outputs) → HIP backward.  Gradients ACCUMULATE into store.g.
        `reduce_now` (data parallel): this is the last micro-batch before an optimizer step — the gradient buckets are all-reduced as they complete
        (the first one between the two backward programs); optimizer_step() then only waits for them."""
        plan = self.plan
        bucketed = reduce_now and self.world > 1 and self._buckets is not None and not self.torch_loss
        if not torch.is_tensor(batch["img"]):
            batch = self.to_device(batch)
        plan.in_view.t.copy_(batch["img"].reshape(-1), non_blocking=True)
        plan.pack()
        plan.forward()
        if not self.torch_loss:
            items = self.loss_op(batch["gt"], batch["masks"])[:4].clone()  # writes d(loss)/d(head outputs) into the plan's gradient views
            if bucketed:
                plan.backward(on_cut=lambda: self._reduce_bucket(0))
                self._reduce_bucket(1)
                self._reduced = True
            elif self.world == 1 and plan.cut_segment is not None:  # MSLESSEG_FORCE_CUT=1: the two-program backward without a collective (its own cost, one GPU)
                plan.backward(on_cut=lambda: None)
            else:
                plan.backward()
            return items
        outs = plan.head_outputs()
        leaves = [[t.detach().requires_grad_() for t in lv] for lv in outs["levels"]]
        proto = outs["proto"].detach().float().requires_grad_()
        tb = {k: v for k, v in batch.items() if k != "img"}
        loss, items = segmentation_loss([tuple(lv) for lv in leaves], proto, tb, self.nc)
        flat = [t for lv in leaves for t in lv] + [proto]
        grads = torch.autograd.grad(loss, flat, allow_unused=True)  # no positive anchor ⇒ the box branch gets no gradient
        hg = plan.head_grads()
        k = 0
        for li, lv in enumerate(hg["levels"]):
            for j, gv in enumerate(lv):
                if j == 1:
                    plan.G(plan.levels[li][1]).t.zero_()  # padding channels of the narrow class head
                if grads[k] is None:
                    gv.zero_()
                else:
                    gv.copy_(grads[k])
                k += 1
        if grads[k] is None:
            hg["proto"].zero_()
        else:
            hg["proto"].copy_(grads[k])
        plan.backward()
        return items

    def optimizer_step(self, lr: float) -> None:
        st = self.store
        if self.world > 1:
            if self._reduced:  # the buckets were issued inside forward_backward(reduce_now=True): this stream waits for them
                for w in self._reduce_works:
                    w.wait()
                self._reduce_works, self._reduced = [], False
            else:
                allreduce_gradients(st.g)  # the one collective of the data path (SUM over ranks)
        norm = torch.linalg.vector_norm(st.g)
        self.gscale.copy_(torch.clamp(self.hyp["clip"] / (norm + 1e-6), max=1.0))
        self.opt_steps += 1
        t = self.opt_steps
        h = self.hyp
        bc1, bc2 = 1 - h["beta1"] ** t, 1 - h["beta2"] ** t
        s = torch.cuda.current_stream(self.device).cuda_stream
        for lo, hi, wd in ((0, st.n_decay, self.wd), (st.n_decay, st.n, 0.0)):
            n = hi - lo
            if n <= 0:
                continue
            if self.optimizer == "SGD":  # momentum warms up from 0.8 to 0.937 with the learning rate [UPSTREAM BaseTrainer warm-up]
                mu = float(np.interp(min(self.ni, max(self.sched.nw, 0)), [0, max(self.sched.nw, 1)], [h["warmup_momentum"], h["momentum"]])) if self.sched.nw > 0 else h["momentum"]
                op = hiplib.make_op(hiplib.OP_SGD, MSL_F32, p=(st.p.data_ptr() + 4 * lo, st.g.data_ptr() + 4 * lo, st.m.data_ptr() + 4 * lo, 0, 0, self.gscale.data_ptr()),
                                    i={0: n & 0x7FFFFFFF, 1: n >> 31, 2: 1 if t == 1 else 0}, f=(lr, mu, wd))
            else:
                op = hiplib.make_op(hiplib.OP_ADAMW, MSL_F32, p=(st.p.data_ptr() + 4 * lo, st.g.data_ptr() + 4 * lo, st.m.data_ptr() + 4 * lo,
                                                                 st.v.data_ptr() + 4 * lo, 0, self.gscale.data_ptr()),
                                    i={0: n & 0x7FFFFFFF, 1: n >> 31, 2: _fbits(wd), 3: _fbits(bc1), 4: _fbits(bc2)}, f=(lr, h["beta1"], h["beta2"], h["eps"]))
            hiplib.launch(op, s)
        st.g.zero_()
        # EMA of parameters and BN running statistics
        self.ema_updates += 1
        d = h["ema_decay"] * (1 - math.exp(-self.ema_updates / h["ema_tau"]))
        for e, src in ((self.ema_p, st.p), (self.ema_b, st.b)):
            n = e.numel()
            hiplib.launch(hiplib.make_op(hiplib.OP_EMA, MSL_F32, p=(e.data_ptr(), src.data_ptr()), i={0: n & 0x7FFFFFFF, 1: n >> 31}, f=(d,)), s)

    # ------------------------------------------------------------------ validation (held-out fold, EMA weights, eval-mode BatchNorm)
    @torch.no_grad()
    def _validate(self, deferred: bool = False):
        """One eval-mode pass over the held-out fold with the EMA weights → (val losses [4], metrics dict | None)  [UPSTREAM SegmentationValidator:
        the validator runs the EMA model in eval mode (running BatchNorm statistics), accumulates v8SegmentationLoss on its raw head outputs for
        the val/* columns and scores NMS(conf 0.001, IoU 0.7, max_det 300) detections for box/mask P, R, mAP50, mAP50-95].  Masks are compared at
        prototype resolution like upstream's default `process_mask`.  `val_max` (not a reference setting) bounds the slices for quick runs and
        is recorded in args.yaml.

        **Sharded over ranks** (every rank calls this): the batches of `val_batches` are dealt round-robin, each rank scores its own, one
        all-gather of the small host-side records (per-image match rows, per-batch loss items) rebuilds on every rank exactly the lists a single
        rank would hold — ultralytics validates on rank 0 alone, which at this trainer's step rate would leave N-1 GPUs idle for a time
        comparable to the epoch's training steps (0.3 s of 0.7 s on one GPU).  Eval-mode outputs do not depend on what else is in the batch, so
        the metrics equal the single-rank run's (tests/test_gpu_ddp_rehearsal.py).

        Two halves: the device half (forward, NMS, loss op, mask counts, matching — all enqueued, results copied to pinned host buffers by
        asynchronous copies) and the host half (per-image rows, the gather over ranks, AP curves: tens of milliseconds of NumPy).
        `deferred=True` returns the host half as a callable instead of running it: `fit()` hands it to every rank's writer thread, so the next
        epoch's steps are queued while the previous epoch's metrics are still being gathered and computed (the gather then runs on the trainer's
        own validation process group, to rank 0 only)."""
        from . import metrics as MT
        from .engine import InferEngine

        if self.val_ds is None or len(self.val_ds) == 0:
            return (lambda to_rank0=False: (np.zeros(4), None)) if deferred else (np.zeros(4), None)
        S, n = self.hyp["imgsz"], len(self.val_ds)
        limit = min(n, self.hyp.get("val_max") or n)
        sd = self.store.state_dict(p=self.ema_p, b=self.ema_b, on_device=True)  # EMA weights stay on the GPU: BN folding and packing run there
        if self._val_engine is None:  # one engine and its plans for the whole training; only the weights change from epoch to epoch
            self._val_engine = InferEngine(sd, self.store.scale, self.nc, self.dtype, str(self.device), conf=0.001, iou=0.7, max_det=300)
            self._val_loss_ops = {}
        else:
            self._val_engine.refresh(sd)
        eng, loss_ops = self._val_engine, self._val_loss_ops
        bounds = val_batches(limit, self.batch, self.world)
        mine = list(range(self.rank, len(bounds), self.world))
        items = torch.zeros(max(len(mine), 1), 4, dtype=torch.float32, device=self.device)  # per-batch loss items, one transfer at the end
        if self._ckpt is not None:
            self._ckpt.wait()  # the previous epoch's deferred host half still reads the pinned result buffers this pass is about to refill
        host = []  # per batch: (first slice index, the seven match tensors in pinned host memory)
        for k, bi in enumerate(mine):
            b0, b1 = bounds[bi]
            idx = list(range(b0, b1))
            nb_ = len(idx)
            plan = eng.plan(nb_, S, S)  # validation slices are letterboxed RGB at the training size: no LetterBox pass
            if self.val_aug is not None:
                batch = self.val_aug.batch(idx, None, mosaic=False, augment=False)
                plan.input.t.copy_(batch["img"].reshape(-1))
                gt, masks_d = batch["gt"], batch["masks"]
            else:
                batch = D.collate([D.plain(self.val_ds, i, S) for i in idx], S)
                plan.input.t.copy_(torch.from_numpy(batch["img"]).reshape(-1))
                gt, masks_d = device_targets(batch, nb_, S, S, self.device)
            plan.run()
            if nb_ not in loss_ops:
                lv = [plan.builder.levels[i] for i in sorted(plan.builder.levels)]
                loss_ops[nb_] = SegLossOp(lv, lv, plan.proto, plan.proto, self.nc, S, S, self.dtype, self.device)  # no_grad: the gradient views are never written
            items[k].copy_(loss_ops[nb_](gt, masks_d, no_grad=True)[:4])
            # metrics: mask areas / intersections counted by MSL_OP_MASK_IOU from the low-res logits, the whole batch matched on the device
            # (metrics.SegStats.match_on_device); the results go to pinned host buffers by asynchronous copies — no host synchronisation here
            mh, mw = plan.proto.H, plan.proto.W
            det = plan.det[:nb_]                                                    # [nb, max_det, 40]: xyxy, conf, cls, coefficients
            G = int(gt.shape[1])
            n_gt = (gt[..., 1:].sum(2) > 0).sum(1)
            counts = None
            if G:
                inter = torch.empty(nb_, det.shape[1], G, dtype=torch.int32, device=self.device)
                parea = torch.empty(nb_, det.shape[1], dtype=torch.int32, device=self.device)
                garea = torch.empty(nb_, G, dtype=torch.int32, device=self.device)
                hiplib.launch(hiplib.make_op(hiplib.OP_MASK_IOU, MSL_F32, p=(plan.lowres.data_ptr(), plan.det.data_ptr(), plan.keep_cnt.data_ptr(), masks_d.data_ptr(),
                                                                             inter.data_ptr(), parea.data_ptr(), garea.data_ptr()),
                                             i={0: nb_, 1: mh, 2: mw, 3: G, 7: det.shape[1], 8: S, 9: S}), torch.cuda.current_stream(self.device).cuda_stream)
                counts = (inter, parea, garea)
            matched = MT.SegStats.match_on_device(det[..., :4], det[..., 4], det[..., 5], None, plan.keep_cnt[:nb_].long(), gt[..., 1:5], gt[..., 0], None, n_gt,
                                                  mask_counts=counts)
            host.append((b0, tuple(self._pinned(("val", bi, j), t) for j, t in enumerate(matched))))
        items_h = self._pinned(("val", "items"), items)
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(self.device))
        n_mine, n_bounds = len(mine), len(bounds)

        def finish(to_rank0: bool = False):
            """Host half (may run in the writer thread): per-image rows, the merge over ranks, AP.  `to_rank0`: gather on the validation group to
            rank 0 only — the other ranks return (None, None) — instead of an all-gather on the default group."""
            # A rank whose host half fails still enters the gather — with the error as its payload — so that no peer waits for it forever; the
            # failure is raised on the ranks that receive it and (by this rank's wait()) in its own main thread.
            try:
                done.synchronize()
                stats = MT.SegStats()
                for b0, tensors in host:
                    stats.add_matched(tuple(t.numpy() for t in tensors), first_id=b0)
                mine_items = items_h[:n_mine].numpy().astype(np.float64)
                payload, failure = {"batches": mine, "items": mine_items, "stats": stats.export()}, None
            except Exception as e:  # noqa: BLE001 — reported, then re-raised below
                payload, failure = {"error": f"rank {self.rank}: {e!r}"}, e
            parts = gather_objects(payload, self.world, group=self._val_group if to_rank0 else None, to_rank0=to_rank0)
            if failure is not None:
                raise failure
            if parts is None:
                return None, None
            errors = [part["error"] for part in parts if "error" in part]
            if errors:
                raise RuntimeError("validation failed on " + "; ".join(errors))
            per_batch = np.zeros((n_bounds, 4))
            for part in parts:
                for bi, it in zip(part["batches"], part["items"]):
                    per_batch[bi] = it
            tot = np.zeros(4)
            for bi in range(n_bounds):  # summed in fold order whatever rank computed the batch
                tot += per_batch[bi]
            allstats = stats if self.world == 1 else MT.SegStats.merged([part["stats"] for part in parts])
            return tot / max(n_bounds, 1), allstats.result()

        return finish if deferred else finish()

    def _pinned(self, key, t: torch.Tensor) -> torch.Tensor:
        """Asynchronous device → pinned-host copy of `t` on the current stream (buffers cached per key and shape)."""
        buf = self._pin.get(key)
        if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
            buf = torch.empty(t.shape, dtype=t.dtype).pin_memory()
            self._pin[key] = buf
        buf.copy_(t, non_blocking=True)
        return buf

    # ------------------------------------------------------------------ files
    def _finish_epoch(self, writer, epoch: int, tl, lr: float, val_finish) -> None:
        """Host tail of an epoch (rank 0; in the writer thread when validation is deferred): validation's host half → results.csv row →
        last.pt every epoch, best.pt when the fitness improved [UPSTREAM BaseTrainer.save_model]."""
        vl, mets = val_finish()
        fitness = mets["fitness"] if mets else -float(tl.sum())
        mcols = [float(mets[c]) for c in RESULT_COLUMNS[6:14]] if mets else [0.0] * 8
        row = [epoch + 1, time.time() - self.t0] + [float(x) for x in tl] + mcols + [float(x) for x in vl] + [lr] * 3
        with open(self.save_dir / "results.csv", "a", newline="") as f:
            # as the reference's files: losses and metrics rounded to five decimals (2.41102, 0.36806, 1.2162), then every column with six significant digits
            row = row[:2] + [round(v, 5) for v in row[2:18]] + row[18:]
            f.write(",".join([str(row[0])] + [f"{v:.6g}" for v in row[1:]]) + "\n")
        is_best = self.best_fitness is None or fitness >= self.best_fitness
        if is_best:
            self.best_fitness = fitness
        extra = {"epoch": epoch, "best_fitness": self.best_fitness, "train_args": {"data": self.data_path, "epochs": self.epochs, "batch": self.batch,
                                                                                     "imgsz": self.hyp["imgsz"], "optimizer": self.optimizer, "lr0": self.lr0}}
        writer.write(extra, is_best)
        if self.verbose:
            print(f"epoch {epoch + 1}/{self.epochs} train {tl.round(4)} val {np.asarray(vl).round(4)} lr {lr:.6g}", flush=True)

    def _write_args(self) -> None:
        """args.yaml with the reference's keys and value types [REF trains/Base/FLAIR_P50c_5folds_50epochs/axial/fold1/args.yaml]: `optimizer` is what
        was asked for ('auto' unless overridden), `amp` a boolean; what this trainer resolved them to sits beside them (optimizer_resolved, amp_dtype)."""
        h = self.hyp
        args = dict(task="segment", mode="train", model=str(self.yolo.ckpt_path), data=self.data_path, epochs=self.epochs, batch=self.batch,
                    imgsz=h["imgsz"], cache=True, optimizer=h.get("optimizer") or "auto", seed=h["seed"], amp=self.dtype == MSL_BF16,
                    lr0=self.lr0, lrf=h["lrf"], momentum=h["momentum"], weight_decay=h["weight_decay"], warmup_epochs=h["warmup_epochs"],
                    warmup_momentum=h["warmup_momentum"], warmup_bias_lr=h["warmup_bias_lr"], nbs=h["nbs"],
                    close_mosaic=h["close_mosaic"], box=GAIN_BOX, cls=GAIN_CLS, dfl=GAIN_DFL, overlap_mask=True, mask_ratio=4,
                    hsv_h=D.HSV[0], hsv_s=D.HSV[1], hsv_v=D.HSV[2], translate=D.TRANSLATE, scale=D.SCALE, fliplr=D.FLIPLR, mosaic=1.0 if h["augment"] else 0.0,
                    # not reference keys: what this trainer resolved / added
                    optimizer_resolved=self.optimizer, amp_dtype="bf16" if self.dtype == MSL_BF16 else "fp32", device_augment=bool(h["device_augment"]),
                    world_size=self.world, val_max=h.get("val_max"), save_dir=str(self.save_dir))
        (self.save_dir / "args.yaml").write_text(yaml.safe_dump(args, sort_keys=False))

    # ------------------------------------------------------------------ main loop
    def fit(self):
        self.t0 = time.time()
        if self.world > 1 and self._val_group is None:
            # collective: every rank enters fit().  A bounded wait: a peer that died turns the writer threads' gather into an error instead of a hang
            self._val_group = torch.distributed.new_group(backend="gloo", timeout=datetime.timedelta(seconds=float(os.environ.get("MSLESSEG_VAL_GROUP_TIMEOUT_S", "600"))))
        if self.rank == 0:
            self._write_args()
            with open(self.save_dir / "results.csv", "w", newline="") as f:
                csv.writer(f).writerow(RESULT_COLUMNS)
        self.store.g.zero_()
        self.epoch_times = []  # per epoch: seconds of train steps / validation / checkpoint hand-off on this rank (bench.py --mode fit-epoch)
        ni, last_opt, done = 0, -1, False
        for epoch in range(self.epochs):
            te0 = time.perf_counter()
            tl_dev, nb_seen = torch.zeros(4, dtype=torch.float32, device=self.device), 0  # summed on the device: no host sync inside the epoch
            lr = self.sched.lr(ni, epoch)
            for batch in self._batches(epoch):
                lr = self.sched.lr(ni, epoch)
                items = self.forward_backward(batch, reduce_now=ni - last_opt >= self.sched.accumulate(ni))
                if ni - last_opt >= self.sched.accumulate(ni):
                    self.optimizer_step(lr)
                    last_opt = ni
                tl_dev += items
                nb_seen += 1
                ni += 1
                self.ni = ni
                if self.max_iters is not None and ni >= self.max_iters:
                    done = True
                    break
            te1 = time.perf_counter()
            if self._ckpt is not None:
                self._ckpt.wait()  # the previous epoch's host tail (it ran behind this epoch's steps) is done with the pinned buffers refilled below
            tl_h = self._pinned(("train", "items"), tl_dev)  # read by the host tail, after the event the writer waits for
            nb_done = max(nb_seen, 1)
            # every rank scores its share of the held-out fold: only the device half runs here; the host half (rows, gather to rank 0, AP, files)
            # rides in every rank's writer thread behind the next epoch's steps
            val_finish = self._validate(deferred=True)
            te2 = time.perf_counter()
            if self._ckpt is None:
                self._ckpt = CheckpointWriter(self)
            if self.rank == 0:
                self._ckpt.submit(lambda w, e=epoch, l=lr, f=val_finish, n=nb_done: self._finish_epoch(w, e, tl_h.numpy().astype(np.float64) / n, l, lambda: f(self.world > 1)))
            else:
                self._ckpt.submit(lambda w, f=val_finish: f(True), snapshot=False)
            self.epoch_times.append({"train_s": te1 - te0, "val_s": te2 - te1, "ckpt_s": time.perf_counter() - te2, "iterations": nb_seen})
            if done:
                break
        torch.cuda.synchronize(self.device)
        if self._ckpt is not None:
            self._ckpt.wait()
        if self.world > 1:
            torch.distributed.barrier()
        # the best checkpoint becomes the model's weights, like ultralytics' trainer reloads best.pt into the YOLO object after training
        best = self.wdir / "best.pt"
        self.yolo.state = params.load_checkpoint(best)["state"] if best.is_file() else self.store.state_dict(p=self.ema_p, b=self.ema_b)
        self.yolo.nc, self.yolo.names, self.yolo._engine = self.nc, self.names, None
        return self
