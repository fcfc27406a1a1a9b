"""Drop-in for the five ultralytics call sites of the reference:

  B1  YOLO(model_path)                                  [REF yolo_mslesseg/utils/utils.py:232-237]
  B2  model.train(data=, epochs=, batch=-1, cache=True, project=, name=, verbose=False)
                                                        [REF yolo_mslesseg/scripts/train.py:358-366]
  B3  model(img_array, verbose=False)[0]                [REF yolo_mslesseg/scripts/generar_predicciones.py:114]
  B4  .masks is None  |  .masks.data.cpu().numpy()      [REF generar_predicciones.py:118-120]

plus the batched entry points the reference's per-slice loop is replaced with (predict_slices / predict_volume).
Errors are plain Python exceptions (the reference wraps them in RuntimeError itself).  One model object per
process, synchronous, not re-entrant — same as the reference's usage.
"""
from __future__ import annotations

import logging
import os
import re
from pathlib import Path
from typing import List, Optional, Sequence, Union

import numpy as np
import torch

from . import geometry, params
from .hiplib import MSL_BF16, MSL_F32, MSL_F32S

LOGGER = logging.getLogger("ultralytics")
_NAME_RE = re.compile(r"^yolo11([nsmlx])-seg(\.pt|\.yaml)?$")


def _precision(p: Optional[str], env: str, default: str) -> int:
    p = (p or os.environ.get(env, default)).lower()
    if p in ("bf16", "bfloat16"):
        return MSL_BF16
    if p in ("fp32", "f32", "float32"):
        return MSL_F32
    if p in ("fp32s", "f32s", "split", "fp32-split"):
        return MSL_F32S
    raise ValueError(f"unknown precision {p!r} (bf16 | fp32 | fp32s)")


class _Staged(torch.Tensor):
    """A device tensor whose host copy has already landed in a pinned buffer of its own (engine.Plan.masks(stage_host=True)): `.cpu()` hands that copy out
    instead of starting a second, pageable transfer — the reference's `pred.masks.data.cpu().numpy()` [REF generar_predicciones.py:120] then costs
    nothing.  Everything else is the device tensor it aliases; tensors derived from it have no staged copy and take the ordinary path."""

    @staticmethod
    def wrap(dev: torch.Tensor, host: torch.Tensor) -> "_Staged":
        t = dev.as_subclass(_Staged)
        t._host = host
        return t

    def cpu(self, *args, **kwargs):
        host = getattr(self, "_host", None)
        if host is None or args or kwargs:
            return self.as_subclass(torch.Tensor).cpu(*args, **kwargs)
        self._host = None  # handed out once: like upstream's, every .cpu() gives the caller a tensor of its own (a second call transfers again)
        return host


class Masks:
    """`.data`: float32 [n, Hlb, Wlb] in {0,1} at the LETTERBOXED size (the reference resizes from there itself)."""

    def __init__(self, data: torch.Tensor, orig_shape):
        self.data, self.orig_shape = data, orig_shape

    def __len__(self):
        return int(self.data.shape[0])

    def cpu(self):
        return Masks(self.data.cpu(), self.orig_shape)

    def numpy(self):
        return self.data.cpu().numpy()


class Boxes:
    """`.data`: [n, 6] = xyxy (letterboxed pixels mapped back to the original image), conf, cls."""

    def __init__(self, data: torch.Tensor, orig_shape):
        self.data, self.orig_shape = data, orig_shape

    def __len__(self):
        return int(self.data.shape[0])

    @property
    def xyxy(self):
        return self.data[:, :4]

    @property
    def conf(self):
        return self.data[:, 4]

    @property
    def cls(self):
        return self.data[:, 5]


class Results:
    def __init__(self, orig_img, names, boxes: Optional[Boxes], masks: Optional[Masks], path=None):
        self.orig_img, self.orig_shape = orig_img, orig_img.shape[:2]
        self.names, self.boxes, self.masks, self.path = names, boxes, masks, path

    def __len__(self):
        return 0 if self.boxes is None else len(self.boxes)


def _scale_boxes(lb: geometry.LetterBox, xyxy: torch.Tensor) -> torch.Tensor:
    """[UPSTREAM ops.scale_boxes]: undo the letterbox (gain = min ratio, pad with the ±0.1 rounding), clip."""
    gain = min(lb.hlb / lb.h0, lb.wlb / lb.w0)
    pad_x = round((lb.wlb - lb.w0 * gain) / 2 - 0.1)
    pad_y = round((lb.hlb - lb.h0 * gain) / 2 - 0.1)
    out = xyxy.clone()
    out[:, [0, 2]] -= pad_x
    out[:, [1, 3]] -= pad_y
    out /= gain
    out[:, [0, 2]] = out[:, [0, 2]].clamp(0, lb.w0)
    out[:, [1, 3]] = out[:, [1, 3]].clamp(0, lb.h0)
    return out


class YOLO:
    def __init__(self, model: Union[str, Path] = "yolo11n-seg.pt", task=None, verbose: bool = False, precision: Optional[str] = None,
                 device: str = "cuda:0", train_precision: Optional[str] = None):
        """`precision` (or MSLESSEG_PRECISION): arithmetic of predict — default **fp32**, what the reference's `model(img)` runs (ultralytics
        predict, half=False) and the engine that reproduces the CPU path exactly (identical NMS indices, identical mask bytes on the trained demo
        checkpoint: tests/test_gpu_trained.py); "bf16" is the opt-in throughput mode (~3x the slices/s; |dDice| up to 1e-3 per plane volume,
        profiles/r02g_precision_trained_p39.json).  `train_precision` (or MSLESSEG_TRAIN_PRECISION): arithmetic of `.train()` — default
        **bf16** compute with fp32 master weights, the MI355X counterpart of the reference's `amp: true` [REF …/args.yaml:28].  An explicit
        `precision=` sets both unless `train_precision=` is given too."""
        self.ckpt_path = Path(model)
        self.task = task or "segment"
        self.device = device
        self.dtype = _precision(precision, "MSLESSEG_PRECISION", "fp32")
        self.train_dtype = _precision(train_precision or precision, "MSLESSEG_TRAIN_PRECISION", "bf16")
        if self.train_dtype == MSL_F32S:  # split-precision products exist for the predict kernels only: training under it runs the exact fp32 engine
            self.train_dtype = MSL_F32
        self.names = {0: "lesion"}
        self._engine = None
        self.trainer = None
        if self.ckpt_path.is_file():
            ck = params.load_checkpoint(self.ckpt_path)
            self.scale, self.nc, self.state = ck["scale"], int(ck["nc"]), ck["state"]
            self.names = {int(k): v for k, v in ck.get("names", {}).items()} or {i: str(i) for i in range(self.nc)}
            self.pretrained = True
        else:
            m = _NAME_RE.match(self.ckpt_path.name)
            if not m:
                raise FileNotFoundError(f"{model}: no such checkpoint")
            # The reference passes the bare name "yolo11n-seg.pt" and lets ultralytics download COCO weights
            # [REF ConfigTrain.py:139].  There is no network here and none is ever attempted: the model starts from
            # a seeded random initialisation instead.
            self.scale, self.nc, self.state = m.group(1), 80, None
            self.pretrained = False
            LOGGER.warning(f"{model} not found: starting from random initialisation (no download is attempted)")

    # ------------------------------------------------------------------ inference
    def _get_engine(self):
        if self._engine is None:
            from .engine import InferEngine

            if self.state is None:
                self.state = params.init_state(self.scale, self.nc, seed=0)
            self._engine = InferEngine({k: (v.float() if v.is_floating_point() else v) for k, v in self.state.items()},
                                       self.scale, self.nc, self.dtype, self.device)
        return self._engine

    @staticmethod
    def _load_image(src) -> np.ndarray:
        if isinstance(src, (str, Path)):
            if str(src).lower().endswith(".png"):  # the reference's slices are PNGs: own decoder, cv2.imread channel order (BGR)
                from .pngio import read_bgr

                return read_bgr(src)
            from PIL import Image  # other formats only if Pillow happens to be installed

            rgb = np.asarray(Image.open(src).convert("RGB"))
            return np.ascontiguousarray(rgb[..., ::-1])
        arr = np.asarray(src)
        if arr.ndim == 2:
            arr = np.repeat(arr[..., None], 3, axis=2)
        if arr.ndim != 3 or arr.shape[2] != 3 or arr.dtype != np.uint8:
            raise TypeError(f"expected uint8 HxWx3 BGR array, got {arr.dtype} {arr.shape}")
        return np.ascontiguousarray(arr)

    def predict(self, source, verbose: bool = False, **kwargs) -> List[Results]:
        srcs = list(source) if isinstance(source, (list, tuple)) else [source]
        imgs = [self._load_image(s) for s in srcs]
        eng = self._get_engine()
        results: List[Optional[Results]] = [None] * len(imgs)
        by_shape = {}
        for i, im in enumerate(imgs):
            by_shape.setdefault(im.shape, []).append(i)
        for shape, idxs in by_shape.items():
            batch = torch.from_numpy(imgs[idxs[0]][None] if len(idxs) == 1 else np.stack([imgs[i] for i in idxs]))  # one slice: a view, not a copy
            # the per-slice call of the reference replays the network program as a hipGraph (captured on the first call of a shape; bit-identical to the
            # eager program, tests/test_gpu_e2e.py::test_graph_replay_equals_eager): ~110 launches become one, 0.1 ms of 1.6 per slice
            plan = eng.predict_batch(batch, graph_replay=len(idxs) == 1 and os.environ.get("MSLESSEG_PREDICT_GRAPH", "1") != "0")
            # two host synchronisations per call: the counts + detection rows (the wait for the network), then the masks with their host copy and
            # "not empty" flags (engine.Plan.masks); everything else below is host arithmetic on what those two brought back
            cnt, det = plan.counts_and_rows()
            masks = plan.masks(stage_host=True, cnt=cnt)
            lb = geometry.letterbox_for(shape[0], shape[1])
            for j, i in enumerate(idxs):
                n = int(cnt[j])
                if n == 0:
                    results[i] = Results(imgs[i], self.names, None, None)
                    continue
                rows, (mk, mk_host, live) = det[j, :n].clone(), masks[j]
                # instances whose mask came out empty are dropped together with their boxes, as the oracle's reading of 8.3.70's
                # construct_result does (oracle/prepost.py postprocess_one); invisible after the reference's np.maximum merge, visible in len()
                if not bool(live.any()):
                    results[i] = Results(imgs[i], self.names, None, None)
                    continue
                if not bool(live.all()):
                    rows, mk, mk_host = rows[live], mk[live.to(mk.device)], None if mk_host is None else mk_host[live]
                boxes = torch.cat([_scale_boxes(lb, rows[:, :4]), rows[:, 4:6]], 1)
                results[i] = Results(imgs[i], self.names, Boxes(boxes, shape[:2]), Masks(mk if mk_host is None else _Staged.wrap(mk, mk_host), shape[:2]),
                                     path=srcs[i] if isinstance(srcs[i], (str, Path)) else None)
        return results  # type: ignore[return-value]

    __call__ = predict

    def predict_slices(self, imgs: Union[np.ndarray, torch.Tensor]) -> np.ndarray:
        """uint8 [N,H,W,3|1] slices of one shape → uint8 [N,W,H] in {0,255}: exactly what the reference writes to
        pred_masks/*.png per slice [REF generar_predicciones.py:175-187], for the whole batch in one pass."""
        t = torch.from_numpy(np.ascontiguousarray(imgs)) if isinstance(imgs, np.ndarray) else imgs
        return self._get_engine().predict_slices(t).cpu().numpy()

    # ------------------------------------------------------------------ training (B2)
    def train(self, data=None, epochs: int = 100, batch: int = -1, cache: bool = True, project=None, name: str = "train",
              verbose: bool = False, **kwargs):
        from .train import Trainer  # noqa: PLC0415  (imports the training kernels lazily)

        self.trainer = Trainer(self, data=data, epochs=epochs, batch=batch, cache=cache, project=project, name=name, verbose=verbose, **kwargs)
        return self.trainer.fit()

    def save(self, path) -> None:
        if self.state is None:
            self.state = params.init_state(self.scale, self.nc, seed=0)
        params.save_checkpoint(path, self.state, self.scale, self.nc, self.names)
