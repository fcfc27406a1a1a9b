"""YOLO11s-seg (BASELINE configs[2]) on the GPU (-m gpu).  The s widths (32 ... 512 channels) take dispatch branches the n model never reaches:
two-chunk / four-chunk persistent 3x3 forms, the 1x1 streaming kernel's Cout > 256 split, other weight-gradient slot widths, a larger
weight-pack arena.  Same bars as for n: the fp32 engine against the oracle (forward) and the oracle's autograd (training), the bf16 engine
against the fp32 engine.  Weights: calibrated random (oracle/synth.py), built here — no s checkpoint exists anywhere."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mslesseg_amd import engine as E  # noqa: E402
from mslesseg_amd.hiplib import MSL_BF16, MSL_F32  # noqa: E402
from test_gpu_train import _oracle_run, _probe, _run_plan  # noqa: E402


@pytest.fixture(scope="module")
def s_state(demo_volumes):
    from oracle import prepost as P
    from oracle import synth

    torch.set_num_threads(16)
    fl = demo_volumes["P39_flair"]
    calib = [P.slice_to_png_array(P.take_slice(fl, "axial", i)) for i in (60, 90, 120)]
    m = synth.calibrated_model(calib, "s", 1, seed=0)
    return {k: (v.float() if v.is_floating_point() else v) for k, v in synth.state_to_bf16(m).items()}


def test_s_forward_fp32_matches_oracle_and_bf16_batch_is_consistent(s_state, demo_volumes):
    from oracle import prepost as P
    from oracle import synth

    om = synth.model_from_state(s_state, scale="s")
    img = P.slice_to_png_array(P.take_slice(demo_volumes["P39_flair"], "axial", 90))
    x = P.preprocess(img)
    with torch.no_grad():
        y, proto = om(x)
    e32 = E.InferEngine(s_state, "s", 1, MSL_F32)
    plan = e32.predict_batch(torch.from_numpy(img[None]))
    torch.cuda.synchronize()
    got = plan.head_tensor().cpu()
    err = float(((got - y).abs() / (1.0 + y.abs())).max())
    perr = float(((plan.proto.torch().float().cpu().permute(0, 3, 1, 2) - proto).abs() / (1.0 + proto.abs())).max())
    assert err < 2e-3 and perr < 2e-3, (err, perr)
    _, idx = P.non_max_suppression(y, nc=1)
    n = int(plan.keep_cnt.cpu()[0])
    assert n == len(idx[0]) and set(plan.keep_idx.cpu()[0, :n].tolist()) == set(idx[0].tolist())
    want = P.generar_prediccion_2D(om, img)
    out = plan.merged(*img.shape[:2]).cpu().numpy()[0]
    assert int((out != want).sum()) <= max(3, int(2e-4 * out.size))
    # bf16, 16-slice batch (persistent weights-resident 3x3 kernels, streaming 1x1 with Cout > 256): finite, copies of a slice agree bit for bit,
    # the stem is one bf16 rounding away from the oracle
    e16 = E.InferEngine(s_state, "s", 1, MSL_BF16)
    p16 = e16.predict_batch(torch.from_numpy(np.stack([img] * 16)))
    torch.cuda.synchronize()
    h16 = p16.head_tensor().float().cpu()
    assert bool(torch.isfinite(h16).all()) and torch.equal(h16[0], h16[7]) and torch.equal(h16[0], h16[15])
    with torch.no_grad():
        ref0 = om.model[0](x)
    got0 = p16.builder.taps["model.0"].torch().float().cpu().permute(0, 3, 1, 2)[:1]
    assert float((got0 - ref0).norm() / ref0.norm()) < 4e-3
    sc = float((h16[0, 4] - y[0, 4]).abs().mean())
    assert sc < 0.05, sc


def test_s_train_forward_backward_fp32_matches_oracle_autograd(s_state):
    rng = np.random.default_rng(1)
    N, H, W = 2, 64, 96
    img = rng.integers(0, 256, size=(N, H, W, 3), dtype=np.uint8)
    R, shapes = _probe(N, H, W)
    feats, mc, p, grads, bufs = _oracle_run(s_state, img, R, scale="s")
    store, plan, fw = _run_plan(s_state, img, R, shapes, MSL_F32, scale="s")
    perr = float((fw["proto"] - p.permute(0, 2, 3, 1).detach()).abs().max() / (1 + p.abs().max()))
    assert perr < 1e-4
    gsd = store.state_dict(p=store.g)
    floor = 1e-5 * max(float(v.abs().max()) for v in grads.values())
    worst = sorted(((float((gsd[k] - ref).abs().max()) / (float(ref.abs().max()) + floor), k) for k, ref in grads.items() if k != "model.23.dfl.conv.weight"), reverse=True)
    assert worst[0][0] < 2e-3, worst[:6]


def test_s_trainer_step_bf16_640_batch16_against_fp32_engine():
    """The benchmarked shape family at scale s: 640x640, batch 16, bf16 kernels (LDS 3x3 / persistent / streaming 1x1 / transposed-read wgrad at
    the s widths) against the fp32 engine from the same seeded initial weights.  An UNTRAINED network amplifies the bf16 rounding of its
    activations from layer to layer (measured against the oracle's autograd at 640x640, tests/tools/dev_grad_diag.py s_init: fp32 engine cosine
    0.999999, bf16 engine 0.82 — already 0.93 one layer below the probe), so the whole-network bound here is loose; the s-width kernels are
    checked one by one against PyTorch references in tests/test_gpu_ops.py (the "YOLO11s-seg widths" cases) and the fp32 engine above."""
    from mslesseg_amd import data as D
    from mslesseg_amd import params
    from mslesseg_amd.train import Trainer
    from mslesseg_amd.yolo import YOLO

    ds = D.SyntheticSegDataset(16, 640, seed=3)
    batch = D.collate([D.plain(ds, i, 640) for i in range(16)], 640)
    st = params.init_state("s", 1, seed=0)
    out = {}
    for name, dt in (("fp32", MSL_F32), ("bf16", MSL_BF16)):
        y = YOLO.__new__(YOLO)
        y.ckpt_path, y.task, y.device, y.names, y._engine, y.trainer = "init-s", "segment", "cuda:0", {0: "lesion"}, None, None
        y.dtype = y.train_dtype = dt
        y.scale, y.nc, y.state, y.pretrained = "s", 1, st, True
        tr = Trainer(y, dataset=ds, val_dataset=None, epochs=1, batch=16, project="gpurun_out/test_runs", name=f"s_{name}", nbs=16, warmup_epochs=0.0, augment=False)
        assert tr.store.n > 10_000_000
        tr.store.g.zero_()
        items = tr.forward_backward(batch).cpu().numpy()
        g = tr.store.g.clone().cpu()
        assert np.isfinite(items).all() and bool(torch.isfinite(g).all())
        before = tr.store.p.clone()
        tr.optimizer_step(tr.lr0)
        assert bool(torch.isfinite(tr.store.p).all()) and float((tr.store.p - before).abs().max()) > 0
        out[name] = (items, g)
        del tr
        torch.cuda.empty_cache()
    (i32, g32), (i16, g16) = out["fp32"], out["bf16"]
    cos = float((g16 @ g32) / (g16.norm() * g32.norm()))
    print(f"scale s train step bf16 vs fp32: items {i16} vs {i32}, flat gradient cosine {cos:.5f}")
    assert np.allclose(i16, i32, rtol=5e-2, atol=1e-3) and cos > 0.6
