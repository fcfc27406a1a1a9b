"""Parity on a TRAINED network and real FLAIR slices (-m gpu): the checkpoint tests/golden/demo_p39_n.pt was trained by this library's own
trainer on the lesion slices of the reference's demo patient P39 (tests/golden/make_demo_checkpoint.py; 80 epochs, mosaic augmentation from the
device feeder, bf16 train engine, every 5th slice held out: mask mAP50 0.75 on those; weights stored bf16-exact) and is the well-conditioned counterpart of the calibrated-random test weights.

north_star tolerance (BASELINE.json): bit-exact indices after NMS, reconstructed-volume Dice within 1e-4 of the CPU reference.
  * fp32 engine (the default of `YOLO()` predict, = the reference's half=False): identical ordered kept-index lists on every slice, identical
    output bytes, |dDice| <= 1e-4 per plane volume and for the 3-plane consensus  -> asserted at exactly that tolerance.
  * bf16 engine (opt-in throughput mode): bf16 storage rounds every activation to 8 significant bits (measured 0.2-0.5 % relative L2 at every
    tap of this network, no growth with depth), which moves near-threshold scores and mask-boundary logits: measured on all 361 lesion slices
    (profiles/r02g_precision_trained_p39.json) 333 identical kept lists, 644 of 13.4 M output bytes differ, |dDice| 4.1e-4 / 1.9e-4 / 1.1e-3
    per plane, 2.0e-4 for the consensus.  It does NOT meet the 1e-4 tolerance; the bounds asserted here are those measured ones with headroom
    (this test predicts every 3rd slice only, which makes the Dice of the partial volumes more sensitive: up to 2.4e-3), and DESIGN.md says so.
The oracle's model arithmetic restates ultralytics 8.3.70 (parity unpinned against the reference itself)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mslesseg_amd import engine as E  # noqa: E402
from mslesseg_amd import volume as V  # noqa: E402
from mslesseg_amd.hiplib import MSL_BF16, MSL_F32, MSL_F32S  # noqa: E402

STRIDE = 3  # every 3rd slice of the reference's slice set (121 of the 361 lesion slices) keeps the oracle's CPU time to a few seconds


@pytest.fixture(scope="module")
def trained_state(golden_dir):
    st = torch.load(golden_dir / "demo_p39_n.pt", map_location="cpu", weights_only=True)
    return {k: (v.float() if v.is_floating_point() else v) for k, v in st.items()}


@pytest.fixture(scope="module")
def oracle_run(trained_state, demo_volumes):
    """Oracle results per plane: slice indices, rendered slices, kept-index lists, output bytes, plane volume, Dice."""
    from oracle import prepost as P
    from oracle import synth

    torch.set_num_threads(16)
    om = synth.model_from_state(trained_state)
    fl, gt = demo_volumes["P39_flair"], demo_volumes["P39_mask"]
    out = {}
    for plano in ("axial", "coronal", "sagital"):
        idx = V.select_slices(gt, plano)[::STRIDE]
        imgs = np.stack([V.slice_as_png_array(V.take_slice(fl, plano, i)) for i in idx])
        kept, outs = [], []
        for im in imgs:
            x = P.preprocess(im)
            with torch.no_grad():
                y, proto = om(x)
            rows, k = P.non_max_suppression(y, nc=1)
            kept.append(k[0].tolist())
            m = P.postprocess_one(rows[0], proto[0], tuple(x.shape[2:]))
            outs.append(P.normalizar_prediccion(P.combinar_predicciones([] if m is None else m.numpy(), im.shape[:2])))
        vol = P.reconstruir_volumen(dict(zip(idx, outs)), gt.shape, plano)
        out[plano] = dict(idx=idx, imgs=imgs, kept=kept, outs=outs, vol=vol, dice=P.dsc_unrounded(gt, vol))
    return out


def _engine_run(eng, oracle_run, gt):
    from oracle import prepost as P

    res = {}
    for plano, o in oracle_run.items():
        same_list = same_set = px = 0
        outs = []
        for b0 in range(0, len(o["idx"]), 64):
            chunk = o["imgs"][b0 : b0 + 64]
            plan = eng.predict_batch(torch.from_numpy(chunk))
            out = plan.merged(*chunk.shape[1:3]).cpu().numpy()
            cnt, kid = plan.keep_cnt.cpu().numpy(), plan.keep_idx.cpu().numpy()
            for j in range(len(chunk)):
                got = kid[j, : cnt[j]].tolist()
                same_list += got == o["kept"][b0 + j]
                same_set += set(got) == set(o["kept"][b0 + j])
                px += int((out[j] != o["outs"][b0 + j]).sum())
            outs += list(out)
        vol = P.reconstruir_volumen(dict(zip(o["idx"], outs)), gt.shape, plano)
        res[plano] = dict(n=len(o["idx"]), same_list=same_list, same_set=same_set, px=px, total=int(o["imgs"].shape[0] * o["imgs"].shape[1] * o["imgs"].shape[2]),
                          vol=vol, dice=P.dsc_unrounded(gt, vol))
    return res


@pytest.mark.parametrize("mode", ["fp32", "fp32s"])
def test_fp32_engine_meets_the_north_star_tolerance_on_trained_weights(trained_state, oracle_run, demo_volumes, mode):
    """The north-star contract (bit-exact kept indices after NMS, <= 2 output bytes, |dDice| <= 1e-4 per plane volume and for the consensus) for
    the exact fp32 engine (the predict default) AND for the split-precision mode MSL_F32S (fp32 tensors, every conv product as three f16 partial
    products on the matrix cores): the same assertions, nothing relaxed."""
    from oracle import prepost as P

    gt = demo_volumes["P39_mask"]
    assert sum(len(k) for o in oracle_run.values() for k in o["kept"]) > 300, "the trained network must detect lesions for this test to mean anything"
    res = _engine_run(E.InferEngine(trained_state, "n", 1, MSL_F32 if mode == "fp32" else MSL_F32S), oracle_run, gt)
    for plano, r in res.items():
        print(f"{mode} {plano}: {r['same_list']}/{r['n']} identical kept lists, {r['px']} of {r['total']} bytes differ, dice {r['dice']:.6f} oracle {oracle_run[plano]['dice']:.6f}")
        assert r["same_list"] == r["n"], f"{plano}: kept indices differ on {r['n'] - r['same_list']} slices"
        assert r["px"] <= 2, f"{plano}: {r['px']} output bytes differ"
        assert abs(r["dice"] - oracle_run[plano]["dice"]) <= 1e-4
    cons = P.combinar_volumenes(res["axial"]["vol"], res["coronal"]["vol"], res["sagital"]["vol"], 2)
    cons_o = P.combinar_volumenes(oracle_run["axial"]["vol"], oracle_run["coronal"]["vol"], oracle_run["sagital"]["vol"], 2)
    assert abs(P.dsc_unrounded(gt, cons) - P.dsc_unrounded(gt, cons_o)) <= 1e-4
    assert min(o["dice"] for o in oracle_run.values()) > 0.25  # a trained model: with every 3rd lesion slice predicted, each plane volume alone overlaps a third of the GT


def test_bf16_engine_measured_deviation_on_trained_weights(trained_state, oracle_run, demo_volumes):
    from oracle import prepost as P

    gt = demo_volumes["P39_mask"]
    res = _engine_run(E.InferEngine(trained_state, "n", 1, MSL_BF16), oracle_run, gt)
    for plano, r in res.items():
        dd = abs(r["dice"] - oracle_run[plano]["dice"])
        print(f"bf16 {plano}: {r['same_list']}/{r['n']} identical kept lists, {r['same_set']} identical sets, {r['px']} of {r['total']} bytes differ, |dDice| {dd:.2e}")
        assert r["same_set"] >= 0.8 * r["n"] and r["same_list"] >= 0.6 * r["n"]
        assert r["px"] <= 3e-4 * r["total"]
        assert dd <= 5e-3, "bf16 is the throughput mode: bounded, not at the 1e-4 tolerance (module docstring)"
    cons = P.combinar_volumenes(res["axial"]["vol"], res["coronal"]["vol"], res["sagital"]["vol"], 2)
    cons_o = P.combinar_volumenes(oracle_run["axial"]["vol"], oracle_run["coronal"]["vol"], oracle_run["sagital"]["vol"], 2)
    assert abs(P.dsc_unrounded(gt, cons) - P.dsc_unrounded(gt, cons_o)) <= 3e-3


def test_default_yolo_predicts_in_fp32_and_whole_volume_dice_on_device(trained_state, oracle_run, demo_volumes, tmp_path, monkeypatch):
    """`YOLO(path)` as the reference constructs it (no precision argument) → fp32 predict; the batched whole-volume path on the reference's
    slice set (`select_slices`) gives the oracle's Dice within 1e-4, computed on the device."""
    from ultralytics import YOLO

    from mslesseg_amd import params

    monkeypatch.delenv("MSLESSEG_PRECISION", raising=False)
    ck = tmp_path / "best.pt"
    params.save_checkpoint(ck, trained_state, "n", 1, {0: "lesion"})
    model = YOLO(ck)
    assert model.dtype == MSL_F32 and model.train_dtype == MSL_BF16
    fl, gt = demo_volumes["P39_flair"], demo_volumes["P39_mask"]
    o = oracle_run["coronal"]
    vol = V.predict_volume(model, fl, "coronal", indices=o["idx"])
    assert int((vol.cpu().numpy() != o["vol"]).sum()) <= 2
    d, d3 = V.dice(torch.from_numpy(gt).to(vol.device), vol.to(torch.uint8))
    assert abs(d - o["dice"]) <= 1e-4 and d3 == round(d, 3)
    # B3/B4 on one slice: the boundary returns one mask per kept instance at the letterboxed size
    r = model(o["imgs"][len(o["imgs"]) // 2], verbose=False)[0]
    k = o["kept"][len(o["imgs"]) // 2]
    assert (r.masks is None and not k) or (r.masks is not None and 0 < len(r.masks) <= len(k))


def test_bf16_train_step_against_the_fp32_engine_on_trained_weights(trained_state, demo_volumes):
    """One real training step (16 P39 slices, mosaic batch from the device feeder, the segmentation loss) in both engines from the same trained
    weights: the bf16 engine's loss items and flat gradient against the fp32 engine's (which tests/test_gpu_train.py pins to the oracle's
    autograd)."""
    from mslesseg_amd import data as D
    from mslesseg_amd.train import Trainer
    from mslesseg_amd.yolo import YOLO

    ds = D.VolumeSliceDataset(demo_volumes["P39_flair"], demo_volumes["P39_mask"], keep=lambda plano, i: i % 8 == 0)
    out = {}
    for name, dt in (("fp32", MSL_F32), ("bf16", MSL_BF16)):
        y = YOLO.__new__(YOLO)
        y.ckpt_path, y.task, y.device, y.names, y._engine, y.trainer = "trained", "segment", "cuda:0", {0: "lesion"}, None, None
        y.dtype = y.train_dtype = dt
        y.scale, y.nc, y.state, y.pretrained = "n", 1, trained_state, True
        tr = Trainer(y, dataset=ds, val_dataset=None, epochs=1, batch=16, project="gpurun_out/test_runs", name=f"trained_{name}", nbs=16, warmup_epochs=0.0)
        batch = tr.aug.batch(list(range(16)), np.random.default_rng(3), mosaic=True, augment=True)
        tr.store.g.zero_()
        items = tr.forward_backward(batch).cpu().numpy()
        out[name] = (items, tr.store.g.clone().cpu(), tr.store)
        assert np.isfinite(items).all()
    (i32, g32, store), (i16, g16, _) = out["fp32"], out["bf16"]
    rel_items = np.abs(i16 - i32) / np.abs(i32)
    g16, g32 = g16.double(), g32.double()
    cos = float((g16 @ g32) / (g16.norm() * g32.norm()))
    rel = float((g16 - g32).norm() / g32.norm())
    per = []
    for key, (off, shp) in store.entries.items():
        n = int(np.prod(shp))
        a, b = g16[off : off + n], g32[off : off + n]
        if float(b.norm()) > 1e-3 * float(g32.norm()):
            per.append((float((a @ b) / (a.norm() * b.norm() + 1e-30)), key))
    per.sort()
    print(f"bf16 vs fp32 train step on trained weights: loss items rel err {rel_items}, flat gradient cosine {cos:.5f} rel L2 {rel:.4f}, worst tensors {per[:4]}")
    # measured (r02f): loss items within 4e-3, flat gradient rel L2 1.6 %, worst tensor cosine 0.964
    assert rel_items.max() < 2e-2
    assert cos > 0.999 and rel < 0.05
    assert per[0][0] > 0.9, per[:6]


def test_predict_variants_mixed_work_list_equals_single_variant_runs(trained_state, demo_volumes, golden_dir, tmp_path):
    """BASELINE configs[4]: a mixed list of (volume, variant, plane) items, one model per enhancement variant [REF ConfigPred.py:150-166,
    mejora_imagen.py:43-184], dealt over ranks without a collective: every item's plane volume equals the single-variant path, each item is
    predicted by exactly one rank, and the variant really selects the weights and the slice rendering."""
    from ultralytics import YOLO

    from mslesseg_amd import params

    synth = torch.load(golden_dir / "synth_n_nc1.pt", map_location="cpu", weights_only=True)
    paths = {}
    for name, st in (("GC", trained_state), ("HE", {k: (v.float() if v.is_floating_point() else v) for k, v in synth.items()})):
        paths[name] = tmp_path / name / "weights" / "best.pt"
        params.save_checkpoint(paths[name], st, "n", 1, {0: "lesion"})
    models = {k: YOLO(p) for k, p in paths.items()}
    fl, gt = demo_volumes["P39_flair"], demo_volumes["P39_mask"]
    idx = {pl: V.select_slices(gt, pl, 12) for pl in ("axial", "coronal", "sagital")}
    items = [(fl, "GC", "axial", idx["axial"]), (fl, "HE", "coronal", idx["coronal"]), (fl, "GC", "sagital", idx["sagital"]), (fl, "HE", "axial", idx["axial"])]
    single = [V.predict_volume(models[m], f, pl, ii, mejora=m).cpu() for f, m, pl, ii in items]
    seen = [0] * len(items)
    for world in (1, 2):
        for rank in range(world):
            outs = V.predict_variants(models, items, rank=rank, world=world)
            for k, o in enumerate(outs):
                if o is not None:
                    assert torch.equal(o.cpu(), single[k]), (world, rank, k)
                    seen[k] += world == 2
    assert seen == [1, 1, 1, 1]
    assert not torch.equal(single[0], V.predict_volume(models["HE"], fl, "axial", idx["axial"], mejora="GC").cpu())  # the weights matter
    with pytest.raises(KeyError):
        V.predict_variants(models, [(fl, "LT", "axial", idx["axial"])], rank=0, world=1, strict=True)


def test_a_bad_item_is_logged_and_skipped_like_the_reference_skips_a_patient(trained_state, demo_volumes, tmp_path, caplog):
    """The reference wraps every patient in try / except → logger.warning("… se omite") → continue [REF scripts/generar_predicciones.py:289-301,
    reconstruir_volumen.py:297-306]; the batched paths keep that: one wrong-shaped volume, one variant without a model and one out-of-range
    slice index in the list leave their entries SKIPPED (falsy, distinct from the None of another rank's item), the good items are still predicted and equal the undisturbed run."""
    import logging

    from ultralytics import YOLO

    from mslesseg_amd import params

    path = tmp_path / "GC" / "weights" / "best.pt"
    params.save_checkpoint(path, trained_state, "n", 1, {0: "lesion"})
    models = {"GC": YOLO(path)}
    fl, gt = demo_volumes["P39_flair"], demo_volumes["P39_mask"]
    idx = V.select_slices(gt, "axial", 6)
    good = (fl, "GC", "axial", idx)
    items = [good, (fl[:, :, 0], "GC", "axial", None), (fl, "LT", "coronal", idx), (fl, "GC", "sagital", [10 ** 6]), good]
    want = V.predict_volume(models["GC"], fl, "axial", idx, mejora="GC").cpu()
    with caplog.at_level(logging.WARNING, logger="ultralytics"):
        outs = V.predict_variants(models, items, rank=0, world=1)
    assert outs[1] is V.SKIPPED and outs[2] is V.SKIPPED and outs[3] is V.SKIPPED and not outs[1]  # a marker of its own: None means "another rank's item"
    assert torch.equal(outs[0].cpu(), want) and torch.equal(outs[4].cpu(), want)
    assert sum("se omite" in r.getMessage() for r in caplog.records) == 3
    # the per-patient loop around the three-plane consensus
    plane_models = {pl: models["GC"] for pl in ("axial", "coronal", "sagital")}
    sel = {pl: V.select_slices(gt, pl, 4) for pl in ("axial", "coronal", "sagital")}
    caplog.clear()
    with caplog.at_level(logging.WARNING, logger="ultralytics"):
        res = V.predict_patients(plane_models, [("P39", fl), ("Pbad", np.zeros((4, 4))), ("P39b", fl)], indices={"P39": sel, "P39b": sel})
    assert res["Pbad"] is V.SKIPPED and sum("Pbad" in r.getMessage() for r in caplog.records) == 1
    assert torch.equal(res["P39"][0], res["P39b"][0]) and res["P39"][0].dtype == torch.uint8
    with pytest.raises(ValueError):
        V.predict_patients(plane_models, [("Pbad", np.zeros((4, 4)))], strict=True)


def test_validator_mask_counts_op_equals_the_dense_formulation(trained_state, demo_volumes):
    """MSL_OP_MASK_IOU (per-prediction mask area and intersection with every GT instance, from the low-res logits) against the dense form it
    replaces: binary [predictions, 160*160] masks multiplied with one-hot ground-truth masks — on real slices, conf 0.001 (hundreds of boxes)."""
    from mslesseg_amd import augment as A
    from mslesseg_amd import data as D
    from mslesseg_amd import hiplib

    ds = D.VolumeSliceDataset(demo_volumes["P39_flair"], demo_volumes["P39_mask"], keep=lambda plano, i: i % 16 == 0)
    B = min(len(ds), 12)
    aug = A.DeviceAugmenter(A.SliceCache(ds, "cuda:0"), 640)
    batch = aug.batch(list(range(B)), None, mosaic=False, augment=False)
    eng = E.InferEngine(trained_state, "n", 1, MSL_BF16, conf=0.001, iou=0.7, max_det=300)
    plan = eng.plan(B, 640, 640)
    plan.input.t.copy_(batch["img"].reshape(-1))
    plan.run()
    gt, labels = batch["gt"], batch["masks"]
    G, mh, mw, P = int(gt.shape[1]), plan.proto.H, plan.proto.W, 300
    assert G >= 2 and int(plan.keep_cnt.max()) > 20
    inter = torch.full((B, P, G), -1, dtype=torch.int32, device="cuda:0")
    parea = torch.full((B, P), -1, dtype=torch.int32, device="cuda:0")
    garea = torch.full((B, G), -1, dtype=torch.int32, device="cuda:0")
    hiplib.launch(hiplib.make_op(hiplib.OP_MASK_IOU, MSL_F32, p=(plan.lowres.data_ptr(), plan.det.data_ptr(), plan.keep_cnt.data_ptr(), labels.data_ptr(), inter.data_ptr(), parea.data_ptr(), garea.data_ptr()),
                                 i={0: B, 1: mh, 2: mw, 3: G, 7: P, 8: 640, 9: 640}), torch.cuda.current_stream().cuda_stream)
    ys = torch.arange(mh, device="cuda:0", dtype=torch.float32)[None, None, :, None]
    xs = torch.arange(mw, device="cuda:0", dtype=torch.float32)[None, None, None, :]
    bl = plan.det[..., :4] * (mw / 640)
    inbox = (xs >= bl[..., 0, None, None]) & (xs < bl[..., 2, None, None]) & (ys >= bl[..., 1, None, None]) & (ys < bl[..., 3, None, None])
    valid = (torch.arange(P, device="cuda:0")[None] < plan.keep_cnt[:, None])
    pm = ((plan.lowres > 0) & inbox & valid[:, :, None, None]).float().reshape(B, P, mh * mw)
    gm = (labels[:, None] == torch.arange(1, G + 1, device="cuda:0")[None, :, None, None]).float().reshape(B, G, mh * mw)
    want_inter = torch.bmm(pm, gm.transpose(1, 2)).round().int()
    assert torch.equal(inter, want_inter) and torch.equal(parea, pm.sum(2).round().int()) and torch.equal(garea, gm.sum(2).round().int())
    assert int(want_inter.sum()) > 100
