#!/usr/bin/env python3
"""Engines (fp32 / bf16 / mixed) against the CPU oracle on a TRAINED checkpoint and real FLAIR slices (run on the GPU box).

    python tests/tools/precision_report.py --ckpt tests/golden/demo_p39_n.pt --out gpurun_out/precision.json [--stride 1] [--modes fp32,bf16]

For every plane of demo patient P39: the reference's slice set (`volume.select_slices`), rendered slices, then per engine
  kept-index agreement with the oracle (identical ordered list / same set / count), bytes of the merged re-oriented mask that differ,
  Dice of the reconstructed plane volume against GT (un-rounded) and |dDice| against the oracle's, and the 3-plane consensus Dice.
The oracle is test infrastructure (fp32 PyTorch-CPU restatement); nothing here feeds the product path."""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
for p in (str(ROOT), str(ROOT / "yolo-mslesseg_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ckpt", default=str(ROOT / "tests" / "golden" / "demo_p39_n.pt"))
    ap.add_argument("--scale", default="n")
    ap.add_argument("--stride", type=int, default=1, help="use every stride-th slice of the reference's slice set")
    ap.add_argument("--modes", default="fp32,bf16")
    ap.add_argument("--out", default=str(ROOT / "gpurun_out" / "precision.json"))
    ap.add_argument("--layers", action="store_true", help="per-layer relative L2 error on the first axial slice")
    args = ap.parse_args()
    from mslesseg_amd import engine as E
    from mslesseg_amd import volume as V
    from oracle import prepost as P
    from oracle import synth

    st = torch.load(args.ckpt, map_location="cpu", weights_only=True)
    st = {k: (v.float() if v.is_floating_point() else v) for k, v in st.items()}
    torch.set_num_threads(16)
    om = synth.model_from_state(st, scale=args.scale)
    z = np.load(ROOT / "tests" / "golden" / "demo_volumes.npz")
    shape = tuple(int(v) for v in z["P39_shape"])
    gt = np.unpackbits(z["P39_mask_bits"])[: int(np.prod(shape))].reshape(shape).astype(np.uint8)
    fl = z["P39_flair_u16"].astype(np.float64)
    engines = {m: E.InferEngine(st, args.scale, 1, E.precision_code(m)) for m in args.modes.split(",")}
    rep = {"ckpt": Path(args.ckpt).name, "planes": {}, "modes": list(engines)}
    vols = {m: {} for m in ["oracle"] + list(engines)}
    for plano in ("axial", "coronal", "sagital"):
        idx = V.select_slices(gt, plano)[:: args.stride]
        imgs = np.stack([V.slice_as_png_array(V.take_slice(fl, plano, i)) for i in idx])
        t0 = time.time()
        o_idx, o_out = [], []
        for im in imgs:
            x = P.preprocess(im)
            with torch.no_grad():
                y, proto = om(x)
            rows, kept = P.non_max_suppression(y, nc=1)
            o_idx.append(kept[0].tolist())
            m = P.postprocess_one(rows[0], proto[0], tuple(x.shape[2:]))
            o_out.append(P.normalizar_prediccion(P.combinar_predicciones([] if m is None else m.numpy(), im.shape[:2])))
        t_or = time.time() - t0
        vols["oracle"][plano] = P.reconstruir_volumen(dict(zip(idx, o_out)), gt.shape, plano)
        d_o = P.dsc_unrounded(gt, vols["oracle"][plano])
        rec = {"slices": len(idx), "oracle_dice": d_o, "oracle_kept_total": sum(len(k) for k in o_idx), "oracle_s": round(t_or, 1)}
        for mname, eng in engines.items():
            same_list = same_set = same_cnt = 0
            px = 0
            outs = []
            for b0 in range(0, len(idx), 64):
                chunk = imgs[b0 : b0 + 64]
                plan = eng.predict_batch(torch.from_numpy(chunk))
                out = plan.merged(*chunk.shape[1:3]).cpu().numpy()
                cnt, kid = plan.keep_cnt.cpu().numpy(), plan.keep_idx.cpu().numpy()
                for j in range(len(chunk)):
                    got = kid[j, : cnt[j]].tolist()
                    want = o_idx[b0 + j]
                    same_list += got == want
                    same_set += set(got) == set(want)
                    same_cnt += len(got) == len(want)
                    px += int((out[j] != o_out[b0 + j]).sum())
                outs += list(out)
            vols[mname][plano] = P.reconstruir_volumen(dict(zip(idx, outs)), gt.shape, plano)
            d = P.dsc_unrounded(gt, vols[mname][plano])
            rec[mname] = {"identical_kept_lists": same_list, "identical_kept_sets": same_set, "identical_counts": same_cnt, "bytes_differing": px,
                          "bytes_total": int(imgs.shape[0] * imgs.shape[1] * imgs.shape[2]), "dice": d, "abs_ddice_vs_oracle": abs(d - d_o),
                          "voxels_differing": int((vols[mname][plano] != vols["oracle"][plano]).sum())}
        rep["planes"][plano] = rec
        print(plano, json.dumps(rec), flush=True)
    cons = {m: P.combinar_volumenes(v["axial"], v["coronal"], v["sagital"], 2) for m, v in vols.items()}
    d_o = P.dsc_unrounded(gt, cons["oracle"])
    rep["consensus"] = {"oracle_dice": d_o, **{m: {"dice": P.dsc_unrounded(gt, cons[m]), "abs_ddice_vs_oracle": abs(P.dsc_unrounded(gt, cons[m]) - d_o),
                                                   "voxels_differing": int((cons[m] != cons["oracle"]).sum())} for m in engines}}
    print("consensus", json.dumps(rep["consensus"]), flush=True)
    if args.layers:
        im = V.slice_as_png_array(V.take_slice(fl, "axial", V.select_slices(gt, "axial")[len(V.select_slices(gt, "axial")) // 2]))
        x = P.preprocess(im)
        m = om.model
        with torch.no_grad():
            refs = {"model.0": m[0](x)}
            refs["model.1"] = m[1](refs["model.0"])
            refs["model.2.cv2"] = m[2](refs["model.1"])
            refs["model.3"] = m[3](refs["model.2.cv2"])
            refs["model.4.cv2"] = m[4](refs["model.3"])
            refs["model.6.cv2"] = m[6](m[5](refs["model.4.cv2"]))
            refs["model.8.cv2"] = m[8](m[7](refs["model.6.cv2"]))
            refs["model.9.cv2"] = m[9](refs["model.8.cv2"])
            refs["model.10.cv2"] = m[10](refs["model.9.cv2"])
            y, proto = om(x)
        rep["layers"] = {}
        for mname, eng in engines.items():
            plan = eng.predict_batch(torch.from_numpy(im[None]))
            torch.cuda.synchronize()
            d = {}
            for ln, ref in refs.items():
                if ln in plan.builder.taps:
                    got = plan.builder.taps[ln].torch().float().cpu().permute(0, 3, 1, 2)
                    d[ln] = float((got - ref).norm() / ref.norm())
            h = plan.head_tensor().cpu()
            d["head.box_max_abs_px"] = float((h[:, :4] - y[:, :4]).abs().max())
            d["head.score_max_abs"] = float((h[:, 4] - y[:, 4]).abs().max())
            d["proto.rel_l2"] = float((plan.proto.torch().float().cpu().permute(0, 3, 1, 2) - proto).norm() / proto.norm())
            rep["layers"][mname] = d
            print(mname, json.dumps(d), flush=True)
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
