"""Dev: bf16 vs fp32 engine vs oracle error statistics (run on the GPU box)."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "yolo-mslesseg_amd")]
from mslesseg_amd import engine as E
from mslesseg_amd.hiplib import MSL_BF16, MSL_F32
from oracle import prepost as P, synth

st = torch.load(ROOT / "tests/golden/synth_n_nc1.pt", map_location="cpu", weights_only=True)
st = {k: (v.float() if v.is_floating_point() else v) for k, v in st.items()}
om = synth.model_from_state(st)
g = np.load(ROOT / "tests/golden/e2e_golden.npz")
e32, e16 = E.InferEngine(st, "n", 1, MSL_F32), E.InferEngine(st, "n", 1, MSL_BF16)
for k in range(5):
    img = np.ascontiguousarray(np.repeat(g[f"img{k}"][..., None], 3, axis=2))
    x = P.preprocess(img)
    with torch.no_grad():
        y, proto = om(x)
    outs = {}
    for name, e in (("f32", e32), ("bf16", e16)):
        plan = e.predict_batch(torch.from_numpy(img[None])); torch.cuda.synchronize()
        h = plan.head_tensor().cpu()
        n = int(plan.keep_cnt.cpu()[0]); keep = set(plan.keep_idx.cpu()[0, :n].tolist())
        out = plan.merged(*img.shape[:2]).cpu().numpy()[0]
        outs[name] = (h, keep, out)
        rows, idx = P.non_max_suppression(y, nc=1)
        ko = set(idx[0].tolist())
        want = P.generar_prediccion_2D(om, img)
        print(f"case{k} {name}: box mean|err| {float((h[:, :4]-y[:, :4]).abs().mean()):.4f} max {float((h[:, :4]-y[:, :4]).abs().max()):.3f} | "
              f"score mean|err| {float((h[:, 4]-y[:, 4]).abs().mean()):.5f} max {float((h[:, 4]-y[:, 4]).abs().max()):.4f} | "
              f"coef rel {float((h[:, 5:]-y[:, 5:]).abs().mean()/y[:, 5:].abs().mean()):.4f} | kept {n} vs {len(ko)} common {len(keep & ko)} | "
              f"mask px diff {int((out != want).sum())}/{out.size} dice {2*((out>0)&(want>0)).sum()/((out>0).sum()+(want>0).sum()+1e-8):.5f}")
    if k == 0:
        m = om.model
        with torch.no_grad():
            refs = {"model.0": m[0](x)}
            refs["model.1"] = m[1](refs["model.0"]); refs["model.2.cv2"] = m[2](refs["model.1"]); refs["model.3"] = m[3](refs["model.2.cv2"])
            refs["model.4.cv2"] = m[4](refs["model.3"]); o6 = m[6](m[5](refs["model.4.cv2"])); refs["model.6.cv2"] = o6
            o8 = m[8](m[7](o6)); refs["model.8.cv2"] = o8; refs["model.9.cv2"] = m[9](o8); refs["model.10.cv2"] = m[10](refs["model.9.cv2"])
        for name, e in (("f32", e32), ("bf16", e16)):
            plan = e.plan(1, 640, 544)
            for ln, ref in refs.items():
                got = plan.builder.taps[ln].torch().float().cpu().permute(0, 3, 1, 2)
                print(f"   {name} {ln}: rel L2 err {float((got-ref).norm()/ref.norm()):.2e}  ref rms {float(ref.pow(2).mean().sqrt()):.3f}")
