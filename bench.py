#!/usr/bin/env python3
"""bench.py — FLAIR slices/sec of the YOLO11n-seg hot path on MI355X (contract: task brief / DESIGN.md §Measurement).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Default workload = BASELINE.json configs[1]: **YOLO11n-seg TRAIN step, 640x640 bf16, 128 slices per GPU** (north_star: batch >= 128).  A "step" = one optimisation step over
one batch of synthetic slices already resident in HBM: weight pack → HIP forward (train-mode BatchNorm) → segmentation loss
→ HIP backward → [all-reduce of the flat gradient over ranks] → gradient clip + fused AdamW + EMA.  Nothing is skipped.
Slices shard over ranks (data parallel); the only collective is the one gradient all-reduce → "scaling": "weak".
`--mode predict` times the inference leg instead (LetterBox → net → NMS → masks → merged uint8 slices); in train mode a short
predict run in both arithmetic modes (fp32 = the exact default of YOLO() predict, bf16 = the throughput mode) is reported under "infer".
`--mode train-e2e` feeds the same step from the data feeder on real FLAIR slices (mosaic on) and reports the resident-batch rate of the same
trainer beside it; `--mode replicas` runs independent trainings without a collective (the reference's fold x plane jobs); `--scale s` switches to
YOLO11s-seg (BASELINE configs[2]).  Rank 0 prints ONE JSON line.  The oracle is imported only for cpu_baseline.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for p in (str(ROOT), str(ROOT / "yolo-mslesseg_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

METRIC = "FLAIR slices/sec train+infer at 1/2/4/8 GPU; Dice vs GT volumes"  # BASELINE.json "metric"
PEAK = {"bf16": 2500.0, "fp32": 157.3,  # dense MFMA TFLOP/s, /opt/skills/guides/MI355X_MICROARCH.md §Chip-level parameters
        "fp32s": 2500.0 / 3}  # split-precision products: three f16 matrix instructions per algorithmic product
PEAK_HBM_GBS = 8000.0
FWD_GFLOP_PER_SLICE_640 = 9.630  # n, nc=1, 640x640, 2*MAC over conv + attention + ConvT; training = 3x
FWD_GFLOP = {"n": 9.630, "s": 32.90}  # SURVEY §8d, nc=1, 640x640


def synthetic_slices(n, h, w, seed):
    """uint8 [n,h,w,3], 3 equal channels: rng.integers(0,256) low-passed by a 3x3 box."""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, size=(n, h + 2, w + 2)).astype(np.float32)
    acc = np.zeros((n, h, w), np.float32)
    for dy in range(3):
        for dx in range(3):
            acc += a[:, dy : dy + h, dx : dx + w]
    g = np.clip(np.rint(acc / 9.0), 0, 255).astype(np.uint8)
    return np.ascontiguousarray(np.repeat(g[..., None], 3, axis=3))


def load_weights(scale="n"):
    """n: the calibrated-random test weights (tests/golden/synth_n_nc1.pt); s (BASELINE configs[2]): seeded random initialisation — there is no
    checkpoint of that scale anywhere and train-mode BatchNorm uses batch statistics anyway."""
    if scale != "n":
        from mslesseg_amd import params

        return {k: (v.float() if v.is_floating_point() else v) for k, v in params.init_state(scale, 1, seed=0).items()}
    st = torch.load(ROOT / "tests" / "golden" / "synth_n_nc1.pt", map_location="cpu", weights_only=True)
    return {k: (v.float() if v.is_floating_point() else v) for k, v in st.items()}


def dist_setup(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # MSLESSEG_DIST_BACKEND=gloo: rehearsal of the multi-rank control flow on a box with fewer GPUs than ranks (several ranks share a GPU, the
    # collective stages through host memory); the measured configuration is always RCCL ("nccl"), one rank per GPU
    backend = os.environ.get("MSLESSEG_DIST_BACKEND", "nccl")
    local = local % max(torch.cuda.device_count(), 1) if backend == "gloo" else local
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    return world, rank, dev, dist


def timed(step, args, dev, world, dist):
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


# ---------------------------------------------------------------------------------------------------- predict leg
def predict_setup(args, dev, rank, state, B):
    from mslesseg_amd import engine as E
    from mslesseg_amd.hiplib import MSL_BF16, MSL_F32

    from mslesseg_amd.hiplib import MSL_F32S

    dtype = {"bf16": MSL_BF16, "fp32": MSL_F32, "fp32s": MSL_F32S}[args.dtype]
    S = args.size
    host = synthetic_slices(B, S, S, seed=rank)
    imgs = torch.from_numpy(host).to(dev)
    bias_shift = 0.0
    if args.target_kept > 0:
        # Real MS slices carry a handful of lesions; the calibrated-random weights keep 300 boxes on every noise slice.
        # Bisect a class-bias shift (weights only — conf/iou/max_det stay the reference's) on a 16-slice probe batch.
        lo, hi = -60.0, 0.0
        for _ in range(14):
            mid = 0.5 * (lo + hi)
            st = dict(state)
            for i in range(3):
                st[f"model.23.cv3.{i}.2.bias"] = state[f"model.23.cv3.{i}.2.bias"] + mid
            probe = E.InferEngine(st, args.scale, 1, dtype, str(dev))
            kept = float(probe.predict_batch(imgs[:16]).keep_cnt.float().mean().item())
            del probe
            lo, hi = (mid, hi) if kept < args.target_kept else (lo, mid)
        bias_shift = 0.5 * (lo + hi)
        state = dict(state)
        for i in range(3):
            state[f"model.23.cv3.{i}.2.bias"] = state[f"model.23.cv3.{i}.2.bias"] + bias_shift
        torch.cuda.empty_cache()
    eng = E.InferEngine(state, args.scale, 1, dtype, str(dev))
    return eng, imgs, host, state, bias_shift


def predict_roofline(eng, imgs, out, args, value_per_gpu):
    from mslesseg_amd import hiplib

    B, S = imgs.shape[0], args.size
    plan = eng.plan(B, S, S)
    table = plan.time_ops(reps=3)
    costs = [plan.op_cost(i) for i in range(len(table))]
    total_ms = sum(t[2] for t in table)
    conv = [i for i in range(len(table)) if table[i][1] == hiplib.OP_CONV]
    conv_ms, conv_flops = sum(table[i][2] for i in conv), sum(costs[i][0] for i in conv)
    dom = max(conv, key=lambda i: table[i][2])
    d_name, _, d_ms = table[dom]
    d_flops, d_bytes = costs[dom]
    peak = PEAK[args.dtype]
    ach = d_flops / (d_ms * 1e-3) / 1e12
    ev = [hiplib.Event() for _ in range(3)]
    lbp, _ = eng.letterbox(B, S, S, 3)
    st_ = torch.cuda.current_stream(imgs.device).cuda_stream
    ev[0].record(st_)
    lbp.run()
    ev[1].record(st_)
    plan.merged(S, S, out=out)
    ev[2].record(st_)
    torch.cuda.synchronize(imgs.device)
    roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None,
            "kernel": f"conv3x3_pers_kernel/conv3x3_lds_kernel/conv_igemm_kernel<{args.dtype}> @ {d_name}", "launch_ms": round(d_ms, 4),
            "algorithmic_gflop_per_launch": round(d_flops / 1e9, 3), "algorithmic_hbm_gbs": round(d_bytes / (d_ms * 1e-3) / 1e9, 1),
            "hbm_frac": round(d_bytes / (d_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
            "all_conv": {"tflops": round(conv_flops / (conv_ms * 1e-3) / 1e12, 2), "frac": round(conv_flops / (conv_ms * 1e-3) / 1e12 / peak, 4), "ms": round(conv_ms, 3),
                         "share_of_program": round(conv_ms / total_ms, 3)},
            "whole_net_frac_of_mfma_roof": round(value_per_gpu * FWD_GFLOP[args.scale] * (S * S / 640.0 / 640.0) / 1e3 / peak, 4),
            "step_breakdown": {"letterbox_ms": round(ev[0].elapsed_ms(ev[1]), 4), "mask_merge_ms": round(ev[1].elapsed_ms(ev[2]), 4), "program_ms": round(total_ms, 3),
                               "mean_kept_instances_per_slice": round(float(plan.keep_cnt.float().mean().item()), 1),
                               "mask_on_fraction": round(float((out > 0).float().mean().item()), 3)}}
    if args.op_table:
        with open(args.op_table, "w") as f:
            f.write(f"# predict: per-op HIP-event times, batch {B}, {S}x{S}, {args.dtype}; total {total_ms:.3f} ms\n")
            for (name, kind, ms), (fl, by) in zip(table, costs):
                f.write(f"{name:34s} kind={kind:2d} {ms:9.4f} ms  {fl / 1e9:9.3f} GFLOP  {fl / (ms * 1e-3) / 1e12 if ms > 0 else 0:8.2f} TF/s  "
                        f"{by / 1e6:9.2f} MB  {by / (ms * 1e-3) / 1e9 if ms > 0 else 0:8.1f} GB/s\n")
    return roof


def cpu_baseline_predict(state, imgs_u8, seconds_budget=20.0, scale="n"):
    """Restated CPU path (ultralytics-on-CPU cannot run here: SURVEY §8d), batch 1 per slice like the reference's loop."""
    from oracle import prepost as P
    from oracle import synth

    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    om = synth.model_from_state(state, scale=scale)
    P.generar_prediccion_2D(om, imgs_u8[0])
    t0, n = time.perf_counter(), 0
    while (time.perf_counter() - t0) < seconds_budget and n < 4096:  # bounded sample: ~20 s of host work, cycling over the slices
        P.generar_prediccion_2D(om, imgs_u8[n % len(imgs_u8)])
        n += 1
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 3), "unit": "slices/s", "cores": cores, "kind": "port",
            "sample": f"{n} synthetic 640x640 slices, batch 1, fp32, oracle restatement of the reference predict loop in {dt:.1f}s"}


# ---------------------------------------------------------------------------------------------------- train leg
def train_setup(args, dev, rank, world, state, B):
    from mslesseg_amd import data as D
    from mslesseg_amd.hiplib import MSL_BF16, MSL_F32
    from mslesseg_amd.train import Trainer
    from mslesseg_amd.yolo import YOLO

    y = bench_model(args, dev, state)
    ds = D.SyntheticSegDataset(B, args.size, seed=rank)
    tr = Trainer(y, dataset=ds, val_dataset=None, epochs=1, batch=B, project=ROOT / "gpurun_out" / "bench_runs", name=f"r{rank}",
                 imgsz=args.size, nbs=B * world, warmup_epochs=0.0, replica=(args.mode == "replicas"))
    batch = D.collate([D.plain(ds, i, args.size) for i in range(B)], args.size)
    dbatch = tr.to_device(batch)
    return tr, dbatch, batch


def train_roofline(tr, dbatch, args, value_per_gpu):
    from mslesseg_amd import hiplib

    plan = tr.plan
    tr.forward_backward(dbatch)  # leaves every buffer in a consistent state for per-op replay
    tr.store.g.zero_()
    rows = []
    for tag, segs in (("pack", [plan.pack_program]), ("fwd", plan.forward_segments), ("bwd", plan.backward_segments)):
        rows += [(tag,) + r for r in plan.time_segments(segs, reps=2)]
    tr.store.g.zero_()
    by_kind = {}
    for tag, what, kind, ms, it in rows:
        by_kind[(tag, kind)] = by_kind.get((tag, kind), 0.0) + ms
    names = {v: k for k, v in vars(hiplib).items() if k.startswith("OP_") and isinstance(v, int)}

    def conv_flops(op, wgrad=False):
        I = op.i
        if wgrad:
            return 2.0 * I[0] * I[4] * I[5] * I[6] * I[7] * I[7] * I[3]
        if I[20] == 3:  # all parity classes of a stride-2 input gradient in one launch: 9 tap-GEMMs per GRADIENT pixel (H, W), not per output pixel
            return 2.0 * I[0] * I[1] * I[2] * I[6] * 9 * I[3]
        return 2.0 * I[0] * I[4] * I[5] * I[6] * I[16]

    es = 2 if args.dtype == "bf16" else 4

    def conv_bytes(op, wgrad=False):
        """Algorithmic HBM bytes of one launch: every input activation read once, every output written once (weights and
        the partial-sum scratch are noise next to the activations); an accumulating input gradient also reads what it adds to."""
        I = op.i
        a_in, a_out = I[0] * I[1] * I[2] * I[3] * es, I[0] * I[4] * I[5] * I[6] * es
        if wgrad:
            return float(a_in + a_out * (2 if I[19] else 1))  # x and dz (fp32 dz: twice the bytes)
        return float(a_in + a_out * (2 if op.p[3] else 1))

    def op_bytes(kind, op):
        """Algorithmic HBM bytes of any op of the train programs: activation tensors it reads and writes, each once (weights, statistics and
        scratch are noise beside them).  Elementwise ops: accesses x N*H*W*C*element size."""
        I = op.i
        if kind == hiplib.OP_CONV:
            return conv_bytes(op)
        if kind == hiplib.OP_CONV_WGRAD:
            return conv_bytes(op, True)
        t = float(I[0]) * I[1] * I[2] * I[3] * es
        acc = {hiplib.OP_BN_ACT: 2 + (1 if op.p[3] else 0), hiplib.OP_BN_ACT_BWD_REDUCE: 2, hiplib.OP_BN_ACT_BWD_APPLY: 3, hiplib.OP_BN_STATS: 1, hiplib.OP_DWCONV: 2,
               hiplib.OP_DW_WGRAD: 2, hiplib.OP_COLSUM: 1, hiplib.OP_ADD_VIEW: 3, hiplib.OP_UPSAMPLE2X: 5, hiplib.OP_UPSAMPLE2X_BWD: 5, hiplib.OP_SPPF_POOL: 4,
               hiplib.OP_SPPF_POOL_BWD: 5}.get(kind)
        if acc is not None:
            return t * acc
        if kind == hiplib.OP_STEM:
            return float(I[0]) * I[1] * I[2] * 3 + float(I[0]) * I[4] * I[5] * I[6] * es
        if kind == hiplib.OP_STEM_WGRAD:
            return float(I[0]) * I[1] * I[2] * 3 + float(I[0]) * I[4] * I[5] * I[6] * es
        return 0.0  # attention, loss, packing, drains: small tensors

    step_bytes = sum(op_bytes(kind, it) for tag, what, kind, ms, it in rows if what == "op")
    wg_ms, wg_fl, cv_ms, cv_fl = 0.0, 0.0, 0.0, 0.0
    cands = []  # every MFMA launch as (ms, flops, kernel label, dtype, tag, op); the top 3 by time are replayed for PMC
    for tag, what, kind, ms, it in rows:
        I = it.i if what == "op" else None
        if kind == hiplib.OP_CONV_WGRAD:
            fl = conv_flops(it, True)
            wg_ms, wg_fl = wg_ms + ms, wg_fl + fl
            # same dispatch rule as msl_launch_conv_wgrad (train_kernels.hip): bf16 tensors, 3x3/p1 (s1|s2) | 2x2/p0/s2 | 1x1/p0/s1, 8-aligned views
            geom = (I[7] == 3 and I[9] == 1 and I[8] in (1, 2)) or (I[7] == 1 and I[9] == 0 and I[8] == 1) or (I[7] == 2 and I[9] == 0 and I[8] == 2)
            trk = it.dtype == hiplib.MSL_BF16 and not I[19] and geom and all(I[j] % 8 == 0 for j in (3, 10, 11, 12, 13))
            cands.append((ms, fl, "conv_wgrad_tr_kernel (bf16 MFMA 16x16x32, LDS transposed reads, pixel contraction)" if trk else "conv_wgrad_kernel (fp32 MFMA 16x16x4, pixel contraction)",
                          "bf16" if trk else "fp32", tag, it))
        elif kind == hiplib.OP_CONV:
            fl = conv_flops(it)
            cv_ms, cv_fl = cv_ms + ms, cv_fl + fl
            one = I[7] == 1 and I[8] == 1 and (I[20] == 0 or (I[20] == 1 and I[6] % 32 == 0)) and args.dtype == "bf16" and I[3] % 8 == 0 and I[6] % 8 == 0 and I[6] <= 256
            kn = "conv3x3_lds_kernel" if I[25] else ("conv1x1_kernel" if one else "conv_igemm_kernel")
            if I[25] and I[20] == 3:
                kn = "conv_s2dgrad_lds_kernel"
            elif I[25]:  # same rule as msl_launch_conv3x3_lds: persistent weights-resident form for full 64-channel blocks over <= 2 whole chunks
                chunk = 32 if args.dtype == "bf16" else 16
                tiles = I[0] * ((I[5] + 31) // 32) * ((I[4] + 7) // 8)
                if I[8] == 1 and I[24] == 4 and I[3] % chunk == 0 and I[3] // chunk <= 2 and tiles >= 1024 and I[23] not in (-8, -4):
                    kn = "conv3x3_pers_kernel"
            mode = "input gradient, transposed-conv gather" if I[22] else ("input gradient" if tag == "bwd" else "forward")
            cands.append((ms, fl, f"{kn}<{args.dtype}> ({mode})", args.dtype, tag, it))
    cands.sort(key=lambda c: -c[0])

    def roof_ms(c):  # time of the launch on the ideal machine: the slower of its two roofs
        by = conv_bytes(c[5], c[5].kind == hiplib.OP_CONV_WGRAD)
        return max(c[1] / (PEAK[c[3]] * 1e12), by / (PEAK_HBM_GBS * 1e9)) * 1e3

    # dominant kernel = the launch with the most work by the roofline (the largest ideal time), not the one that happens to be slowest today
    ms, fl, kname, kdt, tag, op = max(cands, key=roof_ms)
    peak = PEAK[kdt]
    ach = fl / (ms * 1e-3) / 1e12
    by = conv_bytes(op, op.kind == hiplib.OP_CONV_WGRAD)
    ach_gbs = by / (ms * 1e-3) / 1e9
    hbm_bound = by / (PEAK_HBM_GBS * 1e9) > fl / (peak * 1e12)
    I = op.i
    total = sum(r[3] for r in rows)
    lms, lfl, lname, ldt, _, lop = cands[0]
    lby = conv_bytes(lop, lop.kind == hiplib.OP_CONV_WGRAD)
    roof = {"bound": "hbm" if hbm_bound else "mfma", "achieved": round(ach_gbs if hbm_bound else ach, 2), "peak": PEAK_HBM_GBS if hbm_bound else peak,
            "unit": "GB/s" if hbm_bound else "TFLOP/s", "frac": round(ach_gbs / PEAK_HBM_GBS if hbm_bound else ach / peak, 4), "traffic": None,
            "mfma_tflops": round(ach, 2), "mfma_frac": round(ach / peak, 4), "hbm_gbs": round(ach_gbs, 1), "hbm_frac": round(ach_gbs / PEAK_HBM_GBS, 4),
            "arithmetic_intensity_flop_per_byte": round(fl / by, 1), "ridge_flop_per_byte": round(peak * 1e12 / (PEAK_HBM_GBS * 1e9), 1),
            "algorithmic_gbytes_per_launch": round(by / 1e9, 4),
            "dominant_rule": "the MFMA launch with the largest roofline time max(flop / peak, bytes / 8 TB/s); 'bound' names the roof that sets it",
            "longest_launch": {"kernel": lname, "launch_ms": round(lms, 4), "mfma_frac": round(lfl / (lms * 1e-3) / 1e12 / PEAK[ldt], 4),
                               "hbm_frac": round(lby / (lms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                               "shape": {"N": lop.i[0], "H": lop.i[1], "W": lop.i[2], "Cin": lop.i[3], "Ho": lop.i[4], "Wo": lop.i[5], "Cout": lop.i[6], "k": lop.i[7], "stride": lop.i[8]}},
            "kernel": kname, "kernel_dtype": kdt, "launch_ms": round(ms, 4), "algorithmic_gflop_per_launch": round(fl / 1e9, 3),
            "launch_shape": {"N": I[0], "H": I[1], "W": I[2], "Cin": I[3], "Ho": I[4], "Wo": I[5], "Cout": I[6], "k": I[7], "stride": I[8]},
            "all_wgrad": {"tflops": round(wg_fl / (wg_ms * 1e-3) / 1e12, 2), "frac": round(wg_fl / (wg_ms * 1e-3) / 1e12 / PEAK[args.dtype], 4), "ms": round(wg_ms, 3)},
            "all_conv_fwd_dgrad": {"tflops": round(cv_fl / (cv_ms * 1e-3) / 1e12, 2), "frac": round(cv_fl / (cv_ms * 1e-3) / 1e12 / PEAK[args.dtype], 4), "ms": round(cv_ms, 3)},
            "program_ms": {"total_fwd_bwd_pack": round(total, 3),
                           **{f"{t}:{names.get(k, 'torch-attention')}": round(v, 3) for (t, k), v in sorted(by_kind.items(), key=lambda x: -x[1])[:14]}},
            "whole_step_frac_of_bf16_mfma_roof": round(value_per_gpu * 3 * FWD_GFLOP[args.scale] * (args.size * args.size / 640.0 / 640.0) / 1e3 / PEAK["bf16"], 4),
            # the whole step against the HBM roof: algorithmic bytes of every op of the forward + backward programs (each activation tensor an op reads or
            # writes, once) over the measured step time
            "step_bytes": round(step_bytes), "step_hbm_gbs": round(step_bytes / (tr.batch / value_per_gpu) / 1e9, 1),
            "step_hbm_frac": round(step_bytes / (tr.batch / value_per_gpu) / 1e9 / PEAK_HBM_GBS, 4)}
    def shape_of(o):
        J = o.i
        return {"N": J[0], "H": J[1], "W": J[2], "Cin": J[3], "Ho": J[4], "Wo": J[5], "Cout": J[6], "k": J[7], "stride": J[8]}

    pmc = ROOT / "profiles" / "pmc_latest.json"  # written by scripts/pmc_traffic.py from rocprofv3 --pmc passes over --replay-dominant
    if pmc.exists():
        for rec in json.loads(pmc.read_text()).get("records", []):
            if rec.get("kernel") == kname and rec.get("launch_shape") == roof["launch_shape"]:
                roof["traffic"] = rec["traffic_bytes_per_launch"]
                roof["traffic_note"] = rec.get("note", "")
                if rec.get("mfma_util") is not None:  # SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles) of the same replayed launches
                    roof["mfma_util"] = round(rec["mfma_util"], 4)
                    roof["mfma_util_counters"] = {k: rec["sq"][k] for k in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "kernel_cycles") if k in rec["sq"]}
                roof["traffic_source"] = "profiles/pmc_latest.json: a committed rocprofv3 --pmc summary of an earlier run of this kernel and shape (scripts/pmc_traffic.py), not collected in this run"
    ins = ROOT / "profiles" / "in_stream_latest.json"  # scripts/in_stream.py: the same launch inside whole steps (rocprofv3 --kernel-trace), not replayed alone
    if ins.exists():
        for rec in json.loads(ins.read_text()).get("records", []):
            if rec.get("kernel") == kname and rec.get("launch_shape") == roof["launch_shape"]:
                roof["launch_ms_in_stream"] = rec["launch_ms_in_stream"]
                a_in = (by / (rec["launch_ms_in_stream"] * 1e-3) / 1e9) if hbm_bound else (fl / (rec["launch_ms_in_stream"] * 1e-3) / 1e12)
                roof["frac_in_stream"] = round(a_in / (PEAK_HBM_GBS if hbm_bound else peak), 4)
                roof["in_stream_source"] = ("profiles/in_stream_latest.json: mean duration of this launch in a committed rocprofv3 --kernel-trace of whole steps, where the "
                                            "deferred weight-gradient lanes and the head lanes share the device with it; `launch_ms` / `frac` are the launch replayed alone")
    if args.replay_dominant > 0:  # for the PMC passes: the three longest MFMA launches, each alone and back to back, as the LAST dispatches of the process
        s_ = torch.cuda.current_stream(tr.device).cuda_stream
        torch.cuda.synchronize(tr.device)
        roof["replay"] = []
        picks, fams = list(cands[:3]), set()  # the three longest launches + the longest of every kernel family (ranks shift a little under the profiler)
        for c in cands:
            fam = (c[2].split("<")[0].split(" ")[0], c[4])
            if fam not in fams:
                fams.add(fam)
                if c not in picks:
                    picks.append(c)
        # + the launches furthest below their roofline by the op table's excess column (round-3 verdict: the small-map / stride-2 weight gradients had no counter record)
        for c in cands:
            J = c[5].i
            if c[5].kind == hiplib.OP_CONV_WGRAD and (J[1], J[3], J[4], J[6], J[7], J[8]) in ((80, 128, 40, 128, 3, 2), (20, 256, 20, 64, 3, 1)) and c not in picks:
                picks.append(c)
        # a sentinel launch (a 256-element EMA: `ema_kernel`, which no replayed op uses) before every record lets scripts/pmc_traffic.py cut
        # the dispatch list into one segment per record without trusting the kernel labels
        sent = torch.zeros(512, dtype=torch.float32, device=tr.device)
        sentinel = hiplib.make_op(hiplib.OP_EMA, hiplib.MSL_F32, p=(sent.data_ptr(), sent[256:].data_ptr()), i={0: 256}, f=(0.5,))
        for c in picks:
            hiplib.launch(sentinel, s_)
            for _ in range(args.replay_dominant):
                hiplib.launch(c[5], s_)
            roof["replay"].append({"kernel": c[2], "launch_shape": shape_of(c[5]), "launch_ms": round(c[0], 4)})
        torch.cuda.synchronize(tr.device)
    if args.op_table:
        with open(args.op_table, "w") as f:
            f.write(f"# train: per-op HIP-event times, batch {tr.batch}, {args.size}x{args.size}, {args.dtype}; total {total:.3f} ms\n")
            for tag, what, kind, ms_, it in rows:
                shape = ""
                if what == "op":
                    I = it.i
                    shape = f"  N{I[0]} {I[1]}x{I[2]} C{I[3]}" + (f" -> {I[4]}x{I[5]} C{I[6]} k{I[7]} s{I[8]}" if kind in (hiplib.OP_CONV, hiplib.OP_CONV_WGRAD, hiplib.OP_DW_WGRAD, hiplib.OP_DWCONV) else "")
                    if kind in (hiplib.OP_CONV, hiplib.OP_CONV_WGRAD):
                        shape += f"  {conv_flops(it, kind == hiplib.OP_CONV_WGRAD) / (ms_ * 1e-3) / 1e12:7.1f} TF/s" + ("  dgrad" if kind == hiplib.OP_CONV and I[22] else "")
                    by = op_bytes(kind, it)
                    if by > 0:  # algorithmic bytes, their rate, and the launch's time beyond max(bytes at the copy rate, flops at the MFMA peak)
                        fl = conv_flops(it, kind == hiplib.OP_CONV_WGRAD) if kind in (hiplib.OP_CONV, hiplib.OP_CONV_WGRAD) else 0.0
                        floor = max(by / 5.5e12, fl / (PEAK[args.dtype] * 1e12)) * 1e3
                        shape += f"  | {by / 1e6:8.1f} MB {by / (ms_ * 1e-3) / 1e9:7.0f} GB/s  excess {ms_ - floor:7.4f} ms"
                f.write(f"{tag:5s} {names.get(kind, 'torch-attention'):22s} {ms_:9.4f} ms{shape}\n")
    return roof


def cpu_baseline_train(state, batch, seconds_budget=20.0, scale="n"):
    """Oracle train step (train-mode forward + oracle loss + autograd backward) on the host cores, bounded sample."""
    from oracle import loss as OL
    from oracle import yolo11seg as Y

    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    m = Y.build(scale, 1)
    m.load_state_dict(state)
    m.train()
    n_img, t0, steps, bs = 0, time.perf_counter(), 0, 2
    while (time.perf_counter() - t0) < seconds_budget and steps < 400:  # bounded sample: ~20 s of host work
        lo = (steps * bs) % len(batch["img"])
        x = torch.from_numpy(batch["img"][lo : lo + bs]).permute(0, 3, 1, 2).float() / 255
        keep = np.isin(batch["batch_idx"], np.arange(lo, lo + bs))
        tb = {"batch_idx": torch.from_numpy(batch["batch_idx"][keep] - lo), "cls": torch.from_numpy(batch["cls"][keep]),
              "bboxes": torch.from_numpy(batch["bboxes"][keep]), "masks": torch.from_numpy(batch["masks"][lo : lo + bs]).float()}
        feats, mc, p = m(x)
        loss, _ = OL.v8_segmentation_loss(feats, mc, p, tb, nc=1)
        m.zero_grad()
        loss.backward()
        n_img += bs
        steps += 1
    dt = time.perf_counter() - t0
    return {"value": round(n_img / dt, 3), "unit": "slices/s", "cores": cores, "kind": "port",
            "sample": f"{steps} oracle train steps of {bs} synthetic 640x640 slices (fp32 forward + loss + autograd backward, no optimizer) in {dt:.1f}s"}


def slice_extract_bench(dev, host_sample=True):
    """The step before the path: cutting, enhancing and rendering the 182 axial slices of a 182x218x182 volume on the device
    (MSL_OP_SLICE_EXTRACT, one launch) per variant, with the NumPy restatement timed on a few slices beside it."""
    from mslesseg_amd import volume as V
    from mslesseg_amd.enhance import aplicar_mejora

    rng = np.random.default_rng(0)
    vol = np.clip(rng.normal(300.0, 120.0, size=(182, 218, 182)), 0, None)
    src = V.upload_volume(vol, dev)
    idx = list(range(182))
    out = {}
    for mejora in (None, "HE", "CLAHE", "GC", "LT"):
        V.extract_slices(src, vol.shape, "axial", idx, mejora)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(10):
            V.extract_slices(src, vol.shape, "axial", idx, mejora)
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / 10
        rec = {"device_ms_per_182_slices": round(dt * 1e3, 3), "device_slices_per_s": round(182 / dt, 1)}
        if host_sample:
            t0 = time.perf_counter()
            for i in range(60, 76):
                V.slice_as_png_array(aplicar_mejora(V.take_slice(vol, "axial", i), mejora))
            rec["host_numpy_slices_per_s"] = round(16 / (time.perf_counter() - t0), 1)
        out[str(mejora)] = rec
    return out


def volume_plane_bench(eng, dev):
    """The reference's unit of work around the path: one patient volume, one plane — 182 axial slices of 218 x 182 cut from a 182x218x182
    volume, rendered, letterboxed to 640 x 544, predicted, merged and inserted into the plane volume (`volume.predict_volume`).  Timed from the
    host volume (the 58 MB upload is inside) to the finished device volume."""
    from mslesseg_amd import volume as V

    class _Model:  # predict_volume only needs the engine accessor of the YOLO wrapper
        def _get_engine(self):
            return eng

    rng = np.random.default_rng(1)
    vol = np.asfortranarray(np.clip(rng.normal(300.0, 120.0, size=(182, 218, 182)), 0, None))  # x fastest in memory, as read_nifti / get_fdata return it
    m = _Model()
    V.predict_volume(m, vol, "axial")  # plans for the two batch shapes (128 + 54 slices)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        out = V.predict_volume(m, vol, "axial")
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / reps
    res = {"ms_per_plane_volume": round(dt * 1e3, 2), "slices_per_s_incl_upload_extract_insert": round(182 / dt, 1), "slices": 182,
           "note": "host float64 volume -> device plane volume; upload 58 MB + MSL_OP_SLICE_EXTRACT + LetterBox + net + NMS + masks + merge + insert"}
    # BASELINE configs[3]: the whole patient — axial + coronal + sagittal predictions of every slice (582), reconstruct, 3-plane consensus, Dice
    # against a ground-truth volume — what generar_predicciones x 3 + reconstruir_volumen x 3 + generar_consenso + eval do per patient
    gt = torch.from_numpy((rng.random((182, 218, 182)) < 0.01).astype(np.uint8)).to(dev)
    models = {pl: m for pl in ("axial", "coronal", "sagital")}
    V.predict_consensus(models, vol, umbral=2)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(3):
        cons, _ = V.predict_consensus(models, vol, umbral=2)
        d, _ = V.dice(gt, cons)
    torch.cuda.synchronize(dev)
    dt3 = (time.perf_counter() - t0) / 3
    res["patient_three_planes_consensus_dice"] = {"ms_per_patient": round(dt3 * 1e3, 2), "slices": 582, "slices_per_s": round(582 / dt3, 1),
                                                  "note": "BASELINE configs[3]: 3 x (upload + extract + predict + insert) + consensus + Dice on the device, from the host volume"}
    return res


def batch1_latency(eng, state, host, dev, args):
    """The call an UNMODIFIED reference makes: one slice per `modelo(img, verbose=False)[0]`, then `.masks.data.cpu().numpy()`
    [REF scripts/generar_predicciones.py:114-120] — batch 1, fp32 (the predict default), host array in, host masks out.  Also the engine's batch-1
    path that stops at the merged uint8 slice (what `predict_slices` returns), eager and as a hipGraph replay of the network program."""
    y = bench_model(args, dev, state)
    y.dtype = eng.dtype
    y._engine = eng
    n = min(len(host), 32)
    for i in range(3):
        r = y(host[i], verbose=False)[0]
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    kept = 0
    for i in range(n):
        r = y(host[i], verbose=False)[0]
        if r.masks is not None:
            kept += r.masks.data.cpu().numpy().shape[0]
    call_ms = (time.perf_counter() - t0) / n * 1e3
    out = {"reference_call_ms_per_slice": round(call_ms, 3), "reference_call_slices_per_s": round(1e3 / call_ms, 1), "mean_instances": round(kept / n, 1),
           "note": "model(img, verbose=False)[0].masks.data.cpu().numpy() per slice, batch 1, fp32, 640x640 synthetic slices; engine_*: LetterBox + net + NMS + "
                   "masks + merge + D2H of the merged uint8 slice per slice"}
    for name, replay in (("engine_eager_ms_per_slice", False), ("engine_hipgraph_ms_per_slice", True)):
        for i in range(3):
            eng.predict_slices(torch.from_numpy(host[i : i + 1]), graph_replay=replay).cpu()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(n):
            eng.predict_slices(torch.from_numpy(host[i : i + 1]), graph_replay=replay).cpu()
        out[name] = round((time.perf_counter() - t0) / n * 1e3, 3)
    return out


class RepeatDataset:
    """`ds` repeated k times (optionally only the items `select(i)` keeps): a fold-sized dataset from the one demo patient's slices."""

    def __init__(self, ds, k, select=None, raw=True):
        self.ds, self.imgsz = ds, ds.imgsz
        self.map = [i for _ in range(k) for i in range(len(ds)) if select is None or select(i)]
        if raw:
            self.raw = [ds.raw[i] for i in self.map]  # raw slices: resized on the device

    def __len__(self):
        return len(self.map)

    def resized_shape(self, i):
        return self.ds.resized_shape(self.map[i])

    def get(self, i):
        return self.ds.get(self.map[i])


def demo_p39_dataset():
    from mslesseg_amd import data as D

    z = np.load(ROOT / "tests" / "golden" / "demo_volumes.npz")
    shape = tuple(int(v) for v in z["P39_shape"])
    mask = np.unpackbits(z["P39_mask_bits"])[: int(np.prod(shape))].reshape(shape).astype(np.uint8)
    return D.VolumeSliceDataset(z["P39_flair_u16"].astype(np.float64), mask)


def bench_model(args, dev, state):
    from mslesseg_amd.hiplib import MSL_BF16, MSL_F32
    from mslesseg_amd.yolo import YOLO

    y = YOLO.__new__(YOLO)  # a model object around the benchmark weights (no checkpoint file involved)
    y.ckpt_path, y.task, y.device, y.names, y._engine, y.trainer = Path("synthetic-weights"), "segment", str(dev), {0: "lesion"}, None, None
    y.dtype = y.train_dtype = MSL_BF16 if args.dtype == "bf16" else MSL_F32
    y.scale, y.nc, y.state, y.pretrained = args.scale, 1, state, True
    return y


def fit_epoch_bench(args, dev, rank, world, state, dist):
    """`model.train()`'s whole epoch — train steps from the device feeder, validation of the held-out fold, checkpoint — on a fold-sized dataset
    (demo patient P39's 361 lesion slices x 8: 2 312 train / 576 held out, like the reference's folds of 112-174 iterations x batch), per rank:
    seconds of train steps / validation / checkpoint hand-off.  What `--mode train` cannot see: work that only one rank does."""
    from mslesseg_amd.train import Trainer

    base = demo_p39_dataset()
    tr_ds = RepeatDataset(base, 8, select=lambda i: i % 5 != 0)
    va_ds = RepeatDataset(base, 8, select=lambda i: i % 5 == 0)
    B = args.batch or 256  # what batch=-1 resolves to
    epochs = max(args.steps, 2) + 1
    tr = Trainer(bench_model(args, dev, state), dataset=tr_ds, val_dataset=va_ds, epochs=epochs, batch=B, project=ROOT / "gpurun_out" / "bench_runs", name=f"fit{rank}_of{world}",
                 imgsz=args.size, close_mosaic=0)
    t0 = time.perf_counter()
    tr.fit()
    wall = time.perf_counter() - t0
    et = tr.epoch_times[1:]  # the first epoch builds the validation plans
    mine = [float(np.median([e[k] for e in et])) for k in ("train_s", "val_s", "ckpt_s")]
    allr = [mine]
    if world > 1:
        t = torch.tensor(mine, dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        g = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(g, t)
        allr = [[float(v) for v in x] for x in g]
    epoch_s = max(sum(r) for r in allr)
    return {"train_slices": len(tr_ds), "val_slices": len(va_ds), "per_gpu_batch": B, "iterations_per_epoch": tr.nb, "epochs_timed": len(et), "fit_wall_s": round(wall, 2),
            "epoch_s": round(epoch_s, 4), "per_rank_train_val_ckpt_s": [[round(v, 4) for v in r] for r in allr],
            "rank0_only_s": round(allr[0][2], 4), "rank0_only_share_of_epoch": round(allr[0][2] / epoch_s, 4),
            "note": "medians over the timed epochs; epoch_s = wall time per epoch (max over ranks).  Nothing synchronises with the host inside an epoch any more, so the "
                    "per-phase seconds are host ENQUEUE times (the train phase returns early, the validation phase then waits on the queue): read epoch_s.  "
                    "Validation is sharded over the ranks (train._validate); on one rank its host half (AP curves) and the checkpoints run in a writer "
                    "thread behind the next epoch's steps"}, tr


def train_e2e_bench(args, dev, rank, world, state, B):
    """`model.train()`'s real loop: batches come from the data feeder (mosaic on, the reference's augmentation set) instead of one resident batch.
    Dataset: the lesion slices of demo patient P39 rendered as the reference's dataset stage writes them (361 slices, three planes), repeated 8x
    so that an epoch has ~22 iterations at batch 128 like the reference's folds (112-174 iterations per epoch)."""
    from mslesseg_amd import data as D
    from mslesseg_amd import hiplib
    from mslesseg_amd.hiplib import MSL_BF16, MSL_F32
    from mslesseg_amd.train import Trainer
    from mslesseg_amd.yolo import YOLO

    t0 = time.perf_counter()
    base = demo_p39_dataset()
    t_ds = time.perf_counter() - t0

    ds = RepeatDataset(base, 8, raw=not args.host_augment)
    y = bench_model(args, dev, state)
    tr = Trainer(y, dataset=ds, val_dataset=None, epochs=10 ** 6, batch=B, project=ROOT / "gpurun_out" / "bench_runs", name=f"e2e{rank}", imgsz=args.size,
                 nbs=B * world, warmup_epochs=0.0, device_augment=not args.host_augment)

    def stream():
        e = 0
        while True:
            yield from tr._batches(e)
            e += 1

    it = stream()
    lr = tr.lr0

    def step():
        tr.forward_backward(next(it))
        tr.optimizer_step(lr)

    split = {"dataset_build_s": round(t_ds, 1), "slices": len(ds), "iterations_per_epoch": tr.nb}
    if tr.aug is not None:  # the two halves of the feeder on their own: host (random draws + label geometry, one thread) and device (two launches)
        rng = np.random.default_rng(0)
        idx = list(range(B))
        tr.aug.prepare(idx, rng, True, True)
        t0 = time.perf_counter()
        for _ in range(5):
            h = tr.aug.prepare(idx, rng, True, True)
        split["host_prepare_ms_per_batch"] = round((time.perf_counter() - t0) / 5 * 1e3, 2)
        st_ = torch.cuda.current_stream(dev).cuda_stream
        tr.aug.render(h)
        e0, e1 = hiplib.Event(), hiplib.Event()
        e0.record(st_)
        for _ in range(5):
            tr.aug.render(h)
        e1.record(st_)
        torch.cuda.synchronize(dev)
        split["device_render_ms_per_batch"] = round(e0.elapsed_ms(e1) / 5, 3)
    return tr, step, split


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", default="train", choices=["train", "predict", "train-e2e", "replicas", "fit-epoch"],
                    help="train: one resident batch per step (the headline); predict: the inference leg; train-e2e: the same step fed by the data feeder "
                         "(mosaic on); replicas: independent trainings, one per GPU, no collective (SURVEY 8e zero-communication mode); fit-epoch: whole epochs of "
                         "model.train() on a fold-sized dataset (train steps + sharded validation + checkpoint), --steps = epochs timed")
    ap.add_argument("--batch", type=int, default=0, help="slices per GPU per step (default: 128 for both legs — north_star: batch >= 128)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "fp32s"], help="fp32s (predict only): fp32 tensors, split-precision conv products (MSL_F32S)")
    ap.add_argument("--scale", default="n", choices=["n", "s"], help="n = BASELINE configs[1]; s = configs[2] (YOLO11s-seg, seeded random initialisation)")
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-augment", action="store_true", help="train-e2e: the NumPy augmentation path of data.py instead of the device feeder")
    ap.add_argument("--replay-dominant", type=int, default=0, help="after the roofline pass, launch the dominant op this many more times (PMC collection)")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-op replay (for a clean rocprofv3 trace of the timed steps only)")
    ap.add_argument("--no-infer", action="store_true", help="train mode: skip the short predict run reported under 'infer'")
    ap.add_argument("--op-table", default="", help="write the per-op timing table to this file")
    ap.add_argument("--target-kept", type=float, default=12.0,
                    help="predict leg: shift the class bias so NMS keeps about this many instances per synthetic slice (0: leave the "
                         "calibrated-random weights as they are, which saturates max_det=300 on noise slices)")
    args = ap.parse_args()
    if args.dtype == "fp32s" and args.mode != "predict":
        ap.error("--dtype fp32s is a predict mode (train steps run bf16 or fp32)")
    world, rank, dev, dist = dist_setup(args)
    state = load_weights(args.scale)
    S = args.size
    model_name = f"YOLO11{args.scale}-seg"
    cfg_name = "BASELINE configs[1]" if args.scale == "n" else "BASELINE configs[2] model"
    wdesc = "calibrated random weights" if args.scale == "n" else "seeded random initialisation"
    line = {"metric": METRIC, "unit": "slices/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic"}

    if args.mode in ("train", "replicas"):
        B = args.batch or 128
        tr, dbatch, batch = train_setup(args, dev, rank, world, state, B)
        lr = tr.lr0

        dp = args.mode == "train"

        def step():
            tr.forward_backward(dbatch, reduce_now=dp)  # data parallel: the gradient buckets are all-reduced as the backward programs complete them
            tr.optimizer_step(lr)

        dt = timed(step, args, dev, world, dist)
        value = world * B * args.steps / dt
        if world > 1 and tr.collective is not None:
            # every rank's start-up self-check of the gradient collective on the real backend (train.collective_selfcheck: a known sum over RCCL / xGMI)
            recs = [None] * world
            dist.all_gather_object(recs, tr.collective)
            line["rccl_ranks"] = recs
        line.update(value=round(value, 2), ms_per_step=round(dt / args.steps * 1e3, 4),
                    config={"workload": f"train step ({cfg_name}): {model_name} nc=1, {S}x{S}x3 uint8 slices, batch {B}/GPU, {args.dtype} activations+weights / "
                                        f"fp32 master+accumulate; pack + forward(train BN) + loss + backward + {'all-reduce + ' if (world > 1 and dp) else ''}clip + AdamW + EMA; "
                                        f"{wdesc}, 1-6 random polygons per slice",
                            "per_gpu_batch": B, "global_batch": B * world,
                            "parallelism": (f"dp{world}: slices sharded, flat-gradient all-reduce per step in two buckets (head + neck beside the backbone's backward)" if dp else
                                            f"replicas x{world}: independent trainings (the reference's fold x plane jobs), one per GPU, no collective")})
        if rank == 0:
            line["roofline"] = None if args.no_roofline else train_roofline(tr, dbatch, args, value / world)
            line["cpu_baseline"] = None if (args.no_cpu_baseline or world > 1) else cpu_baseline_train(state, batch, scale=args.scale)
            if not args.no_infer and world == 1:
                del tr, dbatch
                torch.cuda.empty_cache()
                if B == 128 and args.mode == "train":
                    # the same step at the batch the reference's call `model.train(batch=-1)` resolves to in this library (train.py: 256 slices per GPU)
                    tr2, db2, _ = train_setup(args, dev, rank, world, state, 256)

                    def step2():
                        tr2.forward_backward(db2)
                        tr2.optimizer_step(tr2.lr0)

                    for _ in range(3):
                        step2()
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    for _ in range(8):
                        step2()
                    torch.cuda.synchronize(dev)
                    d2 = (time.perf_counter() - t0) / 8
                    line["batch_256"] = {"value": round(256 / d2, 2), "unit": "slices/s (1 GPU)", "ms_per_step": round(d2 * 1e3, 3), "per_gpu_batch": 256,
                                         "note": "what batch=-1 resolves to; the headline `value` stays at batch 128 (north_star: batch >= 128) for continuity with round 1"}
                    del tr2, db2
                    torch.cuda.empty_cache()
                line["infer"] = {}
                # the inference leg in both arithmetic modes: fp32 = the default of YOLO() predict (exact parity with the CPU path: identical NMS
                # indices and mask bytes, tests/test_gpu_trained.py), bf16 = the opt-in throughput mode
                for pdt in ("fp32", "fp32s", "bf16"):
                    pargs = argparse.Namespace(**{**vars(args), "steps": 10, "warmup": 3, "dtype": pdt})
                    eng, imgs, host, pstate, shift = predict_setup(pargs, dev, rank, state, 128)
                    t0 = None
                    for i in range(13):
                        if i == 3:
                            torch.cuda.synchronize(dev)
                            t0 = time.perf_counter()
                        eng.predict_slices(imgs)
                    torch.cuda.synchronize(dev)
                    pdt_s = time.perf_counter() - t0
                    line["infer"][pdt] = {"value": round(128 * 10 / pdt_s, 2), "unit": "slices/s (1 GPU)", "ms_per_step": round(pdt_s / 10 * 1e3, 3), "per_gpu_batch": 128,
                                          "workload": f"predict leg: LetterBox+net+NMS+masks+merge, class bias shifted {shift:+.2f} for ~{args.target_kept:g} kept instances/slice",
                                          "mean_kept_instances_per_slice": round(float(eng.plan(128, S, S).keep_cnt.float().mean().item()), 1),
                                          "parity": {"fp32": "exact vs the CPU oracle (north_star tolerance met)",
                                                     "fp32s": "fp32 tensors, conv products as three f16 partial products (MSL_F32S): north_star tolerance met on the trained "
                                                              "checkpoint (tests/test_gpu_trained.py, same assertions as fp32)",
                                                     "bf16": "throughput mode: |dDice| <= 1e-3 per plane volume, not at the 1e-4 tolerance"}[pdt]}
                    if pdt == "fp32":
                        line["infer"]["batch1"] = batch1_latency(eng, pstate, host, dev, args)
                    del eng, imgs
                    torch.cuda.empty_cache()
    elif args.mode == "fit-epoch":
        rec, tr = fit_epoch_bench(args, dev, rank, world, state, dist)
        value = rec["train_slices"] / rec["epoch_s"]
        line.update(value=round(value, 2), ms_per_step=round(rec["epoch_s"] / max(tr.nb, 1) * 1e3, 4), steps=rec["epochs_timed"], warmup=1, scaling="strong",
                    data="real FLAIR slices (demo patient P39, 361 lesion slices x 8, every 5th held out) through the training augmentation",
                    config={"workload": f"whole epochs of model.train() ({cfg_name}): {model_name} nc=1, batch {rec['per_gpu_batch']}/GPU, {args.dtype}; device feeder (mosaic on) + "
                                        f"train steps + eval-mode validation of the held-out fifth (sharded over ranks) + last.pt/best.pt; value = train slices per epoch second",
                            "per_gpu_batch": rec["per_gpu_batch"], "global_batch": rec["per_gpu_batch"] * world, "parallelism": f"dp{world}"})
        if rank == 0:
            line["fit_epoch"] = rec
    elif args.mode == "train-e2e":
        B = args.batch or 128
        tr, step, split = train_e2e_bench(args, dev, rank, world, state, B)
        dt = timed(step, args, dev, world, dist)
        value = world * B * args.steps / dt
        line.update(value=round(value, 2), ms_per_step=round(dt / args.steps * 1e3, 4), data="real FLAIR slices (demo patient P39, 361 lesion slices x 8) through the training augmentation",
                    config={"workload": f"end-to-end train step ({cfg_name}): {model_name} nc=1, batch {B}/GPU, {args.dtype}; every batch drawn fresh from the data feeder "
                                        f"(mosaic 1.0, scale 0.5, translate 0.1, hsv_v 0.4, fliplr 0.5; {'NumPy path (data.py)' if args.host_augment else 'device feeder (augment.py: 2 HIP launches per batch)'}) "
                                        f"+ pack + forward + loss + backward + clip + AdamW + EMA",
                            "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}"})
        if rank == 0:
            # the resident-batch number beside it (what the headline `train` mode measures): same trainer, one batch replayed
            it_batch = tr.aug.batch(list(range(B)), np.random.default_rng(1), True, True) if tr.aug is not None else None
            if it_batch is not None:
                def rstep():
                    tr.forward_backward(it_batch)
                    tr.optimizer_step(tr.lr0)

                rdt = timed(rstep, args, dev, 1, None)
                split["resident_batch_slices_per_s"] = round(B * args.steps / rdt, 1)
                split["e2e_over_resident"] = round(value / world / split["resident_batch_slices_per_s"], 4)
            line["feeder"] = split
    else:
        B = args.batch or 128
        eng, imgs, host, pstate, shift = predict_setup(args, dev, rank, state, B)
        out = None

        def step():
            nonlocal out
            out = eng.predict_slices(imgs)

        dt = timed(step, args, dev, world, dist)
        value = world * B * args.steps / dt
        line.update(value=round(value, 2), ms_per_step=round(dt / args.steps * 1e3, 4),
                    config={"workload": f"predict (infer leg of the metric): {model_name} nc=1, {S}x{S}x3 uint8 slices, LetterBox+net+NMS+masks+merge, batch {B}/GPU; "
                                        f"{wdesc}, class bias shifted {shift:+.2f} for ~{args.target_kept:g} kept instances/slice",
                            "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"slice-sharded x{world}, no collective"})
        if rank == 0:
            line["roofline"] = None if args.no_roofline else predict_roofline(eng, imgs, out, args, value / world)
            line["cpu_baseline"] = None if (args.no_cpu_baseline or world > 1) else cpu_baseline_predict(pstate, host[: min(B, 64)], scale=args.scale)
            if not args.no_roofline and args.target_kept > 0:
                # the same leg with the class bias left alone: every noise slice then keeps max_det = 300 instances (NMS and mask assembly saturated) —
                # the tuned ~12-instance workload above is the realistic one (MS slices carry a handful of lesions), this is its worst case
                sargs = argparse.Namespace(**{**vars(args), "target_kept": 0.0})
                seng, simgs, _, _, _ = predict_setup(sargs, dev, rank, state, B)
                for i in range(8):
                    if i == 3:
                        torch.cuda.synchronize(dev)
                        t0 = time.perf_counter()
                    seng.predict_slices(simgs)
                torch.cuda.synchronize(dev)
                sdt = (time.perf_counter() - t0) / 5
                line["saturated_nms"] = {"value": round(B / sdt, 2), "unit": "slices/s (1 GPU)", "ms_per_step": round(sdt * 1e3, 3),
                                         "mean_kept_instances_per_slice": round(float(seng.plan(B, S, S).keep_cnt.float().mean().item()), 1)}
                del seng, simgs
                torch.cuda.empty_cache()
            line["slice_extract"] = None if args.no_roofline else slice_extract_bench(dev, host_sample=not args.no_cpu_baseline)
            line["volume_plane"] = None if args.no_roofline else volume_plane_bench(eng, dev)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
