#!/usr/bin/env python3
"""bench.py — FLAIR slices/sec of the YOLO11n-seg hot path on MI355X (contract: see the task brief / DESIGN.md §Measurement).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one pass of the hot path over one batch of synthetic slices already resident in HBM:
  predict (default, round 1): uint8 [B,640,640,3] → LetterBox → YOLO11n-seg (nc=1) → decode → NMS → mask assembly →
  merged + re-oriented uint8 [B,640,640] (everything the reference does per slice between cv2.imread and cv2.imwrite
  [REF yolo_mslesseg/scripts/generar_predicciones.py:205-222], for the whole batch).
Slices shard over ranks with no data-path collective (independent units) → "scaling": "weak".
Rank 0 prints ONE JSON line.  The oracle is imported only for the cpu_baseline leg.
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
PEAK_MFMA_BF16_TFLOPS = 2500.0  # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md §Chip-level parameters
PEAK_MFMA_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0
FWD_GFLOP_PER_SLICE_640 = 9.630  # n, nc=1, 640x640, 2*MAC over conv + attention + ConvT


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


def load_weights():
    st = torch.load(ROOT / "tests" / "golden" / "synth_n_nc1.pt", map_location="cpu", weights_only=True)
    return {k: (v.float() if v.is_floating_point() else v) for k, v in st.items()}


def cpu_baseline(state, imgs_u8, seconds_budget=20.0):
    """Restated CPU path (the reference's ultralytics-on-CPU cannot run here: SURVEY §8d), batch 1 per slice exactly like
    the reference's loop [REF generar_predicciones.py:205-222], on this host's cores."""
    from oracle import prepost as P
    from oracle import synth

    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    om = synth.model_from_state(state)
    P.generar_prediccion_2D(om, imgs_u8[0])  # warm-up
    t0, n = time.perf_counter(), 0
    while n < len(imgs_u8) and (time.perf_counter() - t0) < seconds_budget:
        P.generar_prediccion_2D(om, imgs_u8[n])
        n += 1
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 3), "unit": "slices/s", "cores": cores, "kind": "port",
            "sample": f"{n} synthetic 640x640 slices, batch 1, fp32, oracle restatement of the reference predict loop "
                      f"(letterbox+net+NMS+masks+merge) in {dt:.1f}s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=128, help="slices per GPU per step")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--op-table", default="", help="write the per-op timing table to this file")
    ap.add_argument("--target-kept", type=float, default=12.0,
                    help="shift the class bias so NMS keeps about this many instances per synthetic slice (0: leave the "
                         "calibrated-random weights as they are, which saturates max_det=300 on noise slices)")
    args = ap.parse_args()

    from mslesseg_amd import engine as E
    from mslesseg_amd import hiplib
    from mslesseg_amd.hiplib import MSL_BF16, MSL_F32

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)

    dtype = MSL_BF16 if args.dtype == "bf16" else MSL_F32
    state = load_weights()
    B, S = args.batch, args.size
    host = synthetic_slices(B, S, S, seed=rank)  # each rank its own shard of slices
    imgs = torch.from_numpy(host).to(dev)  # inputs resident in HBM before the timed region
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
            probe = E.InferEngine(st, "n", 1, dtype, str(dev))
            kept = float(probe.predict_batch(imgs[:16]).keep_cnt.float().mean().item())
            del probe
            lo, hi = (mid, hi) if kept < args.target_kept else (lo, mid)
        bias_shift = 0.5 * (lo + hi)
        for i in range(3):
            state[f"model.23.cv3.{i}.2.bias"] = state[f"model.23.cv3.{i}.2.bias"] + bias_shift
        torch.cuda.empty_cache()
    eng = E.InferEngine(state, "n", 1, dtype, str(dev))
    out = None

    def step():
        nonlocal out
        out = eng.predict_slices(imgs)

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
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * B * args.steps / dt

    if rank == 0:
        lb_plan = eng.plan(B, S, S)
        # ---- roofline of the dominant kernel, from HIP events on the launch stream
        table = lb_plan.time_ops(reps=3)
        costs = [lb_plan.op_cost(i) for i in range(len(table))]
        total_ms = sum(t[2] for t in table)
        conv_ms = sum(t[2] for t in table if t[1] == hiplib.OP_CONV)
        conv_flops = sum(c[0] for t, c in zip(table, costs) if t[1] == hiplib.OP_CONV)
        dom = max((i for i in range(len(table)) if table[i][1] == hiplib.OP_CONV), key=lambda i: table[i][2])  # dominant MFMA kernel launch
        d_name, d_kind, d_ms = table[dom]
        d_flops, d_bytes = costs[dom]
        peak = PEAK_MFMA_BF16_TFLOPS if dtype == MSL_BF16 else PEAK_MFMA_F32_TFLOPS
        achieved = d_flops / (d_ms * 1e-3) / 1e12 if d_ms > 0 else 0.0
        pmc = None
        pmc_file = ROOT / "profiles" / "pmc_latest.json"
        if pmc_file.exists():
            try:
                pmc = json.loads(pmc_file.read_text()).get(d_name)
            except Exception:
                pmc = None
        roofline = {
            "bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
            "traffic": pmc,
            "kernel": f"conv_igemm_kernel<{args.dtype}> @ {d_name}",
            "launch_ms": round(d_ms, 4), "algorithmic_gflop_per_launch": round(d_flops / 1e9, 3),
            "algorithmic_hbm_gbs": round(d_bytes / (d_ms * 1e-3) / 1e9, 1), "hbm_frac": round(d_bytes / (d_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
            "all_conv": {"tflops": round(conv_flops / (conv_ms * 1e-3) / 1e12, 2), "frac": round(conv_flops / (conv_ms * 1e-3) / 1e12 / peak, 4),
                         "ms": round(conv_ms, 3), "share_of_step": round(conv_ms / total_ms, 3)},
            "whole_net_frac_of_mfma_roof": round(value / world * FWD_GFLOP_PER_SLICE_640 * (S * S / 640.0 / 640.0) / 1e3 / peak, 4),
        }
        # letterbox + merge are launched outside the op program: time them the same way
        ev = [hiplib.Event() for _ in range(3)]
        lbp, _ = eng.letterbox(B, S, S, 3)
        st_ = torch.cuda.current_stream(dev).cuda_stream
        ev[0].record(st_); lbp.run(); ev[1].record(st_); lb_plan.merged(S, S, out=out); ev[2].record(st_)
        torch.cuda.synchronize(dev)
        extra = {"letterbox_ms": round(ev[0].elapsed_ms(ev[1]), 4), "mask_merge_ms": round(ev[1].elapsed_ms(ev[2]), 4),
                 "program_ms": round(total_ms, 3)}
        extra["mean_kept_instances_per_slice"] = round(float(lb_plan.keep_cnt.float().mean().item()), 1)
        extra["mask_on_fraction"] = round(float((out > 0).float().mean().item()), 3)
        roofline["step_breakdown"] = extra
        if args.op_table:
            with open(args.op_table, "w") as f:
                f.write(f"# per-op HIP-event times, batch {B}, {S}x{S}, {args.dtype}; total {total_ms:.3f} ms\n")
                for (name, kind, ms), (fl, by) in zip(table, costs):
                    f.write(f"{name:34s} kind={kind:2d} {ms:9.4f} ms  {fl / 1e9:9.3f} GFLOP  {fl / (ms * 1e-3) / 1e12 if ms > 0 else 0:8.2f} TF/s  "
                            f"{by / 1e6:9.2f} MB  {by / (ms * 1e-3) / 1e9 if ms > 0 else 0:8.1f} GB/s\n")
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(state, host[: min(B, 64)])
        line = {
            "metric": METRIC, "value": round(value, 2), "unit": "slices/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"predict (infer leg of the metric): YOLO11n-seg nc=1, {S}x{S}x3 uint8 slices, LetterBox+net+NMS+masks+merge, "
                                   f"batch {B}/GPU; calibrated random weights (no trained weights exist offline), class bias shifted {bias_shift:+.2f} "
                                   f"for ~{args.target_kept:g} kept instances/slice; training leg not built yet",
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"slice-sharded x{world}, no collective"},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
