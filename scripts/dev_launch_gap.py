"""Dev: what a launch costs on this box — N back-to-back launches of a 256-element kernel (MSL_OP_EMA) on one stream, eager and as a hipGraph."""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "yolo-mslesseg_amd")]
from mslesseg_amd import hiplib
dev = "cuda:0"; st = torch.cuda.current_stream().cuda_stream
t = torch.zeros(512, device=dev)
op = hiplib.make_op(hiplib.OP_EMA, hiplib.MSL_F32, p=(t.data_ptr(), t[256:].data_ptr()), i={0: 256}, f=(0.5,))
for n in (100, 1000, 5000):
    for _ in range(50): hiplib.launch(op, st)
    torch.cuda.synchronize(); e0, e1 = hiplib.Event(), hiplib.Event()
    t0 = time.perf_counter(); e0.record(st)
    for _ in range(n): hiplib.launch(op, st)
    e1.record(st); t_host = time.perf_counter() - t0; torch.cuda.synchronize()
    print(f"{n} tiny launches: device {e0.elapsed_ms(e1) / n * 1e3:.2f} us per launch, host enqueue {t_host / n * 1e6:.2f} us per launch", flush=True)
prog = hiplib.Program([op] * 1000)
prog.run(st); torch.cuda.synchronize()
e0, e1 = hiplib.Event(), hiplib.Event(); e0.record(st); prog.run(st); e1.record(st); torch.cuda.synchronize()
print(f"1000 tiny launches as one msl_run_program call: {e0.elapsed_ms(e1):.3f} ms")
prog.replay(st); torch.cuda.synchronize()
e0, e1 = hiplib.Event(), hiplib.Event(); e0.record(st); prog.replay(st); e1.record(st); torch.cuda.synchronize()
print(f"1000 tiny launches as a hipGraph replay: {e0.elapsed_ms(e1):.3f} ms")
