#!/bin/bash
# Run ON the GPU box: kernel trace of a few train steps, then the idle-gap analysis.
set -e
R=$(pwd); mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/gaps_prof" -o train -- python3 "$R/bench.py" --steps 8 --warmup 3 --no-roofline --no-infer --no-cpu-baseline > "$R/gpurun_out/gaps_bench.json" 2> "$R/gpurun_out/gaps.err"
cd "$R"
T=$(find gpurun_out/gaps_prof -name '*kernel_trace.csv' | head -1)
python3 scripts/dev_trace_gaps.py "$T" > gpurun_out/trace_gaps.txt
rm -rf gpurun_out/gaps_prof
cat gpurun_out/trace_gaps.txt
