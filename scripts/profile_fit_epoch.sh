#!/bin/bash
# Run ON the GPU box: rocprofv3 kernel stats of bench.py --mode fit-epoch (train steps + sharded validation + checkpoints).
set -e
R=$(pwd); mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/fit_prof" -o fit -- python3 "$R/bench.py" --mode fit-epoch --no-cpu-baseline > "$R/gpurun_out/fit_prof_bench.json" 2> "$R/gpurun_out/fit_prof.err"
cd "$R"
S=$(find gpurun_out/fit_prof -name '*kernel_stats.csv' | head -1)
cp "$S" gpurun_out/fit_epoch_kernel_stats.csv
rm -rf gpurun_out/fit_prof
head -40 gpurun_out/fit_epoch_kernel_stats.csv | cut -c1-160
