#!/bin/bash
# Roofline evidence of the CURRENT tree; run ON the GPU box from the repo root:  bash scripts/profile_head.sh <tag>   (tag e.g. r03a)
#   1. rocprofv3 --kernel-trace --stats over whole train steps (no per-op replay, no predict leg) -> gpurun_out/<tag>_kernel_stats_train_bf16_b128.csv
#      and the per-dispatch trace the in-stream duration of the dominant launch is read from (scripts/in_stream.py)
#   2. the three --pmc passes of scripts/pmc_traffic.py (FETCH_SIZE, WRITE_SIZE, SQ) -> profiles/pmc_latest.json + gpurun_out/pmc_latest.json
#   3. the default bench line with the op table -> gpurun_out/<tag>_bench.json, gpurun_out/<tag>_op_table_train.txt
# Each rocprofv3 call has the program itself after `--` (python3 bench.py ...), never a shell or env hop.
set -e
TAG=${1:-r03a}
R=$(pwd)
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/${TAG}_prof" -o train -- python3 "$R/bench.py" --steps 10 --warmup 3 --no-roofline --no-infer --no-cpu-baseline > "$R/gpurun_out/${TAG}_prof_bench.json" 2> "$R/gpurun_out/${TAG}_prof.err"
cd "$R"
S=$(find gpurun_out/${TAG}_prof -name '*kernel_stats.csv' | head -1)
T=$(find gpurun_out/${TAG}_prof -name '*kernel_trace.csv' | head -1)
cp "$S" gpurun_out/${TAG}_kernel_stats_train_bf16_b128.csv
echo "stats: $S trace: $T"
python3 scripts/pmc_traffic.py > gpurun_out/${TAG}_pmc.log 2>&1
python3 scripts/in_stream.py "$T" gpurun_out/pmc_latest.json > gpurun_out/${TAG}_in_stream.json
cp gpurun_out/${TAG}_in_stream.json gpurun_out/in_stream_latest.json
python3 bench.py --op-table gpurun_out/${TAG}_op_table_train.txt > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -c 1500 gpurun_out/${TAG}_bench.json
# the kernel trace is large: keep only the summaries
rm -rf gpurun_out/${TAG}_prof
