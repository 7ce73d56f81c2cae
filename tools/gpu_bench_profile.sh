#!/bin/bash
# Run on the GPU box (through gpurun): bench line + rocprofv3 kernel trace / stats of the same command.
# Outputs go to gpurun_out/<tag>/; tools/summarize_profile.py condenses them into profiles/<tag>_*.
# (HBM traffic: bench.py makes its own --pmc passes -- roofline.traffic -- so none are made here.)
set -u
TAG=${1:-r02}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python bench.py --steps 10 --warmup 1 > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$?" >> "$OUT/bench.err"
# kernel trace + stats of the same command (CPU baseline, vendor and PMC legs off: same kernels).  The
# irregular stand-in runs the KEYED instantiation of k_num_block, so the headline kernel keeps its own row.
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python bench.py --steps 10 --warmup 1 --no-cpu --no-pmc --no-vendor > "$OUT/trace_bench.json" 2> "$OUT/trace.err"
echo "trace rc=$?" >> "$OUT/trace.err"
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
find "$OUT" -name "*.db" -delete
du -sh "$OUT"
