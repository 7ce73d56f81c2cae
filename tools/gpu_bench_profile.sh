#!/bin/bash
# Run on the GPU box (through gpurun): bench line + rocprofv3 kernel stats + PMC passes.
# Outputs go to gpurun_out/<tag>/; copy what should be judged into profiles/.
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$PWD"
timeout 900 python bench.py --steps 10 --warmup 1 > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$?" >> "$OUT/bench.err"
# kernel trace + stats of the same command (CPU baseline and the big SpMV skipped: same kernels)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python bench.py --steps 10 --warmup 1 --no-cpu > "$OUT/trace_bench.json" 2> "$OUT/trace.err"
echo "trace rc=$?" >> "$OUT/trace.err"
# HBM traffic, separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass)
NSPARSE_PROFILE_SERIAL=1 timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o fetch -- python bench.py --steps 3 --warmup 1 --no-cpu --spmv-steps 5 > /dev/null 2> "$OUT/pmc_fetch.err"
NSPARSE_PROFILE_SERIAL=1 timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o write -- python bench.py --steps 3 --warmup 1 --no-cpu --spmv-steps 5 > /dev/null 2> "$OUT/pmc_write.err"
find "$OUT" -name "*.csv" | head -30
# keep only small summaries (gpurun_out is capped at 64 MiB)
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
find "$OUT" -name "*.db" -delete
du -sh "$OUT"
