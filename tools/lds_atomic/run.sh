#!/bin/bash
# Builds and runs the LDS-atomic micro-benchmark on the GPU box; result -> gpurun_out/lds_atomic.json
# (copy the summary you want judged to profiles/r02_lds_atomic.json).
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics lds_atomic_bench.hip -o lds_atomic_bench
mkdir -p ../../gpurun_out
./lds_atomic_bench > ../../gpurun_out/lds_atomic.json
tail -c 600 ../../gpurun_out/lds_atomic.json
