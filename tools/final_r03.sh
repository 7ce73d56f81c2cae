#!/bin/bash
# round-3 closing run on the GPU box: the whole -m gpu suite, the per-config profiles, the bench line + its trace
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -vE "^Read mtx" | grep -E "passed|failed|Error|error|FAILED" | tail -6
bash tools/profile_configs.sh r03 webbase1m stencil rmat18 rmat22 cant cant_irr 2>&1 | grep -E "^== " 
bash tools/gpu_bench_profile.sh r03b 2>&1 | tail -3
NSPARSE_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --spmv-steps 5 --no-cpu --no-pmc --no-vendor 2> gpurun_out/r03b/b2.err | tail -1 | cut -c1-300
tail -2 gpurun_out/r03b/b2.err
