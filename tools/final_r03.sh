#!/bin/bash
# round-3 closing run on the GPU box: the whole -m gpu suite, the per-config profiles, the bench line + its trace
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -vE "^Read mtx" | grep -E "passed|failed|Error|error|FAILED" | tail -6
if [ "${1:-}" = "profiles" ]; then
  bash tools/profile_configs.sh r03 webbase1m stencil rmat18 rmat22 cant cant_irr 2>&1 | grep -E "^== "
fi
bash tools/gpu_bench_profile.sh r03b 2>&1 | tail -3
NSPARSE_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --spmv-steps 5 --no-cpu --no-pmc --no-vendor 2> gpurun_out/r03b/b2.err | tail -1 | cut -c1-300
tail -2 gpurun_out/r03b/b2.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03b/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["timing"]["reference_compatible_ms"], d["timing"]["alloc_async_ms"], d["roofline"]["frac"])
for k in ("spmv", "spmv_hbm"):
    s = d[k]; print(k, {q: s.get(q) for q in ("ms_per_spmv", "value", "frac_hbm_peak", "host_us_per_spmv", "driver")})
PY
