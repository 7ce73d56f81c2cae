#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/exp9
timeout 900 python bench.py --steps 10 --warmup 1 > gpurun_out/exp9/bench.json 2> gpurun_out/exp9/bench.err
echo "bench rc=$?"; tail -5 gpurun_out/exp9/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/exp9/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "n_gpus")}, d["config"]["workload"][:60])
print("timing", d["timing"]["reference_compatible_ms"], d["timing"]["alloc_async_ms"])
print("roofline", d["roofline"]["kernel"][:80], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["ms_per_launch"])
print("regular", {k: d["regular_brick"][k] for k in ("value", "ms_per_step", "reference_compatible_ms", "twin_rows")})
print("sweep", d["structure_sweep"]["points"], d["structure_sweep"].get("twins_off"))
for k in ("spmv", "spmv_hbm"):
    s = d[k]; print(k, {q: s.get(q) for q in ("ms_per_spmv", "value", "frac_hbm_peak", "host_us_per_spmv", "hipgraph", "driver", "ans_check_fails", "traffic")})
print("cpu", d["cpu_baseline"]["value"], "vendor", d["vendor_baseline"])
PY
echo "=== 2 ranks on one GPU, gloo smoke"
NSPARSE_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --spmv-steps 5 --no-cpu --no-pmc --no-vendor 2> gpurun_out/exp9/b2.err | tail -1 | cut -c1-400
tail -3 gpurun_out/exp9/b2.err
