#!/bin/bash
# Round-4 GPU run, ONE gpurun call (stages can be picked: bash tools/final_r04.sh [tests] [ab] [spmv] [bench] [profiles]):
#   tests     the whole -m gpu suite (new files first, so that a failure in them shows early)
#   ab        tools/ab_lean3.sh: round-3 hash kernels vs the four lean builds, serialised per-bin times
#   spmv      cache-resident SpMV: split-row kernel on / off against rocSPARSE csrmv
#   bench     the bench line (with its configs block) + kernel trace of the same command -> gpurun_out/r04/
#   profiles  per-config kernel stats + PMC (tools/profile_configs.sh) with the library's default kernels
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
stages=${@:-tests ab spmv bench}
for s in $stages; do
case $s in
tests)
  timeout 1500 python -m pytest tests/test_bench_gpu.py tests/test_aux_gpu.py tests/test_partition_gpu.py tests/test_dist_native_gpu.py -m gpu -x -q 2>&1 \
    | grep -vE "^Read mtx|^RCCL|^HIP version|^ROCm version|^Hostname|^Librccl" | tail -25 > gpurun_out/r04/tests_new.log
  tail -25 gpurun_out/r04/tests_new.log
  timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -vE "^Read mtx|^RCCL|^HIP version|^ROCm version|^Hostname|^Librccl" | tail -30 > gpurun_out/r04/tests_all.log
  tail -12 gpurun_out/r04/tests_all.log ;;
ab)
  bash tools/ab_lean3.sh stencil webbase1m rmat18 rmat22 2>&1 | tail -24 | tee gpurun_out/r04/ab_lean3.txt ;;
spmv)
  for sp in 0 1 2 4 8; do
    echo "== NSPARSE_SPMV_SPLIT=$sp"
    NSPARSE_SPMV_SPLIT=$sp timeout 600 python bench.py --no-cpu --no-pmc --no-irregular --no-configs --no-large --steps 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['spmv']
print({k:s.get(k) for k in ('ms_per_spmv','ms_kernel_events','value','plan','ans_check_fails')}, s.get('vendor_csrmv'))"
  done 2>&1 | tee gpurun_out/r04/spmv_split.txt ;;
bench)
  bash tools/gpu_bench_profile.sh r04 2>&1 | tail -3
  python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["timing"]["reference_compatible_ms"], d["timing"]["alloc_async_ms"], d["roofline"]["frac"], d["runtime"]["system_rocm_runtime"])
for c in (d.get("configs") or {}).get("cases", []):
    print({k: c.get(k) for k in ("case", "ms", "gflops", "nnz_C", "structure_check", "traffic_over_compulsory", "skipped", "error")}, (c.get("roofline") or {}).get("frac"))
for k in ("spmv", "spmv_hbm"):
    s = d[k]; print(k, {q: s.get(q) for q in ("ms_per_spmv", "value", "frac_hbm_peak", "host_us_per_spmv")}, s.get("vendor_csrmv"))
print("cpu", d.get("cpu_baseline"))
PY
  tail -5 gpurun_out/r04/bench.err ;;
profiles)
  bash tools/profile_configs.sh r04 webbase1m stencil rmat18 rmat22 cant_irr 2>&1 | grep -E "^== " ;;
esac
done
