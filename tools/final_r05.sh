#!/bin/bash
# The prepared GPU run (rounds 5-6), ONE gpurun call; stages can be picked: bash tools/final_r05.sh [tests] [smoke] [bench] [ab] [spmv] [profiles] [vprofiles]
#   tests     the whole -m gpu suite, no -x (every failure is listed); each of the first failures is then re-run
#             against the variant libraries nsparse_amd/lib_<commit>/ (built beforehand from git worktrees of the
#             commits between the last proven tree and HEAD), so a red test names the commit that broke it
#   smoke     __graft_entry__.smoke()
#   bench     the bench line + kernel trace of the same command -> gpurun_out/r05/
#   ab        tools/ab_variants.sh: default kernels vs the lean hash forms and the stateless heavy-row tiles (one experiments library), serialised per-bin times
#   spmv      cache-resident SpMV: split-row kernel widths against rocSPARSE csrmv
#   profiles  per-config kernel stats + PMC (tools/profile_configs.sh)
#   vprofiles the same for the R-MAT cases through the stateless heavy-row kernels of the variant library
#             (nsparse_amd/lib_exp, NSPARSE_HEAVY_FLAT=7) -> gpurun_out/<tag>_flat/: kernel times AND the HBM traffic
#             of k_num_flat / k_num_ranked_flat / k_sym_flat next to the cursor kernels' -- the A/B on bytes, not only on time
export TMPDIR=/tmp
TAG=${NSPARSE_TAG:-r05}
O=gpurun_out/$TAG
mkdir -p $O
FILTER="^Read mtx|^RCCL|^HIP version|^ROCm version|^Hostname|^Librccl"
stages=${@:-tests smoke bench}
for s in $stages; do
echo "#### stage $s $(date +%T)"
case $s in
tests)
  timeout 2400 python -m pytest tests -m gpu -q -rf -p no:cacheprovider 2>&1 | grep -vE "$FILTER" > $O/tests_all.log
  tail -15 $O/tests_all.log
  grep "^FAILED" $O/tests_all.log | sed 's/^FAILED //; s/ - .*$//' | head -6 > $O/failed.txt
  if [ -s $O/failed.txt ]; then
    : > $O/bisect.txt
    while read -r t; do
      for v in $(ls -d nsparse_amd/lib_*/ 2>/dev/null | grep -E 'lib_[0-9a-f]{7}/$'); do
        NSPARSE_LIB_DIR=$PWD/${v%/} timeout 600 python -m pytest "$t" -m gpu -q -x -p no:cacheprovider 2>&1 | grep -vE "$FILTER" | tail -1 \
          | sed "s|^|$t @ $v: |" >> $O/bisect.txt
      done
      timeout 600 python -m pytest "$t" -m gpu -q -x -p no:cacheprovider 2>&1 | grep -vE "$FILTER" | tail -60 > "$O/fail_$(echo "$t" | tr -c 'A-Za-z0-9_\n' _).log"
    done < $O/failed.txt
    cat $O/bisect.txt
  fi ;;
smoke)
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -vE "$FILTER" | tail -5 | tee $O/smoke.log ;;
bench)
  bash tools/gpu_bench_profile.sh $TAG 2>&1 | tail -3
  python - "$O" <<'PY'
import json, sys
O = sys.argv[1]
try:
    d = json.loads(open(O + "/bench.json").read().strip().splitlines()[-1])
except Exception as e:
    print("NO BENCH LINE:", repr(e)); print(open(O + "/bench.err").read()[-3000:]); sys.exit(0)
print({k: d[k] for k in ("metric", "value", "ms_per_step", "dtype")}, d["timing"].get("reference_compatible_ms"), d["timing"].get("alloc_async_ms"))
print("roofline", d.get("roofline"))
print("runtime", d.get("runtime"))
for c in (d.get("configs") or {}).get("cases", []):
    print({k: c.get(k) for k in ("case", "ms", "gflops", "nnz_C", "structure_check", "traffic_over_compulsory", "skipped", "error")}, (c.get("roofline") or {}).get("frac"))
for k in ("spmv", "spmv_hbm"):
    s = d.get(k) or {}; print(k, {q: s.get(q) for q in ("ms_per_spmv", "value", "frac_hbm_peak", "host_us_per_spmv")}, s.get("vendor_csrmv"))
print("cpu", d.get("cpu_baseline"))
print("driver_run_s", d.get("driver_run_s"))
PY
  tail -5 $O/bench.err ;;
ab)
  bash tools/ab_variants.sh stencil webbase1m rmat18 rmat22 2>&1 | tail -40 | tee $O/ab_variants.txt
  cp gpurun_out/ab_variants.log $O/ab_variants.log ;;
spmv)
  for sp in 0 1 2 4 8; do
    echo "== NSPARSE_SPMV_SPLIT=$sp"
    NSPARSE_LIB_DIR=$PWD/nsparse_amd/lib_exp NSPARSE_SPMV_SPLIT=$sp timeout 600 python bench.py --no-cpu --no-pmc --no-irregular --no-configs --no-large --steps 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['spmv']
print({k:s.get(k) for k in ('ms_per_spmv','ms_kernel_events','value','plan','ans_check_fails')}, s.get('vendor_csrmv'))"
  done 2>&1 | tee $O/spmv_split.txt ;;
profiles)
  bash tools/profile_configs.sh $TAG webbase1m stencil rmat18 rmat22 cant_irr 2>&1 | grep -E "^== " ;;
vprofiles)
  NSPARSE_LIB_DIR=$PWD/nsparse_amd/lib_exp NSPARSE_HEAVY_FLAT=7 bash tools/profile_configs.sh ${TAG}_flat rmat18 rmat22 2>&1 | grep -E "^== " ;;
esac
done
echo "#### done $(date +%T)"
