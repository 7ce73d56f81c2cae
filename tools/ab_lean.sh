#!/bin/bash
# A/B of the lean hash kernels (NSPARSE_TB_LEAN=0: round-3 kernels, 3: lean) on the power-law / stencil cases.
# usage: bash tools/ab_lean.sh [cases...]   -> gpurun_out/ab_lean.log
out=gpurun_out/ab_lean.log; : > $out
cases=${@:-stencil webbase1m rmat18 rmat22}
for c in $cases; do
  for lean in 0 3; do
    echo "== $c TB_LEAN=$lean" >> $out
    NSPARSE_RUN_CHECK=${CHECK:-0} NSPARSE_TB_LEAN=$lean bash tools/quick_bench.sh $c >> $out 2>&1
    NSPARSE_BIN_TIMING=0 NSPARSE_RUN_CHECK=0 NSPARSE_TB_LEAN=$lean bash tools/quick_bench.sh $c >> $out 2>&1
  done
done
cat $out
