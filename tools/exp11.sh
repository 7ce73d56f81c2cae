#!/bin/bash
export TMPDIR=/tmp
for c in stencil webbase1m rmat22 rmat18 cant_irr brick40; do
  for v in lib lib_v2 lib_v1; do
    echo "=== $c $v"; NSPARSE_LIB_DIR=$PWD/nsparse_amd/$v NSPARSE_RUN_CHECK=$([ $v = lib_v1 ] && echo 1 || echo 0) timeout 300 python tools/run_configs.py $c 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('ms','gflops','rpt_ok','col_ok','val_fails')})"
  done
done
for c in stencil webbase1m; do for v in lib lib_v1; do echo "=== serial $c $v"; NSPARSE_LIB_DIR=$PWD/nsparse_amd/$v timeout 300 python tools/one_call_cfg.py $c 3 2>&1 | tail -1 | cut -c1-700; done; done
