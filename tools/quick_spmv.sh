#!/bin/bash
run() { echo "== $*"; env "$@" timeout 600 python bench.py --no-cpu --steps 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k in ('spmv','spmv_hbm'):
    s=d[k]; print(k, s['ms_per_spmv'], 'ms', s['value'], 'GB/s frac', s['frac_hbm_peak'], s['plan'], 'fails', s['ans_check_fails'])"; }
for cfg in "$@"; do run $cfg; done
