#!/bin/bash
# (NSPARSE_SPMV_PIPE / NSPARSE_SPMV_PLAIN: the library must be built with EXTRA=-DNSPARSE_EXPERIMENTS, see csrc/Makefile)
# usage: bash tools/quick_spmv.sh "NSPARSE_SPMV_PIPE=0" "NSPARSE_SPMV_PIPE=1" ...  (one bench run per setting)
run() { echo "== $*"; env "$@" timeout 600 python bench.py --no-cpu --no-pmc --no-irregular --no-vendor --steps 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k in ('spmv','spmv_hbm'):
    s=d[k]; print(k, s['ms_per_spmv'], 'ms (events', s['ms_kernel_events'], ')', s['value'], 'GB/s frac', s['frac_hbm_peak'], s['plan'], 'fails', s.get('ans_check_fails'))"; }
for cfg in "$@"; do run $cfg; done
