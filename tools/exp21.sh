#!/bin/bash
export TMPDIR=/tmp
for c in rmat22 rmat18; do
NSPARSE_LIB_DIR=$PWD/nsparse_amd/lib_vprof NSPARSE_TB_PROF=1 timeout 300 python tools/one_call_cfg.py $c 2 2>&1 | grep -E "^\[tb\] numeric bin [34]" | tail -2
done
