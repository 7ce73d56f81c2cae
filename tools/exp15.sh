#!/bin/bash
export TMPDIR=/tmp
NSPARSE_LIST=1 NSPARSE_LIST_DRY=1 timeout 300 python tools/one_call_cfg.py rmat22 3 2>&1 | tail -1 | cut -c1-700
