#!/bin/bash
# randomised soak of the parity sweep beyond its 40 committed cases (tests/test_fuzz_gpu.py)
export TMPDIR=/tmp
run() { echo "== $*"; env "$@" timeout 1500 python -m pytest tests/test_fuzz_gpu.py -x -q 2>&1 | grep -vE "^Read mtx" | grep -E "passed|failed|Error|assert" | tail -3; }
run NSPARSE_FUZZ_SEEDS=2000 NSPARSE_FUZZ_BASE=300000
run NSPARSE_FUZZ_SEEDS=1200 NSPARSE_FUZZ_BASE=310000 NSPARSE_FUZZ_PREC=s
run NSPARSE_FUZZ_SEEDS=600 NSPARSE_FUZZ_BASE=320000 NSPARSE_TB_BUCKET=1 NSPARSE_LIST=2
run NSPARSE_FUZZ_SEEDS=600 NSPARSE_FUZZ_BASE=330000 NSPARSE_FLAT=1 NSPARSE_TB_PERSIST=2
run NSPARSE_FUZZ_SEEDS=400 NSPARSE_FUZZ_BASE=340000 NSPARSE_FUSED=0 NSPARSE_FLAT=0
