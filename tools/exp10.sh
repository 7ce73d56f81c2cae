#!/bin/bash
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -vE "^Read mtx" | tail -6
