#!/usr/bin/env python3
"""The kernels of bench.py's workloads, launched a few times each, WITHOUT torch: the command
bench.py wraps in `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` to read the HBM traffic of its
dominant kernels on the input it just timed (roofline.traffic).  Also usable by hand:
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o p -- python tools/pmc_one.py bench
Workloads: bench = cant-class brick SpGEMM + its AMB SpMV + the nlpkkt-class AMB SpMV;
           irregular = the irregular cant-class SpGEMM; spgemm = the brick SpGEMM only."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nsparse_amd as ns  # noqa: E402
from bench import STANDINS, synth  # noqa: E402

REPS = 6


def spgemm(lib, A):
    a = lib.csr_from_numpy(A["rpt"], A["col"], A["val"], A["N"])
    b = lib.csr_from_numpy(A["rpt"], A["col"], A["val"], A["N"])
    lib.csr_memcpy(C.byref(a))
    lib.csr_memcpy(C.byref(b))
    c = ns.sfCSR()
    for _ in range(REPS):
        lib.spgemm_kernel_hash(C.byref(a), C.byref(b), C.byref(c))
        lib.release_csr(c)
    lib.release_csr(a)
    lib.release_csr(b)


def spmv(lib, A):
    m = lib.csr_from_numpy(A["rpt"], A["col"], A["val"], A["N"])
    lib.csr_memcpy(C.byref(m))
    w = 8
    d_x = lib.dmalloc((A["N"] + 20) * w)
    d_y = lib.dmalloc((A["M"] + 64) * w)
    lib.h2d(d_x, np.random.default_rng(1).random(A["N"] + 20))
    plan, amb = ns.sfPlan(), ns.sfAMB()
    lib.init_plan(C.byref(plan))
    lib.sf_csr2amb(C.byref(amb), C.byref(m), d_x, C.byref(plan))
    for _ in range(REPS):
        lib.sf_spmv_amb(d_y, C.byref(amb), d_x, C.byref(plan))
    lib.release_amb(amb)
    lib.release_csr(m)
    lib.dfree(d_x)
    lib.dfree(d_y)


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "bench"
    lib = ns.load("d")
    if what in ("bench", "spgemm"):
        k, p, s = STANDINS["cant"]
        A = synth(lib, k, *p, s)
        spgemm(lib, A)
        if what == "bench":
            spmv(lib, A)
            k, p, s = STANDINS["nlpkkt120"]
            spmv(lib, synth(lib, k, *p, s))
    elif what == "spmv_hbm":
        k, p, s = STANDINS["nlpkkt120"]
        spmv(lib, synth(lib, k, *p, s))
    elif what == "irregular":
        k, p, s = STANDINS["cant_irregular"]
        spgemm(lib, synth(lib, k, *p, s))
    else:
        raise SystemExit(f"unknown workload {what}")


if __name__ == "__main__":
    main()
