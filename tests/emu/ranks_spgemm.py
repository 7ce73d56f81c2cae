#!/usr/bin/env python3
"""The native row-partitioned SpGEMM at world > 1 with ranks as THREADS of this process -- for the CPU emulation
(NSPARSE_LIB_DIR=tests/emu/lib: EMU_DEVICES fake devices, in-process RCCL stand-in), where it is the only way the
collective half of include/nsparse_dist.h (communicator by unique id, nsparse_dist_spgemm_gather: size all-reduce,
agreed allocation, per-rank broadcasts, row-pointer shift) can execute at world > 1 without 2+ GPUs.  On a real
multi-GPU box the same script runs one thread per GPU.

    python tests/emu/ranks_spgemm.py <world> <kind> <p0> <p1> <p2> [d|s]   -> one JSON line
"""
import ctypes as C
import json
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nsparse_amd as ns  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402  (checker only)


def main():
    world, kind, p0, p1, p2 = (int(v) for v in sys.argv[1:6])
    prec = sys.argv[6] if len(sys.argv) > 6 else "d"
    lib, dl, orc = ns.load(prec), ns.load_dist(prec), Oracle(prec)
    m = ns.sfCSR()
    lib.nsparse_synth_csr(C.byref(m), kind, p0, p1, p2, 0x5EED0022, 0, 0)
    A = lib.csr_host_to_numpy(m)
    lib.release_cpu_csr(m)
    ref = orc.spgemm(A, A)
    host = lib.csr_from_numpy(A["rpt"], A["col"], A["val"], A["N"])
    work = np.zeros(A["M"], dtype=np.int64)
    assert dl.nsparse_dist_spgemm_row_work(C.byref(host), C.byref(host), work.ctypes.data_as(C.POINTER(C.c_longlong))) == 0
    cuts = dl.partition_work(work, world)
    ident = C.create_string_buffer(ns.DIST_ID_BYTES)
    assert dl.nsparse_dist_unique_id(ident) == 0
    lib.hip.hipSetDevice.argtypes = [C.c_int]
    out, errs = [None] * world, []

    def rank(r):
        try:
            assert lib.hip.hipSetDevice(r % max(1, int(dl.nsparse_dist_device_count()))) == 0
            h = C.c_void_p()
            assert dl.nsparse_dist_init(C.byref(h), ident, r, world) == 0, "communicator"
            b_dev = lib.csr_from_numpy(A["rpt"], A["col"], A["val"], A["N"])
            lib.csr_memcpy(C.byref(b_dev))
            blk = ns.sfCSR()
            assert dl.nsparse_dist_csr_row_block(C.byref(host), int(cuts[r]), int(cuts[r + 1]), C.byref(blk)) == 0
            lib.csr_memcpy(C.byref(blk))
            c, cf = ns.sfCSR(), ns.sfCSR()
            assert dl.nsparse_dist_spgemm(h, C.byref(blk), C.byref(b_dev), C.byref(c)) == 0
            assert dl.nsparse_dist_barrier(h) == 0
            v = (C.c_double * 2)(float(c.nnz), float(r))
            assert dl.nsparse_dist_allreduce_f64(h, v, 2, 0) == 0
            rc = dl.nsparse_dist_spgemm_gather(h, cuts.ctypes.data_as(ns.capi.c_int_p), C.byref(c), C.byref(cf))
            assert rc == 0, f"gather -> {rc}"
            got = dict(M=cf.M, N=cf.N, nnz=cf.nnz, rpt=lib.d2h(cf.d_rpt, (cf.M + 1,), np.int32),
                       col=lib.d2h(cf.d_col, (cf.nnz,), np.int32), val=lib.d2h(cf.d_val, (cf.nnz,), lib.real))
            out[r] = dict(nnz_sum=int(v[0]), rank_sum=int(v[1]), rpt_ok=bool(np.array_equal(got["rpt"], ref["rpt"])),
                          col_ok=bool(np.array_equal(got["col"], ref["col"])), val_fails=int(orc.check_spgemm(got, ref)),
                          block_rows=int(c.M), block_nnz=int(c.nnz))
            dl.nsparse_dist_release_gathered(cf)
            for x in (c, blk, b_dev):
                lib.release_csr(x)
            lib.release_cpu_csr(blk)
            dl.nsparse_dist_destroy(h)
        except Exception as e:  # reported below
            errs.append((r, repr(e)))
    th = [threading.Thread(target=rank, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    print(json.dumps({"world": world, "M": int(A["M"]), "nnz_C": int(ref["nnz"]), "cuts": [int(c) for c in cuts], "ranks": out,
                      "errors": errs}))
    return 1 if errs else 0


if __name__ == "__main__":
    sys.exit(main())
