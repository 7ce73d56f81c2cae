#!/usr/bin/env python3
"""B entries fetched per product in the heavy-row numeric kernels, counted on the CPU emulation (tests/emu).

    python tools/emu_fetch_counts.py [case ...]            cases: rmat14 rmat16 rmat18s rmat22s wide3m   (default: rmat14 rmat16 wide3m)
    EMU_LIB_DIR=tests/emu/lib_exp NSPARSE_HEAVY_FLAT=1 python tools/emu_fetch_counts.py
                                                           the same workloads through the stateless tiles (heavy_flat.h;
                                                           lib_exp = make -C tests/emu EXTRA=-DNSPARSE_EXPERIMENTS OUT=.../lib_exp)

Bytes, not clocks: how many column indices / values of B a kernel LOADS for every product it accumulates is a property of
the algorithm that a CPU can count (spgemm/common.h: NSP_COUNT, compiled in under NSP_EMU only).  A product needs one
column and one value of B; "bytes per product / (4 + w)" = 1.00 is the floor.  The answer of every run is checked against
the oracle.  Test infrastructure: loads tests/emu/lib, never the product library."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["NSPARSE_LIB_DIR"] = os.environ.get("EMU_LIB_DIR", os.path.join(ROOT, "tests", "emu", "lib"))
os.environ.setdefault("EMU_CLOCK_DIV", "2000")

import numpy as np  # noqa: E402

import nsparse_amd as ns  # noqa: E402
from gpu_util import spgemm, synth  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

FAMILIES = {0: "k_num_tiled", 1: "k_num_ranked", 2: "k_num_ranked<SYM>", 3: "k_num_flat", 4: "walk_products", 5: "k_num_ranked_flat", 7: "k_sym_flat"}
IN_EXTENT = {3: 0, 5: 1, 7: 2}  # family -> slot of family 6 holding "entries loaded that lie inside their extent"


def cases(lib):
    def wide3m():
        import scipy.sparse as sp
        rng = np.random.default_rng(77)
        m, k, n = 48, 3000, 3_000_000
        a = sp.random(m, k, density=120 / k, format="csr", random_state=rng, dtype=np.float64)
        b = sp.random(k, n, density=220 / n, format="csr", random_state=rng, dtype=np.float64)
        a.sort_indices()
        b.sort_indices()
        A = dict(M=m, N=k, rpt=a.indptr.astype(np.int32), col=a.indices.astype(np.int32), val=a.data)
        B = dict(M=k, N=n, rpt=b.indptr.astype(np.int32), col=b.indices.astype(np.int32), val=b.data)
        return A, B
    return {
        "rmat14": lambda: (synth(lib, 3, 14, 16, 0, seed=0x5EED0022),) * 2,      # heavy rows above the ranked tile capacity
        "rmat16": lambda: (synth(lib, 3, 16, 16, 0, seed=0x5EED0022),) * 2,      # hub rows of A beyond 4096 entries
        "rmat18s": lambda: (synth(lib, 3, 18, 0, 1500000, seed=0x5EED0022),) * 2,  # R-MAT-18 at a third of the edges
        "rmat22s": lambda: (synth(lib, 3, 22, 0, 1500000, seed=0x5EED0022),) * 2,  # config 5 at a fifth of its edges: 4 M columns, lists
        "wide3m": wide3m,                                                           # 3 M columns: ranked tiles + lists
    }


def main():
    want = [a for a in sys.argv[1:] if not a.startswith("-")] or ["rmat14", "rmat16", "wide3m"]
    lib = ns.load("d")
    emu = C.CDLL(os.path.join(os.environ["NSPARSE_LIB_DIR"], "libnsparse_d.so"))
    emu.emu_get_fetch_counts.argtypes = [C.POINTER(C.c_longlong), C.c_int]
    orc = Oracle("d")
    w = 8
    print("# B entries fetched per product, heavy-row numeric kernels, CPU emulation; NSPARSE_HEAVY_FLAT=%s NSPARSE_RANKED_DENS=%s"
          % (os.environ.get("NSPARSE_HEAVY_FLAT", "(default)"), os.environ.get("NSPARSE_RANKED_DENS", "(default)")))
    print("%-10s %-20s %12s %12s %12s %9s %9s %9s %8s %10s" % ("case", "kernel", "products", "B.col loads", "B.val loads",
                                                               "col/prod", "val/prod", "bytes/min", "tiles", "in-extent"))
    for name in want:
        A, B = cases(lib)[name]()
        buf = (C.c_longlong * 64)()
        emu.emu_get_fetch_counts(buf, 1)
        t = time.time()
        got, st = spgemm(lib, A, B)
        dt = time.time() - t
        emu.emu_get_fetch_counts(buf, 1)
        ref = orc.spgemm_omp(A, B) if hasattr(orc, "spgemm_omp") and A["M"] > 100000 else orc.spgemm(A, B)
        ok = (np.array_equal(got["rpt"], ref["rpt"]) and np.array_equal(got["col"], ref["col"])
              and orc.check_spgemm(got, dict(ref, M=A["M"])) == 0)
        for f, label in FAMILIES.items():
            ncol, nval, nprod, ntile = (buf[f * 4 + i] for i in range(4))
            if nprod == 0 and ncol == 0:
                continue
            # stateless kernels: entries inside their extent per product (the rest of "loads" are lanes of a 4-wide vector
            # load past the end of a short extent -- same 16-byte request, no further sector)
            inx = "%10.3f" % (buf[6 * 4 + IN_EXTENT[f]] / max(nprod, 1)) if f in IN_EXTENT else "%10s" % "-"
            print("%-10s %-20s %12d %12d %12d %9.3f %9.3f %9.3f %8d %s" % (
                name, label, nprod, ncol, nval, ncol / max(nprod, 1), nval / max(nprod, 1),
                (4 * ncol + w * nval) / max((4 + w) * nprod, 1), ntile, inx))
        print("# %s: M %d nnz(A) %d nnz(C) %d heavy rows %d parity %s (%.0f s)" % (
            name, A["M"], len(A["col"]), got["nnz"], st.num_bin_size[5], "ok" if ok else "FAILED", dt))
        if not ok:
            sys.exit(1)
        sys.stdout.flush()


if __name__ == "__main__":
    main()
