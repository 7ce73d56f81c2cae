"""Helpers for the -m gpu parity tests: everything goes through the C-ABI."""
import ctypes as C
import os

import numpy as np

import nsparse_amd as ns


def spgemm(lib, A, B=None, numeric_again=False):
    """C = A B through csr_memcpy / spgemm_kernel_hash / csr_memcpyDtH.  Returns (C dict, stats)."""
    B = A if B is None else B
    a = lib.csr_from_numpy(A["rpt"], A["col"], A["val"], A["N"])
    b = lib.csr_from_numpy(B["rpt"], B["col"], B["val"], B["N"])
    c = ns.sfCSR()
    lib.csr_memcpy(C.byref(a))
    lib.csr_memcpy(C.byref(b))
    flop = C.c_longlong()
    lib.get_spgemm_flop(C.byref(a), C.byref(b), a.M, C.byref(flop))
    lib.spgemm_kernel_hash(C.byref(a), C.byref(b), C.byref(c))
    st = ns.SpgemmStats()
    lib.nsparse_get_spgemm_stats(C.byref(st))
    lib.csr_memcpyDtH(C.byref(c))
    out = lib.csr_host_to_numpy(c)
    lib.release_cpu_csr(c)
    out["flop"] = flop.value
    if numeric_again:
        # numeric-only re-run on the kept structure with a poisoned value array
        poison = np.full(max(c.nnz, 1), np.nan, dtype=lib.real)
        lib.h2d(c.d_val, poison)
        lib.nsparse_spgemm_hash_numeric(C.byref(a), C.byref(b), C.byref(c))
        out["val_again"] = lib.d2h(c.d_val, (c.nnz,), lib.real)
        out["col_again"] = lib.d2h(c.d_col, (c.nnz,), np.int32)
    lib.release_csr(c)
    lib.release_csr(a)
    lib.release_csr(b)
    return out, st


def oracle_fp64_accumulated(orc_d, A, B=None):
    """Reference for the FLOAT build: the float inputs multiplied and summed in double (products of two
    floats are exact in double), the sums rounded to float once.  The library multiplies in float and
    accumulates in double, so against this reference the reference's own 1e-6 rule
    (nsparse.cu:300-353) holds; the float oracle sums in float in CSR order and is itself up to
    sqrt(products) * 6e-8 away from the exact sum."""
    B = A if B is None else B
    up = lambda m: dict(m, val=np.asarray(m["val"], dtype=np.float64))
    ref = orc_d.spgemm(up(A), up(B))
    return dict(ref, val=ref["val"].astype(np.float32))


class DeviceAMB:
    """CSR on the device converted to AMB; y = A x through sf_spmv_amb."""

    def __init__(self, lib, A, seg_size=None, block_size=None, chunk=64):
        self.lib, self.A = lib, A
        lib.nsparse_set_amb_chunk(chunk)
        self.csr = lib.csr_from_numpy(A["rpt"], A["col"], A["val"], A["N"])
        lib.csr_memcpy(C.byref(self.csr))
        w = lib.real().itemsize
        self.d_x = lib.dmalloc((A["N"] + 20) * w)
        self.d_y = lib.dmalloc((A["M"] + 64) * w)
        # poison the over-allocated tails: the kernel must not depend on them
        lib.h2d(self.d_x, np.full(A["N"] + 20, np.nan, dtype=lib.real))
        self.plan = ns.sfPlan()
        if seg_size is None:
            lib.init_plan(C.byref(self.plan))
        else:
            lib.set_plan(C.byref(self.plan), seg_size, block_size)
        self.amb = ns.sfAMB()
        lib.h2d(self.d_x, np.zeros(A["N"], dtype=lib.real))
        lib.sf_csr2amb(C.byref(self.amb), C.byref(self.csr), self.d_x, C.byref(self.plan))

    def arrays(self):
        return self.lib.amb_to_numpy(self.amb)

    def spmv(self, x):
        lib = self.lib
        lib.h2d(self.d_x, np.ascontiguousarray(x, dtype=lib.real))
        lib.h2d(self.d_y, np.full(self.A["M"] + 64, 7.0, dtype=lib.real))
        lib.sf_spmv_amb(self.d_y, C.byref(self.amb), self.d_x, C.byref(self.plan))
        y = lib.d2h(self.d_y, (self.A["M"] + 64,), lib.real)
        assert (y[self.A["M"]:] == 7.0).all(), "kernel wrote past y[M-1]"
        return y[:self.A["M"]]

    def close(self):
        lib = self.lib
        lib.release_amb(self.amb)
        lib.release_csr(self.csr)
        lib.dfree(self.d_x)
        lib.dfree(self.d_y)
        lib.nsparse_set_amb_chunk(64)


def synth(lib, kind, p0, p1=0, p2=0, seed=1, rows=(0, 0)):
    m = ns.sfCSR()
    lib.nsparse_synth_csr(C.byref(m), kind, p0, p1, p2, seed, rows[0], rows[1])
    A = lib.csr_host_to_numpy(m)
    lib.release_cpu_csr(m)
    return A


def row_windows(A, B):
    """Per row of C = A B: (products, column span) -- numpy restatement used by tests/bench to
    predict the bin of every row."""
    alen = np.diff(A["rpt"]).astype(np.int64)
    blen = np.diff(B["rpt"]).astype(np.int64)
    K = len(blen)
    bmin = np.full(K, np.iinfo(np.int64).max, dtype=np.int64)
    bmax = np.full(K, -1, dtype=np.int64)
    nz = blen > 0
    starts = B["rpt"][:-1].astype(np.int64)
    if len(B["col"]):
        bmin[nz] = np.minimum.reduceat(B["col"].astype(np.int64), starts[nz])
        bmax[nz] = np.maximum.reduceat(B["col"].astype(np.int64), starts[nz])
    M = len(alen)
    prod = np.zeros(M, dtype=np.int64)
    lo = np.full(M, np.iinfo(np.int64).max, dtype=np.int64)
    hi = np.full(M, -1, dtype=np.int64)
    ne = alen > 0
    if len(A["col"]):
        st = A["rpt"][:-1].astype(np.int64)[ne]
        ac = A["col"].astype(np.int64)
        prod[ne] = np.add.reduceat(blen[ac], st)
        lo[ne] = np.minimum.reduceat(bmin[ac], st)
        hi[ne] = np.maximum.reduceat(bmax[ac], st)
    span = np.where(hi >= lo, hi - lo + 1, 0)
    return prod, span


def twin_rows(A, prod, tiny=32):
    """Rows of A whose column pattern (the stored sequence of column ids) another row already has
    (twin_probe in csrc/spgemm/setup.h): all but one row of every pattern class are left out of the
    symbolic bins and take that row's structure.  Rows with at most `tiny` products (prod: products per
    row; the tiny symbolic bin) are not probed.  Which row of a class leads is not fixed on the device;
    the mask marks all but the first, the counts are the same."""
    rpt = np.asarray(A["rpt"], dtype=np.int64)
    col = np.ascontiguousarray(A["col"], dtype=np.int32)
    M = len(rpt) - 1
    tw = np.zeros(M, dtype=bool)
    if M >= 131072 and os.environ.get("NSPARSE_TWIN_SAMPLE", "1") != "0":
        # big matrices (setup.h: TwinSample): the map is used only if a pattern occurs twice among the sampled rows --
        # the first 64 of every 1024, with 1 .. 4096 entries
        seen, repeat = set(), False
        for r in range(M):
            if (r & 1023) < 64 and 1 <= rpt[r + 1] - rpt[r] <= 4096:
                k = col[rpt[r]:rpt[r + 1]].tobytes()
                if k in seen:
                    repeat = True
                    break
                seen.add(k)
        if not repeat:
            return tw
    seen = set()
    for r in range(M):
        if rpt[r + 1] > rpt[r] and prod[r] > tiny:
            k = col[rpt[r]:rpt[r + 1]].tobytes()
            if k in seen:
                tw[r] = True
            else:
                seen.add(k)
    return tw


def bins_of(n, span, ladder, work=None):
    """numpy twin of bin_of() in csrc/spgemm/common.h; ladder = 15 ints from nsparse_get_spgemm_bins.
    work = products of the row (numeric phase; the symbolic phase bins by the products themselves)."""
    n = np.asarray(n, dtype=np.int64)
    span = np.asarray(span, dtype=np.int64)
    work = n if work is None else np.asarray(work, dtype=np.int64)
    tiny, hash_t, dspan, ratio = ladder[0], ladder[1:5], ladder[5:8], ladder[8]
    b = 1 + sum((n > t).astype(np.int64) for t in hash_t)
    if len(ladder) > 9 and ladder[11] > 0:
        bspan, bratio, bmin = ladder[9:11], ladder[11], ladder[12]
        if len(ladder) > 13 and ladder[13] > 0:
            b = np.where((n > ladder[13]) & (span > 0) & (span <= ladder[14]), 10, b)
        bits = (n > bmin) & (span > 0) & (span <= bspan[1]) & (span <= bratio * n)
        b = np.where(bits, 9 + (span > bspan[0]), b)
    if len(ladder) > 15 and ladder[15] > 0:  # ranked window (numeric ladder): after the dense rule, before the rest
        rank = (span > 0) & (span <= ladder[15]) & (n <= ladder[17]) & ((span <= ladder[16] * n) | (span <= 4 * work))
        b = np.where(rank, 9, b)
    if ratio > 0:
        dense = (span > 0) & (span <= dspan[2]) & ((span <= ratio * n) | (4 * span <= ratio * work))
        b = np.where(dense, 6 + (span > dspan[0]) + (span > dspan[1]), b)
    return np.where(n <= tiny, 0, b)


def numeric_bins(row_nz, row_prod, span, sym, num):
    """Numeric bin of every row in a FULL spgemm_kernel_hash call: a row can use the numeric dense
    window only if the symbolic dense kernel wrote its column bitmap (symbolic bin >= 6) and the
    window fits the numeric ladder; everything else is binned by nnz alone."""
    span = np.asarray(span, dtype=np.int64)
    sb = bins_of(row_prod, span, sym)
    limit = max(num[7], num[15] if len(num) > 15 else 0)  # widest window a bitmap is handed over for
    has_bm = (sb >= 6) & (sb <= 8) & (span <= limit)
    return bins_of(row_nz, np.where(has_bm, span, 0), num, work=row_prod)


def ladders(lib):
    sym = (C.c_int * 18)()
    num = (C.c_int * 18)()
    lib.nsparse_get_spgemm_bins(sym, num)
    return list(sym), list(num)


def experiments_lib_dir():
    """The -DNSPARSE_EXPERIMENTS sibling of the library directory in use (nsparse_amd/lib -> nsparse_amd/lib_exp,
    tests/emu/lib -> tests/emu/lib_exp: __graft_entry__.build() makes both), or None when it has not been built."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cur = os.environ.get("NSPARSE_LIB_DIR") or os.path.join(root, "nsparse_amd", "lib")
    exp = cur.rstrip("/") + "_exp"
    return exp if os.path.exists(os.path.join(exp, "libnsparse_d.so")) else None


# Switches that only the -DNSPARSE_EXPERIMENTS build reads (csrc/internal.h: exp_env): measurement knobs and the opt-in
# kernel families.  The product library treats them as constants, so a test that sets one runs on the variant library.
EXPERIMENT_SWITCHES = ("NSPARSE_TB_LEAN", "NSPARSE_RANKED_DENS", "NSPARSE_FUSED_FORCE", "NSPARSE_FUSED_BIG", "NSPARSE_TWIN_SAMPLE",
                       "NSPARSE_SYM_CURSOR", "NSPARSE_SPMV_SPLIT", "NSPARSE_SPMV_PIPE", "NSPARSE_SPMV_PLAIN", "NSPARSE_SPMV_ABL",
                       "NSPARSE_SPMV_REMAP", "NSPARSE_SPMV_NOREMAP", "NSPARSE_KEYED", "NSPARSE_KEYED_B", "NSPARSE_TILED",
                       "NSPARSE_HEAVY_FLAT", "NSPARSE_WINDOW_KERNEL", "NSPARSE_FLAT", "NSPARSE_BLK_DESC", "NSPARSE_STREAM_PRIO",
                       "NSPARSE_PRIO_BINS")


def variant_env(env, base=None):
    """`env` (additions to the environment of a child process) with NSPARSE_LIB_DIR pointing at the experiments variant
    of the library in use (or of `base`, a library directory) when it sets a switch only that build reads; skips the
    test when the variant has not been built."""
    if not any(k in EXPERIMENT_SWITCHES for k in env) or "NSPARSE_LIB_DIR" in env and env["NSPARSE_LIB_DIR"].rstrip("/").endswith("_exp"):
        return env
    if base is not None:
        d = base.rstrip("/") + "_exp"
        d = d if os.path.exists(os.path.join(d, "libnsparse_d.so")) else None
    else:
        d = experiments_lib_dir()
    if d is None:
        import pytest
        pytest.skip("needs the -DNSPARSE_EXPERIMENTS variant library (__graft_entry__.build() makes lib_exp)")
    return dict(env, NSPARSE_LIB_DIR=d)


def spgemm_subprocess(A, env, prec="d", B=None, numeric_again=False):
    """Run spgemm() on A (times B, default A) in a fresh interpreter with extra environment (the
    library reads its tuning switches once per process).  Returns (C dict, dict of stats lists)."""
    import json
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = variant_env(env)
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, "a.npz"), rpt=A["rpt"], col=A["col"], val=A["val"], M=A["M"], N=A["N"])
        Bm = A if B is None else B
        np.savez(os.path.join(td, "b.npz"), rpt=Bm["rpt"], col=Bm["col"], val=Bm["val"], M=Bm["M"], N=Bm["N"])
        code = (
            "import sys, json, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r);"
            "import nsparse_amd as ns; from gpu_util import spgemm;"
            "ld = lambda f: (lambda z: dict(rpt=z['rpt'], col=z['col'], val=z['val'], M=int(z['M']), N=int(z['N'])))(np.load(f));"
            "A = ld(%r); B = ld(%r);"
            "got, st = spgemm(ns.load(%r), A, B, numeric_again=%r);"
            "np.savez(%r, **{k: got[k] for k in ('rpt', 'col', 'val', 'val_again', 'col_again') if k in got});"
            "print(json.dumps(dict(sym=list(st.sym_bin_size), num=list(st.num_bin_size), fails=st.sym_fail_rows,"
            " build=ns.load(%r).nsparse_build_info().decode())))"
        ) % (root, os.path.join(root, "tests"), os.path.join(td, "a.npz"), os.path.join(td, "b.npz"), prec,
             bool(numeric_again), os.path.join(td, "c.npz"), prec)
        r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True,
                           env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-3000:]
        stats = json.loads(r.stdout.strip().splitlines()[-1])
        z = np.load(os.path.join(td, "c.npz"))
        got = dict(M=A["M"], N=Bm["N"], nnz=int(z["rpt"][-1]), rpt=z["rpt"], col=z["col"], val=z["val"])
        if numeric_again:
            got["val_again"], got["col_again"] = z["val_again"], z["col_again"]
    return got, stats
