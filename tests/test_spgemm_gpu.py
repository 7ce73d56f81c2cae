"""Parity of the HIP hash SpGEMM with the CPU oracle, through the C-ABI.

Rules (reference check_spgemm_answer, nsparse.cu:300-353): nnz, rpt and col EXACT (ascending
columns), values within 1e-9 relative (double) / 1e-6 (float)."""
import ctypes as C
import os

import numpy as np
import pytest

import nsparse_amd as ns
from conftest import GOLDEN, TEST_MTX, load_golden
from gpu_util import (bins_of, ladders, numeric_bins, oracle_fp64_accumulated, row_windows, spgemm, spgemm_subprocess,
                      synth, twin_rows)

pytestmark = pytest.mark.gpu


def assert_parity(orc, got, ref, signed=False):
    assert got["nnz"] == ref["nnz"]
    assert np.array_equal(got["rpt"], ref["rpt"]), "C.rpt differs"
    assert np.array_equal(got["col"], ref["col"]), "C.col differs"
    if not signed:
        assert orc.check_spgemm(got, ref) == 0, "values outside the reference tolerance"
    else:
        # cancelling sums: compare against the magnitude of the products instead
        eps = 1e-12 if orc.precision == "d" else 1e-5
        scale = np.abs(ref["val"]).max()
        assert np.abs(got["val"] - ref["val"]).max() <= eps * scale * 64


def test_test_mtx_known_answer(lib_d, oracle_d):
    m = ns.sfCSR()
    lib_d.init_csr_matrix_from_file(C.byref(m), os.path.join(GOLDEN, "test.mtx").encode())
    A = lib_d.csr_host_to_numpy(m)
    lib_d.release_cpu_csr(m)
    got, st = spgemm(lib_d, A)
    assert got["rpt"].tolist() == TEST_MTX["c_rpt"]
    assert got["col"].tolist() == TEST_MTX["c_col"]
    assert got["val"].tolist() == TEST_MTX["c_val"]
    assert got["flop"] == 2 * TEST_MTX["n_prod"] and st.n_prod == TEST_MTX["n_prod"]
    assert list(st.sym_bin_size)[:2] == [5, 0] and list(st.num_bin_size)[:2] == [5, 0]


@pytest.mark.parametrize("prec", ["d", "s"])
@pytest.mark.parametrize("name", ["banded2k", "banded_signed1k", "rmat_s10"])
def test_golden_vectors(name, prec, lib_d, lib_s, oracle_d, oracle_s):
    lib, orc = (lib_d, oracle_d) if prec == "d" else (lib_s, oracle_s)
    g = load_golden(name)
    got, st = spgemm(lib, g, numeric_again=True)
    ref = dict(M=g["M"], nnz=len(g["c_col"]), rpt=g["c_rpt"], col=g["c_col"],
               val=g["c_val"].astype(lib.real))
    assert_parity(orc, got, ref, signed="signed" in name)
    assert got["flop"] == 2 * int(g["row_prod"].sum())
    # numeric-only re-run reproduces the values on the kept structure
    assert np.array_equal(got["col_again"], got["col"])
    np.testing.assert_allclose(got["val_again"], got["val"], rtol=1e-9 if prec == "d" else 1e-5,
                               atol=1e-12 if prec == "d" else 1e-5)


def _bins_match(orc, st, row_prod, row_nz, lib, A, B=None):
    B = A if B is None else B
    sym, num = ladders(lib)
    prod, span = row_windows(A, B)
    assert np.array_equal(prod, row_prod)
    tw = twin_rows(A, row_prod)  # not binned in the symbolic phase
    assert st.twin_rows == int(tw.sum())
    assert list(st.sym_bin_size)[:11] == np.bincount(bins_of(row_prod, span, sym)[~tw], minlength=11).tolist()
    assert list(st.num_bin_size)[:10] == np.bincount(numeric_bins(row_nz, row_prod, span, sym, num), minlength=10).tolist()


@pytest.mark.parametrize("kind,p,prec", [
    (0, (6, 6, 20), "d"),          # FEM brick: wide wave-per-row groups, numeric bins 1-2
    (0, (20, 20, 6), "d"),         # wider cross-section: 5 K-column windows, window bins 7 / 8 in both phases
    (0, (20, 20, 6), "s"),
    (1, (24, 24, 24), "d"),        # scalar stencil: 27-long B rows
    (2, (60000, 200000, 0), "s"),  # power law: bin 0 + heavy tail (webbase class, fp32)
    (2, (60000, 200000, 0), "d"),
    (3, (12, 8, 0), "d"),          # R-MAT: hub rows
])
def test_synthetic_vs_oracle(kind, p, prec, lib_d, lib_s, oracle_d, oracle_s):
    lib, orc = (lib_d, oracle_d) if prec == "d" else (lib_s, oracle_s)
    A = synth(lib, kind, *p, seed=0x5EED0022)
    ref = orc.spgemm(A, A)
    got, st = spgemm(lib, A)
    assert_parity(orc, got, ref)
    rp, tot, mx = orc.nprod(A["rpt"], A["col"], A["rpt"])
    assert st.n_prod == tot and st.max_prod_row == mx and st.nnz_c == ref["nnz"]
    assert st.max_nnz_row == int(ref["row_nz"].max())
    _bins_match(orc, st, rp, ref["row_nz"], lib, A)


def test_rectangular_and_empty_rows(lib_d, oracle_d):
    """A (M x K) * B (K x N) with K != M != N, empty rows in A and B, an all-empty C row."""
    rng = np.random.default_rng(7)
    import scipy.sparse as sp
    A = sp.random(300, 500, density=0.01, random_state=rng, format="csr")
    B = sp.random(500, 90000, density=0.0005, random_state=rng, format="csr")
    A.sort_indices(); B.sort_indices()
    da = dict(M=300, N=500, rpt=A.indptr.astype(np.int32), col=A.indices.astype(np.int32), val=A.data)
    db = dict(M=500, N=90000, rpt=B.indptr.astype(np.int32), col=B.indices.astype(np.int32), val=B.data)
    ref = oracle_d.spgemm(da, db)
    got, _ = spgemm(lib_d, da, db)
    assert got["M"] == 300 and got["N"] == 90000
    assert_parity(oracle_d, got, ref)
    assert (np.diff(got["rpt"]) == 0).any()


def test_empty_matrix(lib_d, oracle_d):
    A = dict(M=64, N=64, rpt=np.zeros(65, np.int32), col=np.zeros(0, np.int32), val=np.zeros(0))
    got, st = spgemm(lib_d, A)
    assert got["nnz"] == 0 and not got["rpt"].any() and st.n_prod == 0


def _force_rows(n_rows, n_cols, row_len, rng, extra=None):
    """square matrix whose first row touches `row_len` distinct long rows"""
    import scipy.sparse as sp
    A = sp.random(n_rows, n_cols, density=row_len / n_cols, random_state=rng, format="lil")
    if extra:
        for r, cols in extra.items():
            A[r, cols] = 1.0 + rng.random(len(cols))
    A = A.tocsr()
    A.sort_indices()
    return dict(M=n_rows, N=n_cols, rpt=A.indptr.astype(np.int32), col=A.indices.astype(np.int32),
                val=A.data.astype(np.float64))


def test_every_bin_is_exercised(lib_d, oracle_d):
    """Rows sized to land in every symbolic bin (0-5, incl. the try-in-LDS bin) and every numeric
    bin incl. the global-table bin; structure must still be exact."""
    rng = np.random.default_rng(11)
    n = 100000  # wider than the largest dense window (65536): the big rows must use the hash bins
    hub_cols = np.sort(rng.choice(n, 3500, replace=False))
    big_cols = np.sort(rng.choice(n, 400, replace=False))
    A = _force_rows(n, n, 12, rng, extra={0: hub_cols, 1: big_cols, 2: big_cols[:150], 3: big_cols[:60]})
    ref = oracle_d.spgemm(A, A)
    got, st = spgemm(lib_d, A)
    assert_parity(oracle_d, got, ref)
    rp, _, _ = oracle_d.nprod(A["rpt"], A["col"], A["rpt"])
    _bins_match(oracle_d, st, rp, ref["row_nz"], lib_d, A)
    assert st.sym_bin_size[5] + st.sym_bin_size[9] + st.sym_bin_size[10] >= 1, "no big symbolic row"
    assert st.num_bin_size[5] >= 1, "no row reached the global numeric bin"
    assert sum(1 for b in list(st.num_bin_size)[:9] if b > 0) >= 4
    # same rows through the global-memory hash table instead of the column-tiled LDS windows
    got_g, st_g = spgemm_subprocess(A, {"NSPARSE_TILED": "0"})
    assert st_g["num"][5] >= 1
    assert_parity(oracle_d, got_g, ref)


def test_unsorted_rows_of_b(lib_d, oracle_d):
    """Rows of B in arbitrary column order (legal CSR; the loader does not sort): windows come from
    min/max, the product walk does not care, the tiled kernel (needs sorted B) must step aside."""
    rng = np.random.default_rng(13)
    n = 30000
    hub = np.sort(rng.choice(n, 2500, replace=False))
    A = _force_rows(n, n, 10, rng, extra={3: hub})
    B = dict(A, col=A["col"].copy(), val=A["val"].copy())
    for i in range(n):
        b, e = B["rpt"][i], B["rpt"][i + 1]
        p = rng.permutation(e - b)
        B["col"][b:e], B["val"][b:e] = B["col"][b:e][p], B["val"][b:e][p]
    ref = oracle_d.spgemm(A, B)
    assert ref["row_nz"][3] > 5461
    got, st = spgemm(lib_d, A, B)
    assert st.num_bin_size[5] >= 1
    assert_parity(oracle_d, got, ref)


def test_symbolic_overflow_falls_back_to_global_table(lib_d, oracle_d):
    """A row with > 24576 distinct output columns.  Default build: the LDS bit window counts it
    (bin 9).  NSPARSE_DENSE=0: it must fail over from the 32768-key LDS table to the global table.
    Both give the oracle's structure."""
    rng = np.random.default_rng(5)
    n = 120000
    cols = np.sort(rng.choice(n, 3000, replace=False))
    A = _force_rows(n, n, 14, rng, extra={7: cols})
    ref = oracle_d.spgemm(A, A)
    assert ref["row_nz"][7] > 24576
    got, st = spgemm(lib_d, A)
    assert st.sym_bin_size[9] >= 1 and st.sym_fail_rows == 0
    assert_parity(oracle_d, got, ref)
    got0, st0 = spgemm_subprocess(A, {"NSPARSE_DENSE": "0"})
    assert st0["fails"] >= 1 and st0["sym"][5] >= 1 and sum(st0["sym"][6:]) == 0
    assert_parity(oracle_d, got0, ref)


def test_dense_window_and_hash_paths_agree(lib_d, oracle_d):
    """Banded / FEM rows take the dense-window bins (6-8); NSPARSE_DENSE=0 (separate process) sends
    the same rows through the hash bins.  Both must give the oracle's structure."""
    A = synth(lib_d, 0, 6, 6, 20, seed=5)
    got, st = spgemm(lib_d, A)
    ref = oracle_d.spgemm(A, A)
    assert_parity(oracle_d, got, ref)
    assert sum(list(st.sym_bin_size)[6:9]) > 0 and sum(list(st.num_bin_size)[6:9]) > 0
    got0, st0 = spgemm_subprocess(A, {"NSPARSE_DENSE": "0"})
    assert sum(st0["sym"][6:]) == 0 and sum(st0["num"][6:]) == 0
    assert_parity(oracle_d, got0, ref)
    np.testing.assert_allclose(got0["val"], got["val"], rtol=1e-9)


@pytest.mark.parametrize("kind,p", [(2, (40000, 130000, 0)), (3, (11, 8, 0)), (0, (5, 5, 12))])
def test_unsorted_output_mode(kind, p, lib_d, oracle_d):
    """nsparse_spgemm_set_sorted(0): same rpt, same SET of (col, val) per row, order free."""
    A = synth(lib_d, kind, *p, seed=21)
    ref = oracle_d.spgemm(A, A)
    old = lib_d.nsparse_spgemm_set_sorted(0)
    try:
        got, st = spgemm(lib_d, A)
    finally:
        lib_d.nsparse_spgemm_set_sorted(old)
    assert np.array_equal(got["rpt"], ref["rpt"])
    # sort every row by column, then the usual rule applies
    order = np.lexsort((got["col"], np.repeat(np.arange(A["M"]), np.diff(got["rpt"]))))
    canon = dict(got, col=got["col"][order], val=got["val"][order])
    assert_parity(oracle_d, canon, ref)
    if kind != 0:
        assert not np.array_equal(got["col"], ref["col"]), "expected at least one unsorted row"


def test_workspace_cache_off_is_identical(lib_d, oracle_d):
    g = load_golden("banded2k")
    lib_d.nsparse_set_workspace_cache(0)
    try:
        a, _ = spgemm(lib_d, g)
    finally:
        lib_d.nsparse_set_workspace_cache(1)
    b, _ = spgemm(lib_d, g)
    assert np.array_equal(a["col"], b["col"]) and np.array_equal(a["rpt"], b["rpt"])


def test_full_size_cant_class_fp32(lib_s, oracle_s):
    """The same matrix through the float build: structure exact, values to the reference's 1e-6
    (the LDS accumulators are double in this build too, so the sums are the better-rounded side)."""
    A = synth(lib_s, 0, 9, 9, 257, seed=0x5EED0022)
    got, st = spgemm(lib_s, A)
    ref = oracle_s.spgemm(A, A)
    assert st.num_bin_size[6] == A["M"]
    assert_parity(oracle_s, got, ref)


def test_full_size_cant_class_properties(lib_d, oracle_d):
    """BASELINE config 2 size (62,451 rows, ~4.3 M nnz, ~0.3 G products).  Oracle parity on the
    full matrix (the C oracle needs ~1 s) plus size-independent properties: ascending columns,
    symmetry of the structure of A^2 for symmetric A, linearity in the values."""
    A = synth(lib_d, 0, 9, 9, 257, seed=0x5EED0022)
    assert A["M"] == 62451
    got, st = spgemm(lib_d, A)
    ref = oracle_d.spgemm(A, A)
    assert_parity(oracle_d, got, ref)
    for i in range(0, A["M"], 997):
        c = got["col"][got["rpt"][i]:got["rpt"][i + 1]]
        assert (np.diff(c) > 0).all()
    import scipy.sparse as sp
    S = sp.csr_matrix((np.ones(got["nnz"]), got["col"], got["rpt"]), shape=(A["M"], A["N"]))
    assert (S != S.T).nnz == 0
    A2 = dict(A, val=A["val"] * 3.0)
    got2, _ = spgemm(lib_d, A2)
    np.testing.assert_allclose(got2["val"], 9.0 * got["val"], rtol=1e-9)


def test_nnz_c_beyond_int_is_refused():
    """nnz(C) >= 2^31 cannot be represented by sfCSR (int rpt / nnz, nsparse.h:62-75); upstream
    wraps silently.  Here: error -40, no C arrays, the true count in the statistics; and an abort
    without NSPARSE_NO_ABORT.  A = ones(M x 1), B = ones(1 x N): C is dense M x N."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, ctypes as C, numpy as np; sys.path.insert(0, %r);"
        "import nsparse_amd as ns; lib = ns.load('d'); M, N = 65536, 32769;"
        "a = lib.csr_from_numpy(np.arange(M + 1, dtype=np.int32), np.zeros(M, np.int32), np.ones(M), 1);"
        "b = lib.csr_from_numpy(np.array([0, N], np.int32), np.arange(N, dtype=np.int32), np.ones(N), N);"
        "lib.csr_memcpy(C.byref(a)); lib.csr_memcpy(C.byref(b)); c = ns.sfCSR();"
        "lib.spgemm_kernel_hash(C.byref(a), C.byref(b), C.byref(c));"
        "st = ns.SpgemmStats(); lib.nsparse_get_spgemm_stats(C.byref(st));"
        "print('RESULT', lib.nsparse_last_error(), st.nnz_c, c.nnz, bool(c.d_col), bool(c.d_rpt))"
    ) % root
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True,
                       env=dict(os.environ, NSPARSE_NO_ABORT="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][-1].split()
    assert line[1:] == ["-40", str(65536 * 32769), "0", "False", "False"], line
    assert "does not fit" in r.stderr
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True,
                       env={k: v for k, v in os.environ.items() if k != "NSPARSE_NO_ABORT"})
    assert r.returncode != 0 and "does not fit" in r.stderr


@pytest.mark.parametrize("prec", ["d", "s"])
def test_heavy_rows_tiled_and_ranked_kernels_agree(prec, lib_d, lib_s, oracle_d, oracle_s):
    """Rows beyond the LDS hash tables (nnz > 5461) through each of the two heavy-row kernels alone
    -- dense column tiles (NSPARSE_RANKED_DENS=0) and the bitmap-ranked accumulator
    (NSPARSE_RANKED_DENS=-1) -- and through the default mix.  R-MAT scale 14 has rows above the
    ranked tile capacity (10240 / 20480 values), so tiles are cut."""
    lib, orc = (lib_d, oracle_d) if prec == "d" else (lib_s, oracle_s)
    A = synth(lib, 3, 14, 16, 0, seed=0x5EED0022)
    ref = orc.spgemm(A, A)
    assert (ref["row_nz"] > 5461).sum() > 100 and ref["row_nz"].max() > 10240
    # fp32: an entry of these rows sums up to ~10^4 products; the float oracle adds them in float in
    # CSR order and is itself ~1e-5 from the exact sum.  The reference's 1e-6 rule is applied against
    # the fp64-accumulated oracle instead (structure from either: identical).
    ref_s = oracle_fp64_accumulated(oracle_d, A) if prec == "s" else None

    def check(got):
        if prec == "d":
            assert_parity(orc, got, ref)
            return
        assert np.array_equal(got["rpt"], ref["rpt"]) and np.array_equal(got["col"], ref["col"])
        assert orc.check_spgemm(got, ref_s) == 0, "fp32 values outside 1e-6 of the fp64-accumulated oracle"

    got, st = spgemm(lib, A)
    assert st.num_bin_size[5] == (ref["row_nz"] > 5461).sum()
    check(got)
    for dens in ("0", "-1"):
        g, s = spgemm_subprocess(A, {"NSPARSE_RANKED_DENS": dens}, prec=prec)
        assert s["num"][5] == st.num_bin_size[5]
        check(g)


def test_window_wider_than_the_bitmap(lib_d, oracle_d):
    """3 M columns: symbolic bin 10 -- the cursor kernel over three 2^20-column tiles, or with
    NSPARSE_SYM_CURSOR=0 the bit window in three pieces --, the numeric phase takes the ranked kernel
    (a dense tiling would need 245 tiles)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(77)
    m, k, n = 48, 3000, 3_000_000
    a = sp.random(m, k, density=120 / k, format="csr", random_state=rng, dtype=np.float64)
    b = sp.random(k, n, density=220 / n, format="csr", random_state=rng, dtype=np.float64)
    a.sort_indices()
    b.sort_indices()
    A = dict(M=m, N=k, rpt=a.indptr.astype(np.int32), col=a.indices.astype(np.int32), val=a.data)
    B = dict(M=k, N=n, rpt=b.indptr.astype(np.int32), col=b.indices.astype(np.int32), val=b.data)
    ref = oracle_d.spgemm(A, B)
    assert ref["row_nz"].min() > 8192
    got, st = spgemm(lib_d, A, B)
    assert st.sym_bin_size[10] == m and st.sym_fail_rows == 0 and st.num_bin_size[5] == m
    assert_parity(oracle_d, got, ref)
    # the same rows without the symbolic cursor kernel: bit window in three pieces
    got2, st2 = spgemm_subprocess(A, {"NSPARSE_SYM_CURSOR": "0"}, B=B)
    assert st2["sym"][10] == m
    assert_parity(oracle_d, got2, ref)


def test_wide_windows_take_the_window_bins(lib_d, oracle_d):
    """20 x 20 cross-section: the window of a C row is ~5 K columns for 375 non-zeros.  Symbolic bin 7,
    numeric bins 7 / 8 through the products rule (span > 8 nnz but <= 2 products); LDS sized by the bin."""
    A = synth(lib_d, 0, 20, 20, 6, seed=3)
    ref = oracle_d.spgemm(A, A)
    got, st = spgemm(lib_d, A, numeric_again=True)
    assert_parity(oracle_d, got, ref)
    assert st.sym_bin_size[7] > 0 and st.num_bin_size[8] > 0
    assert st.num_bin_size[7] + st.num_bin_size[8] > 0.8 * A["M"]
    np.testing.assert_allclose(got["val_again"], got["val"], rtol=1e-12)  # numeric-only re-run, MODE 2
    assert np.array_equal(got["col_again"], got["col"])


@pytest.mark.parametrize("prec", ["d", "s"])
def test_twin_rows_take_their_leaders_structure(prec, lib_d, lib_s, oracle_d, oracle_s):
    """Rows of A with the column pattern of another row (different values) are left out of the symbolic
    bins, wherever they are: the rows of the runs below are shuffled.  Classes of 1 .. 150 rows, patterns
    of 1 .. 1500 entries (tiny, hash and window bins, a deferred long row), look-alikes that differ in
    one entry, empty rows, and NSPARSE_TWINS=0 for the same answer; the numeric-only re-run must keep
    working on the shared structure."""
    lib, orc = (lib_d, oracle_d) if prec == "d" else (lib_s, oracle_s)
    rng = np.random.default_rng(2024)
    k = 6000
    pats = []
    for run, ln in [(3, 30), (150, 12), (1, 7), (5, 1), (4, 400), (2, 1500), (3, 0), (70, 64), (6, 3)]:
        cols = np.sort(rng.choice(k, size=ln, replace=False)).astype(np.int32)
        pats += [cols] * run
        if ln > 1:  # a look-alike: same length, last entry differs
            alt = cols.copy()
            alt[-1] = alt[-1] + 1 if alt[-1] + 1 < k and (ln < 2 or alt[-1] + 1 != alt[-2]) else alt[-1]
            pats.append(alt)
    m = len(pats)
    # half of the rows keep their neighbours, half go anywhere
    order = np.concatenate([np.arange(m // 2), m // 2 + rng.permutation(m - m // 2)])
    pats = [pats[i] for i in order]
    rpt = np.zeros(m + 1, dtype=np.int32)
    rpt[1:] = np.cumsum([len(p) for p in pats])
    col = np.concatenate(pats).astype(np.int32)
    val = rng.uniform(0.5, 1.5, size=len(col))
    A = dict(M=m, N=k, rpt=rpt, col=col, val=val.astype(lib.real))
    import scipy.sparse as sp
    b = sp.random(k, 9000, density=8 / 9000, format="csr", random_state=rng, dtype=np.float64)
    b.sort_indices()
    B = dict(M=k, N=9000, rpt=b.indptr.astype(np.int32), col=b.indices.astype(np.int32),
             val=(b.data + 0.5).astype(lib.real))
    ref = orc.spgemm(A, B)
    got, st = spgemm(lib, A, B, numeric_again=True)
    tw = twin_rows(A, row_windows(A, B)[0])
    assert st.twin_rows == int(tw.sum()) and st.twin_rows > 200
    assert sum(st.sym_bin_size) + st.twin_rows == m and sum(st.num_bin_size) == m
    assert_parity(orc, got, ref)
    assert np.array_equal(got["col_again"], got["col"])
    np.testing.assert_allclose(got["val_again"], got["val"], rtol=1e-9 if prec == "d" else 2e-6)
    got0, st0 = spgemm_subprocess(A, {"NSPARSE_TWINS": "0"}, prec=prec, B=B)
    assert sum(st0["sym"]) == m
    assert np.array_equal(got0["rpt"], got["rpt"]) and np.array_equal(got0["col"], got["col"])


def test_no_rows_and_no_columns(lib_d):
    """M = 0 (the empty block of a row-partitioned run), K = 0 and an all-empty B: zero-size grids are
    not launchable, the call must return an empty C without touching the device kernels."""
    z = lambda m, n: dict(M=m, N=n, rpt=np.zeros(m + 1, np.int32), col=np.zeros(0, np.int32), val=np.zeros(0))
    B = synth(lib_d, 0, 3, 3, 4, seed=1)
    got, st = spgemm(lib_d, z(0, B["M"]), B)
    assert got["M"] == 0 and got["N"] == B["N"] and got["nnz"] == 0 and got["rpt"].tolist() == [0]
    got, st = spgemm(lib_d, z(5, 0), z(0, 7))
    assert got["M"] == 5 and got["N"] == 7 and got["nnz"] == 0 and not got["rpt"].any()
    got, st = spgemm(lib_d, B, z(B["N"], 9))
    assert got["nnz"] == 0 and got["M"] == B["M"] and got["flop"] == 0


def test_chained_product_reads_the_hint_of_c(lib_d, oracle_d):
    """D = (A A) A with the C of the first call passed on as it is (device arrays only): the library
    reads nnz_max of its inputs as a hint, so spgemm_kernel_hash must set it on its output; a garbage
    hint on a caller-built sfCSR must be treated as unknown.  sf_csr2amb on C reads the same field."""
    A = synth(lib_d, 4, 30000, 95000, 0, seed=11)
    a = lib_d.csr_from_numpy(A["rpt"], A["col"], A["val"], A["N"])
    lib_d.csr_memcpy(C.byref(a))
    c, d = ns.sfCSR(), ns.sfCSR()
    lib_d.spgemm_kernel_hash(C.byref(a), C.byref(a), C.byref(c))
    ref_c = oracle_d.spgemm(A, A)
    assert c.nnz == ref_c["nnz"] and c.nnz_max == int(ref_c["row_nz"].max())
    lib_d.spgemm_kernel_hash(C.byref(c), C.byref(a), C.byref(d))
    lib_d.csr_memcpyDtH(C.byref(d))
    got = lib_d.csr_host_to_numpy(d)
    lib_d.release_cpu_csr(d)
    ref_d = oracle_d.spgemm(dict(ref_c, M=A["M"], N=A["N"]), A)
    assert_parity(oracle_d, got, ref_d)
    # the same product with a nonsense hint on both inputs
    lib_d.release_csr(d)
    c.nnz_max, a.nnz_max = -7, 2_000_000_000
    lib_d.spgemm_kernel_hash(C.byref(c), C.byref(a), C.byref(d))
    lib_d.csr_memcpyDtH(C.byref(d))
    got2 = lib_d.csr_host_to_numpy(d)
    lib_d.release_cpu_csr(d)
    assert np.array_equal(got2["rpt"], got["rpt"]) and np.array_equal(got2["col"], got["col"])
    # AMB conversion of C (auto plan) with the hint the product wrote
    c.nnz_max = int(ref_c["row_nz"].max())
    w = 8
    d_x = lib_d.dmalloc((A["N"] + 20) * w)
    x = np.random.default_rng(0).random(A["N"] + 20)
    lib_d.h2d(d_x, x)
    plan, amb = ns.sfPlan(), ns.sfAMB()
    lib_d.init_plan(C.byref(plan))
    lib_d.sf_csr2amb(C.byref(amb), C.byref(c), d_x, C.byref(plan))
    d_y = lib_d.dmalloc((A["M"] + 64) * w)
    lib_d.sf_spmv_amb(d_y, C.byref(amb), d_x, C.byref(plan))
    y = lib_d.d2h(d_y, (A["M"],), np.float64)
    assert oracle_d.ans_check(oracle_d.csr_spmv(ref_c["rpt"], ref_c["col"], ref_c["val"], x[:A["N"]]), y) == 0
    arr = lib_d.amb_to_numpy(amb)
    ora = oracle_d.csr2amb(dict(ref_c, M=A["M"], N=A["N"]), int(plan.seg_size), int(plan.block_size), 64)
    for k in ("cs", "cl", "sellcs_col", "s_write_permutation", "write_permutation"):
        assert np.array_equal(arr[k], getattr(ora, k)), k
    lib_d.release_amb(amb)
    lib_d.dfree(d_x)
    lib_d.dfree(d_y)
    for m in (a, c, d):
        lib_d.release_csr(m)


def test_concurrent_callers_serialise(lib_d, oracle_d):
    """Two host threads in spgemm_kernel_hash at once (ctypes releases the GIL): the entry points take
    one process-wide lock, so the calls run one after the other instead of sharing the device-side
    counters of a call in flight (upstream is not re-entrant either)."""
    import threading
    mats = [synth(lib_d, 0, 6, 6, 30, seed=5), synth(lib_d, 4, 80000, 250000, 0, seed=6)]
    refs = [oracle_d.spgemm(A, A) for A in mats]
    errs = []

    def work(i):
        try:
            for _ in range(6):
                got, _ = spgemm(lib_d, mats[i])
                assert np.array_equal(got["rpt"], refs[i]["rpt"]) and np.array_equal(got["col"], refs[i]["col"])
                assert oracle_d.check_spgemm(got, refs[i]) == 0
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs


@pytest.mark.parametrize("lean", ["0", "3", "7", "11", "15"])  # bits 2 / 3: branch-free retry rounds / pipelined walk (template forms)
def test_hash_bins_both_kernel_families(lean, lib_d, oracle_d):
    """NSPARSE_TB_LEAN=3 / 0: the hash bins 1-4 of both phases through the lean kernels of round 4 (lean.h: owner-array
    product walk, 24-bit multiplicative hash, in-register sort of one-wavefront rows) and through round 3's k_sym_tb /
    k_num_tb, with the window bins off so that all rows hash: stencil, R-MAT and web-graph rows.  Both ship (one is
    the default, the other the switch), so both must be right."""
    for kind, p in ((1, (20, 20, 20)), (3, (12, 8, 0)), (4, (30000, 95000, 0))):
        A = synth(lib_d, kind, *p, seed=0x5EED0022)
        ref = oracle_d.spgemm(A, A)
        got, st = spgemm_subprocess(A, {"NSPARSE_TB_LEAN": lean, "NSPARSE_DENSE": "0"})
        assert sum(st["sym"][6:]) == 0 and sum(st["num"][6:]) == 0
        assert_parity(oracle_d, got, ref)


@pytest.mark.parametrize("prec", ["d", "s"])
def test_node_block_kernel_rows_longer_than_a_batch(prec, lib_d, lib_s, oracle_d, oracle_s):
    """The node-block numeric kernel (block.h) parks 96 A entries per batch: a chain of 3-dof nodes coupled
    to their 20 neighbours on either side has 123 entries per row (two batches), twin rows (groups of 3),
    runs of 3 twin B rows, and a 5-dof variant has groups cut at 3 + 2 and runs that end inside a batch (38 runs
    of 3 chunks: more task records than one stretch holds).  2 dof and 30 neighbours: 48 runs per batch, i.e. two
    passes; 4 dof and 22 neighbours: 180 entries per row, 48 runs of 3 chunks per batch -- two passes of two
    stretches each, the second pass starting inside a stretch.  The last two also with the numbering shuffled
    inside bands of four nodes (keyed runs, cut at the wavefront).
    Also a numeric-only re-run (MODE 2 of the first kernel on the same structure)."""
    import scipy.sparse as sp
    lib, orc = (lib_d, oracle_d) if prec == "d" else (lib_s, oracle_s)
    rng = np.random.default_rng(31)
    for dof, half, shuffled in ((3, 20, False), (5, 9, False), (2, 30, False), (4, 22, False), (2, 30, True), (4, 22, True)):
        nodes = 500
        band = sp.diags([np.ones(nodes - abs(k)) for k in range(-half, half + 1)], range(-half, half + 1), format="csr")
        a = sp.kron(band, np.ones((dof, dof)), format="csr")
        if shuffled:
            perm = np.arange(a.shape[0])
            for s0 in range(0, len(perm), 4 * dof):
                perm[s0:s0 + 4 * dof] = s0 + rng.permutation(min(4 * dof, len(perm) - s0))
            a = a[perm][:, perm].tocsr()
        a.data = rng.uniform(0.5, 1.5, a.nnz)
        a.sort_indices()
        A = dict(M=a.shape[0], N=a.shape[1], rpt=a.indptr.astype(np.int32), col=a.indices.astype(np.int32),
                 val=a.data.astype(lib.real))
        ref = oracle_fp64_accumulated(oracle_d, A) if prec == "s" else orc.spgemm(A, A)
        got, st = spgemm(lib, A, numeric_again=True)
        assert int(np.diff(A["rpt"]).max()) == dof * (2 * half + 1) and st.twin_rows > 0.4 * A["M"]
        assert sum(list(st.num_bin_size)[6:10]) == A["M"], "expected every row in a window bin"
        assert np.array_equal(got["rpt"], ref["rpt"]) and np.array_equal(got["col"], ref["col"])
        assert orc.check_spgemm(got, ref) == 0
        assert np.array_equal(got["col_again"], got["col"])
        np.testing.assert_allclose(got["val_again"], got["val"], rtol=1e-9 if prec == "d" else 2e-6)


def test_fused_tails_match_the_kernel_chains(lib_d, oracle_d):
    """Matrices of up to 256 K rows run the helper chains behind the big kernels as one launch each
    (csrc/spgemm/fused.h); NSPARSE_FUSED=0 keeps the chains.  Same C, same bins, for an FEM brick (twins,
    node-block groups, window bins) and a power law (tiny / hash bins, deferred long rows)."""
    for kind, p in ((0, (7, 6, 21)), (2, (50000, 180000, 0))):
        A = synth(lib_d, kind, *p, seed=77)
        ref = oracle_d.spgemm(A, A)
        got, st = spgemm(lib_d, A)
        assert_parity(oracle_d, got, ref)
        got0, st0 = spgemm_subprocess(A, {"NSPARSE_FUSED": "0"})
        assert np.array_equal(got0["rpt"], got["rpt"]) and np.array_equal(got0["col"], got["col"])
        assert list(st.sym_bin_size) == st0["sym"] and list(st.num_bin_size) == st0["num"]


@pytest.mark.parametrize("m", [1023, 1024, 262143, 262144])
def test_fused_tails_at_their_size_limits(m, lib_d, oracle_d):
    """The fused tails take M + 1 <= 256 * 1024 scan entries: M = 262143 is the last size they run, 262144 the
    first for the chains; 1023 / 1024 rows put the scan tail (entry M) in the first / a second workgroup.
    Bidiagonal-plus-random rows: every bin offset and C.rpt entry is checked against the oracle."""
    rng = np.random.default_rng(m)
    ln = rng.integers(0, 4, size=m)
    ln[rng.integers(0, m, size=5)] = 40
    rpt = np.zeros(m + 1, dtype=np.int32)
    rpt[1:] = np.cumsum(ln)
    col = np.concatenate([np.sort(rng.choice(m, size=k, replace=False)) for k in ln]).astype(np.int32)
    A = dict(M=m, N=m, rpt=rpt, col=col, val=rng.uniform(0.5, 1.5, size=len(col)))
    ref = oracle_d.spgemm(A, A)
    got, st = spgemm(lib_d, A)
    assert_parity(oracle_d, got, ref)
    assert sum(st.num_bin_size) == m and sum(st.sym_bin_size) + st.twin_rows == m


@pytest.mark.parametrize("prec", ["d", "s"])
def test_keyed_runs_of_twin_b_rows(prec, lib_d, lib_s, oracle_d, oracle_s):
    """C = A * A on a 3-dof brick whose numbering is shuffled inside bands (synth kind 5): twin rows are not
    neighbours, so the node-block kernel builds its runs of B rows from the pattern leaders of A's rows
    (two copies of one matrix: k_b_info compares the structures).  Against the oracle, against
    NSPARSE_KEYED=0, and with a B of the same shape and nnz but ANOTHER structure (columns of some rows
    moved): the comparison must notice and the product must still be right."""
    lib, orc = (lib_d, oracle_d) if prec == "d" else (lib_s, oracle_s)
    A = synth(lib, 5, 6, 6, 24, seed=0x5EED0022)
    got, st = spgemm(lib, A)
    assert st.twin_rows > A["M"] // 2
    assert_parity(orc, got, orc.spgemm(A, A))
    got0, _ = spgemm_subprocess(A, {"NSPARSE_KEYED": "0"}, prec=prec)
    assert np.array_equal(got0["rpt"], got["rpt"]) and np.array_equal(got0["col"], got["col"])
    np.testing.assert_allclose(got0["val"], got["val"], rtol=1e-9 if prec == "d" else 2e-6)
    # same shape, same nnz, other structure: last column of every 7th row moved to a free place
    B = dict(A, col=A["col"].copy())
    rpt, n = A["rpt"], A["N"]
    moved = 0
    for r in range(0, A["M"], 7):
        seg = B["col"][rpt[r]:rpt[r + 1]]
        if len(seg) and seg[-1] + 1 < n:
            seg[-1] += 1
            moved += 1
    assert moved > 100
    got, _ = spgemm(lib, A, B)
    assert_parity(orc, got, orc.spgemm(A, B))


_MASK_SCRIPT = r"""
import ctypes as C, json, sys
sys.path.insert(0, "tests")
import numpy as np
import nsparse_amd as ns
from gpu_util import spgemm, synth, variant_env
lib = ns.load("d")
A = synth(lib, 0, 9, 9, 800, seed=5)          # 194,400 rows: the fused tails want 190 co-resident workgroups
got, st = spgemm(lib, A)
co, fb = C.c_int(), C.c_int()
ok = lib.nsparse_fused_state(C.byref(co), C.byref(fb))
np.savez(sys.argv[1], rpt=got["rpt"], col=got["col"], val=got["val"])
print(json.dumps(dict(coresident=co.value, fallbacks=fb.value, fused_ok=ok, err=lib.nsparse_last_error())))
"""


def test_fused_tails_under_a_cu_mask(tmp_path, oracle_d):
    """A quarter of the CUs (HSA_CU_MASK): the census behind the fused tails must see fewer co-resident
    workgroups than the grid barrier of a 194 K-row matrix needs, and the call must take the kernel chains --
    same C, no trap.  With NSPARSE_FUSED_FORCE=1 (census skipped) the barrier really times out: the call is
    repeated with the chains, the context stops fusing, the process stays usable."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    lib = ns.load("d")
    A = synth(lib, 0, 9, 9, 800, seed=5)
    ref = oracle_d.spgemm_omp(A, A)
    outs = {}
    for tag, extra in (("plain", {}), ("mask", {"HSA_CU_MASK": "0:0-31"}),
                       ("mask_forced", {"HSA_CU_MASK": "0:0-31", "NSPARSE_FUSED_FORCE": "1"})):
        npz = str(tmp_path / f"{tag}.npz")
        r = subprocess.run([sys.executable, "-c", _MASK_SCRIPT, npz], cwd=ROOT, env=dict(os.environ, **variant_env(extra)),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (tag, r.stderr[-1500:])
        info = json.loads(r.stdout.strip().splitlines()[-1])
        z = np.load(npz)
        assert np.array_equal(z["rpt"], ref["rpt"]) and np.array_equal(z["col"], ref["col"]), tag
        assert oracle_d.check_spgemm(dict(M=A["M"], nnz=int(z["rpt"][-1]), rpt=z["rpt"], col=z["col"], val=z["val"]), ref) == 0
        assert info["err"] == 0
        outs[tag] = info
    print("[cu mask]", outs)
    assert outs["plain"]["coresident"] >= 190 and outs["plain"]["fallbacks"] == 0 and outs["plain"]["fused_ok"] == 1
    if outs["mask"]["coresident"] >= 190:
        pytest.skip("HSA_CU_MASK has no effect on this box: the census sees every CU")
    assert outs["mask"]["fallbacks"] == 0 and outs["mask"]["fused_ok"] == 1      # chains by the census, no time-out
    assert outs["mask_forced"]["fallbacks"] == 1 and outs["mask_forced"]["fused_ok"] == 0


@pytest.mark.parametrize("kind,dims", [(0, (5, 5, 12)), (5, (6, 6, 12))])
def test_non_finite_values_stay_in_their_columns(kind, dims, oracle_d):
    """An Inf in A must reach exactly the entries of C the reference's algorithm gives it to (the oracle's
    row-by-row sums).  The node-block kernel forms a0 v0 + a1 v1 + a2 v2 for runs of up to three B rows; for a
    shorter run the unused a's used to be the NEXT parked entries of A with v = 0, and Inf * 0 = NaN leaked into
    products of unrelated entries (round-2 advisor finding)."""
    lib = ns.load("d")
    A = synth(lib, kind, *dims, seed=21)
    val = A["val"].copy()
    rng = np.random.default_rng(2)
    hit = rng.choice(len(val), 5, replace=False)
    val[hit] = np.inf
    A = dict(A, val=val)
    got, st = spgemm(lib, A)
    ref = oracle_d.spgemm(A, A)
    assert np.array_equal(got["rpt"], ref["rpt"]) and np.array_equal(got["col"], ref["col"])
    bad_ref = ~np.isfinite(ref["val"])
    bad_got = ~np.isfinite(got["val"])
    assert bad_ref.sum() > 0 and np.array_equal(bad_got, bad_ref), (int(bad_got.sum()), int(bad_ref.sum()))
    ok = ~bad_ref
    np.testing.assert_allclose(got["val"][ok], ref["val"][ok], rtol=1e-9)


@pytest.mark.parametrize("lean", ["0", "3", "15"])
def test_big_table_bins_on_clustered_columns(lean, oracle_d):
    """Rows of the two big-table numeric bins (683 .. 5461 non-zeros) through both kernel families (NSPARSE_TB_LEAN).
    B is a diagonal matrix, so a row of C has exactly the columns of its row of A and every A entry reaches a
    one-entry row of B (the DIRECT rounds of the lean walk): a uniform row, rows with 100 / 400 consecutive columns
    (probe clusters of a multiplicative hash), and a cluster of 3,000 consecutive columns plus one far outlier."""
    lib = ns.load("d")
    N = 1 << 20
    rng = np.random.default_rng(17)
    rows = []
    rows.append(np.sort(rng.choice(N, 3000, replace=False)))                                        # uniform
    rows.append(np.unique(np.concatenate([rng.choice(N, 2500, replace=False), 40960 + np.arange(100)])))   # 128-path
    rows.append(np.unique(np.concatenate([rng.choice(N, 4000, replace=False), 614400 + np.arange(400)])))  # 512-path
    rows.append(np.concatenate([np.arange(3000), [N - 1]]))                                          # fallback
    rows.append(np.sort(rng.choice(N, 900, replace=False)))                                          # 4096-slot bin
    rows.append(np.concatenate([np.arange(5, 2005), [N - 7]]))                                       # fallback, other bin
    rpt = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    A = dict(M=len(rows), N=N, rpt=rpt, col=np.concatenate(rows).astype(np.int32), val=rng.random(int(rpt[-1])) + 0.5)
    B = dict(M=N, N=N, rpt=np.arange(N + 1, dtype=np.int32), col=np.arange(N, dtype=np.int32), val=rng.random(N) + 0.5)
    got, st = spgemm_subprocess(A, {"NSPARSE_TB_LEAN": lean}, "d", B=B)
    ref = oracle_d.spgemm(A, B)
    assert np.array_equal(got["rpt"], ref["rpt"]) and np.array_equal(got["col"], ref["col"])
    assert oracle_d.check_spgemm(got, ref) == 0
    assert st["num"][3] + st["num"][4] == len(rows), st["num"][:8]


@pytest.mark.parametrize("lean", ["0", "3", "15"])
def test_one_wavefront_bin(lean, oracle_d):
    """The rows of numeric bin 1 (17 .. 170 non-zeros) and symbolic bin 1 through both kernel families
    (NSPARSE_TB_LEAN): a 27-point stencil (every row in those bins; the lean walk spreads the 189 chunks of a row
    over three rounds through its owner array, the sort of <= 128 keys runs in registers), then short rows of A
    that meet a B with 5 % hub rows of 120 entries (owner windows that a long row straddles, partial chunks)."""
    lib = ns.load("d")
    A = synth(lib, 1, 70, 70, 12, seed=5)  # 58,800 rows, windows of 20 K columns: beyond the dense-window bins
    rng = np.random.default_rng(3)
    got, st = spgemm_subprocess(A, {"NSPARSE_TB_LEAN": lean}, "d")
    ref = oracle_d.spgemm(A, A)
    assert np.array_equal(got["rpt"], ref["rpt"]) and np.array_equal(got["col"], ref["col"])
    assert oracle_d.check_spgemm(got, ref) == 0
    assert st["num"][1] > 0.9 * A["M"], st["num"][:8]
    # mixed rows (one B row far longer than the others) and short ones side by side
    m = 5000
    lens = rng.integers(1, 6, m)
    cols = [np.sort(rng.choice(m, k, replace=False)) for k in lens]
    rptA = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    A2 = dict(M=m, N=m, rpt=rptA, col=np.concatenate(cols).astype(np.int32), val=rng.random(int(rptA[-1])) + 0.5)
    lensB = np.where(rng.random(m) < 0.05, 120, rng.integers(2, 12, m))
    colsB = [np.sort(rng.choice(m, k, replace=False)) for k in lensB]
    rptB = np.concatenate([[0], np.cumsum(lensB)]).astype(np.int32)
    B2 = dict(M=m, N=m, rpt=rptB, col=np.concatenate(colsB).astype(np.int32), val=rng.random(int(rptB[-1])) + 0.5)
    got, st = spgemm_subprocess(A2, {"NSPARSE_TB_LEAN": lean}, "d", B=B2)
    ref = oracle_d.spgemm(A2, B2)
    assert np.array_equal(got["rpt"], ref["rpt"]) and np.array_equal(got["col"], ref["col"])
    assert oracle_d.check_spgemm(got, ref) == 0
    assert st["num"][1] > 0, st["num"][:8]


@pytest.mark.parametrize("kind,dims,twins", [(1, (52, 52, 52), False), (0, (12, 12, 310), True), (5, (12, 12, 310), True)])
def test_twin_sample_of_big_matrices(kind, dims, twins, lib_d, oracle_d):
    """Matrices of 131,072 rows and more: k_b_info looks at a sample of the rows of A (the first 64 of every 1024) and
    k_row_products uses the pattern map only if the sample holds a pattern twice (setup.h: TwinSample) -- a scalar
    stencil skips the map (one returning atomic per row for nothing), finite-element matrices keep it whether their
    twin rows are neighbours (kind 0) or scattered (kind 5: seen by chance, a sixteenth of the rows is sampled).  The
    count is exact on the sample: gpu_util.twin_rows mirrors the rule."""
    A = synth(lib_d, kind, *dims, seed=11)
    assert A["M"] >= 131072
    ref = oracle_d.spgemm(A, A)
    got, st = spgemm(lib_d, A)
    assert_parity(oracle_d, got, ref)
    rp, _, _ = oracle_d.nprod(A["rpt"], A["col"], A["rpt"])
    tw = twin_rows(A, rp)
    assert st.twin_rows == int(tw.sum())
    assert (st.twin_rows > A["M"] // 2) if twins else st.twin_rows == 0
    # the switch: always probe
    got0, st0 = spgemm_subprocess(A, {"NSPARSE_TWIN_SAMPLE": "0"}, "d")
    assert np.array_equal(got0["rpt"], ref["rpt"]) and np.array_equal(got0["col"], ref["col"])


def test_keyed_runs_for_a_general_b(oracle_d):
    """A * B with B != A, both finite-element matrices whose twin rows are scattered (kind 5, different
    renumberings): the node-block kernel gets the pattern leaders of B's rows from k_b_twins (setup.h) instead of
    from A's own map; and a row block of A times the whole A (what a rank of the partitioned product computes)."""
    lib = ns.load("d")
    A = synth(lib, 5, 6, 6, 40, seed=11)
    B = synth(lib, 5, 6, 6, 40, seed=12)
    assert A["M"] == B["M"] and not np.array_equal(A["col"][:200], B["col"][:200])
    got, st = spgemm(lib, A, B)
    ref = oracle_d.spgemm(A, B)
    assert np.array_equal(got["rpt"], ref["rpt"]) and np.array_equal(got["col"], ref["col"])
    assert oracle_d.check_spgemm(got, ref) == 0
    from nsparse_amd.dist import csr_row_block
    blk = csr_row_block(A, 1024, 3000)
    got, st = spgemm(lib, dict(blk, N=A["M"]), A)
    ref = oracle_d.spgemm(dict(blk, N=A["M"]), A)
    assert np.array_equal(got["rpt"], ref["rpt"]) and np.array_equal(got["col"], ref["col"])
    assert oracle_d.check_spgemm(got, ref) == 0
    assert st.twin_rows > 0


@pytest.mark.parametrize("prec", ["d", "s"])
def test_numeric_rerun_of_ranked_window_rows(prec, lib_d, lib_s):
    """nsparse_spgemm_hash_numeric on power-law rows: without the bitmaps of the symbolic phase the
    re-run bins MORE rows into the ranked window (numeric bin 9), whose kernel then rebuilds its bitmap from C.col
    (k_num_block<128, 65536, MODE 2>) -- an instantiation no other test reached (round 5: kernel coverage on the CPU
    emulation, profiles/r05_emu_kernel_coverage_d.txt).  Same columns, same values as the full call."""
    lib = lib_d if prec == "d" else lib_s
    for p in ((10, 8, 0), (12, 8, 0)):
        A = synth(lib, 3, *p, seed=3)
        A = dict(A, val=A["val"].astype(lib.real))
        got, st = spgemm(lib, A, numeric_again=True)
        st2 = ns.SpgemmStats()
        lib.nsparse_get_spgemm_stats(C.byref(st2))
        assert st2.num_bin_size[9] > st.num_bin_size[9] > 0
        assert np.array_equal(got["col_again"], got["col"])
        np.testing.assert_allclose(got["val_again"], got["val"], rtol=1e-9 if prec == "d" else 2e-6)
