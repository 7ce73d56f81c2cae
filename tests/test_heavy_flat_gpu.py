"""Stateless heavy-row tiles (nsparse_amd/csrc/spgemm/heavy_flat.h, round 6): the dense column tiles of the heavy numeric
bin as flat product walks over a panel table of B, against the oracle.  Opt-in (NSPARSE_HEAVY_FLAT=1 in the
-DNSPARSE_EXPERIMENTS variant library, nsparse_amd/lib_exp) until it has been timed on the device, so every case runs in
a fresh interpreter on that library.  Replaces the same reference code as k_num_tiled:
cuda-c/src/kernel/kernel_spgemm_hash_d.cu:929-1027 (calculate_value_col_bin_each_gl)."""
import numpy as np
import pytest

from gpu_util import experiments_lib_dir, oracle_fp64_accumulated, spgemm_subprocess, synth

pytestmark = pytest.mark.gpu


def assert_parity(orc, got, ref):
    assert got["nnz"] == ref["nnz"]
    assert np.array_equal(got["rpt"], ref["rpt"]), "C.rpt differs"
    assert np.array_equal(got["col"], ref["col"]), "C.col differs"
    assert orc.check_spgemm(got, ref) == 0, "values outside the reference tolerance"


def flat(A, B=None, prec="d", numeric_again=False, **env):
    """NSPARSE_HEAVY_FLAT=7: dense tiles (bit 0), list-driven ranked tiles (bit 1) and the symbolic twin for windows wider
    than 2^20 columns (bit 2), all stateless."""
    d = experiments_lib_dir()
    if d is None:
        pytest.skip("no experiments variant library beside the one in use (__graft_entry__.build() makes lib_exp)")
    got, st = spgemm_subprocess(A, dict(dict(NSPARSE_LIB_DIR=d, NSPARSE_HEAVY_FLAT="7"), **env), prec=prec, B=B,
                                numeric_again=numeric_again)
    assert st["build"].split()[-1] == "experiments", st["build"]
    return got, st


def csr(m):
    m = m.tocsr()
    m.sum_duplicates()
    m.sort_indices()
    return dict(M=m.shape[0], N=m.shape[1], rpt=m.indptr.astype(np.int32), col=m.indices.astype(np.int32),
                val=m.data.astype(np.float64))


@pytest.mark.parametrize("prec", ["d", "s"])
def test_rmat14_heavy_rows(prec, lib_d, lib_s, oracle_d, oracle_s):
    """560 rows beyond the LDS hash tables, up to 13 tiles each; with NSPARSE_RANKED_DENS=0 every one of them takes the
    stateless dense tiles (default: the thin ones stay with the ranked cursor kernel, which shares the row queue rules)."""
    lib, orc = (lib_d, oracle_d) if prec == "d" else (lib_s, oracle_s)
    A = synth(lib, 3, 14, 16, 0, seed=0x5EED0022)
    ref = orc.spgemm(A, A)
    ref_s = oracle_fp64_accumulated(oracle_d, A) if prec == "s" else None
    for dens in ("12", "0"):
        got, st = flat(A, prec=prec, NSPARSE_RANKED_DENS=dens)
        assert st["num"][5] == (ref["row_nz"] > 5461).sum() > 100
        if prec == "d":
            assert_parity(orc, got, ref)
        else:
            assert np.array_equal(got["rpt"], ref["rpt"]) and np.array_equal(got["col"], ref["col"])
            assert orc.check_spgemm(got, ref_s) == 0


def test_hub_rows_of_a_and_empty_tiles(oracle_d):
    """Rows of A with 5000 entries (five batches of the flat walk: two whose extent locations stay in registers, three
    that park them in the heavy bin's slab, the last one partial) over a B that mixes rows of 2-3 entries with rows of
    thousands, whose columns leave panels 2 and 3 of 6 empty (a tile without a product emits nothing and leaves the
    window clean)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(61)
    k, n = 6000, 6 * 12288
    lens = np.where(rng.random(k) < 0.02, rng.integers(1500, 4000, k), rng.integers(0, 4, k))
    allowed = np.concatenate([np.arange(0, 2 * 12288), np.arange(4 * 12288, n)])
    rows, cols = [], []
    for r, ln in enumerate(lens):
        rows.append(np.full(ln, r))
        cols.append(rng.choice(allowed, size=ln, replace=False))
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    B = csr(sp.coo_matrix((rng.random(len(rows)) + 0.5, (rows, cols)), shape=(k, n)))
    ar = np.repeat(np.arange(6), 5000)
    ac = np.concatenate([rng.choice(k, size=5000, replace=False) for _ in range(6)])
    A = csr(sp.coo_matrix((rng.random(len(ar)) + 0.5, (ar, ac)), shape=(6, k)))
    ref = oracle_d.spgemm(A, B)
    assert ref["row_nz"].min() > 5461
    got, st = flat(A, B)
    assert st["num"][5] == 6
    assert_parity(oracle_d, got, ref)


def test_short_rows_outside_the_table(oracle_d):
    """B with 2.5 M rows and 50 column panels: a pointer row per B row would take 500 MB, so only rows of more than 16
    entries get one; the others (here: most of the entries of every A row) are walked whole by every tile of the C row
    and filtered by column."""
    import scipy.sparse as sp
    rng = np.random.default_rng(62)
    k, n, used = 2_500_000, 49 * 12288 + 100, 5 * 12288
    pick = rng.choice(k, size=4000, replace=False)
    lens = np.where(rng.random(4000) < 0.05, rng.integers(800, 3000, 4000), rng.integers(1, 17, 4000))
    rows, cols = [], []
    for r, ln in zip(pick, lens):
        rows.append(np.full(ln, r))
        cols.append(rng.choice(used, size=ln, replace=False))
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    B = csr(sp.coo_matrix((rng.random(len(rows)) + 0.5, (rows, cols)), shape=(k, n)))
    ar = np.repeat(np.arange(4), 1500)
    ac = np.concatenate([rng.choice(pick, size=1500, replace=False) for _ in range(4)])
    A = csr(sp.coo_matrix((rng.random(len(ar)) + 0.5, (ar, ac)), shape=(4, k)))
    ref = oracle_d.spgemm(A, B)
    assert ref["row_nz"].min() > 5461
    got, st = flat(A, B)
    assert st["num"][5] == 4
    assert_parity(oracle_d, got, ref)


def test_window_wider_than_the_bitmap_with_lists(oracle_d):
    """3 M columns: the symbolic phase through k_sym_flat (three tiles of 85 panels per row, column lists written), then
    every heavy row -- thin, with a list -- through k_num_ranked_flat; tiles start at the panel of the next listed column
    and end at panel boundaries.  Also with the cursor kernel in ONE of the two phases (bits 2 / 3 alone)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(77)
    m, k, n = 48, 3000, 3_000_000
    a = sp.random(m, k, density=120 / k, format="csr", random_state=rng, dtype=np.float64)
    b = sp.random(k, n, density=220 / n, format="csr", random_state=rng, dtype=np.float64)
    A, B = csr(a), csr(b)
    ref = oracle_d.spgemm(A, B)
    assert ref["row_nz"].min() > 8192
    got, st = flat(A, B, numeric_again=True)
    assert st["num"][5] == m and st["sym"][10] == m
    assert_parity(oracle_d, got, ref)
    assert np.array_equal(got["col_again"], got["col"])
    np.testing.assert_allclose(got["val_again"], got["val"], rtol=1e-9)
    for bits in ("4", "3"):  # stateless symbolic + cursor numeric, and the other way round: the lists are interchangeable
        g, _ = flat(A, B, NSPARSE_HEAVY_FLAT=bits)
        assert_parity(oracle_d, g, ref)


def test_hub_rows_of_a_on_a_wide_matrix(oracle_d):
    """2.2 M columns, rows of A with 2600 entries: the thin heavy rows go through k_sym_flat (the chain for the batch
    beyond its two register batches) and k_num_ranked_flat (that batch from the slab), several list-driven tiles each."""
    import scipy.sparse as sp
    rng = np.random.default_rng(78)
    m, k, n = 5, 3000, 2_200_000
    ar = np.repeat(np.arange(m), 2600)
    ac = np.concatenate([rng.choice(k, size=2600, replace=False) for _ in range(m)])
    A = csr(sp.coo_matrix((rng.random(len(ar)) + 0.5, (ar, ac)), shape=(m, k)))
    B = csr(sp.random(k, n, density=40 / n, format="csr", random_state=rng, dtype=np.float64))
    ref = oracle_d.spgemm(A, B)
    assert ref["row_nz"].min() > 8192
    got, st = flat(A, B, numeric_again=True)
    assert st["num"][5] == m and st["sym"][10] == m
    assert_parity(oracle_d, got, ref)
    np.testing.assert_allclose(got["val_again"], got["val"], rtol=1e-9)


@pytest.mark.parametrize("dens", ["-1", "12"])
def test_numeric_rerun_has_a_list_for_every_heavy_row(dens, lib_d, oracle_d):
    """A numeric-only re-run uses C.col as the list of every row.  With NSPARSE_RANKED_DENS=-1 all 560 heavy rows of
    R-MAT-14 go through the list-driven stateless tiles in the re-run (W = 2^18, CAP = 10240): their dense low-column
    stretch puts more than CAP listed columns into one 12288-column panel, so tiles are cut INSIDE a panel and the walk
    filters by column.  The first run (no lists on a matrix this narrow) takes the cursor kernel for the same rows."""
    A = synth(lib_d, 3, 14, 16, 0, seed=0x5EED0022)
    ref = oracle_d.spgemm(A, A)
    assert ref["row_nz"].max() > 10240
    got, st = flat(A, numeric_again=True, NSPARSE_RANKED_DENS=dens)
    assert_parity(oracle_d, got, ref)
    assert np.array_equal(got["col_again"], got["col"])
    again = dict(got, val=got["val_again"])
    assert oracle_d.check_spgemm(again, ref) == 0


def test_config5_code_paths_through_the_stateless_tiles(lib_d, oracle_d):
    """R-MAT scale 22 at a fifth of config 5's edges (tests/test_configs_gpu.py has the cursor kernels on the same input):
    4 M columns, 343 panels, only the B rows of more than 16 entries in the table, lists from the symbolic cursor kernel,
    thick heavy rows through k_num_flat and thin ones through k_num_ranked_flat."""
    A = synth(lib_d, 3, 22, 0, 1500000, seed=0x5EED0022)
    got, st = flat(A, numeric_again=True)
    assert st["num"][5] > 500
    ref = oracle_d.spgemm_omp(A, A)
    assert np.array_equal(got["rpt"], ref["rpt"]) and np.array_equal(got["col"], ref["col"])
    assert oracle_d.check_spgemm(got, dict(ref, M=A["M"])) == 0
    assert np.array_equal(got["col_again"], got["col"])
    np.testing.assert_allclose(got["val_again"], got["val"], rtol=1e-9)


@pytest.mark.parametrize("seed", range(8))
def test_random_heavy_rows_at_panel_and_batch_edges(seed, oracle_d):
    """Randomised heavy rows whose sizes sit ON the edges of the stateless kernels: matrix widths that are exact multiples
    of the 12288-column panel (and one more), column windows that start and end exactly on panel boundaries, rows of A
    with 1023 .. 2049 and 4100 entries (the 1024-entry batches, the two register batches, the slab batches), narrow
    matrices (dense tiles, every row of B in the table) and matrices wider than 2^20 columns (symbolic twin, lists,
    ranked tiles)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(4000 + seed)
    G = 12288
    n = int([3 * G, 3 * G + 1, 7 * G - 1, 50_000, 1_100_000 + seed, 86 * G, 2_300_000, 5 * G][seed])
    k = 4500
    lo_p = int(rng.integers(0, max(1, n // G - 2)))
    c_lo = lo_p * G if seed % 2 == 0 else int(rng.integers(0, max(1, n - 2 * G)))
    width = n - c_lo if n <= 100_000 else int(rng.choice([2 * G, 85 * G, n - c_lo]))
    c_hi = min(n, c_lo + max(width, 2 * G))
    lens = np.where(rng.random(k) < 0.06, rng.integers(300, 1500, k), rng.integers(0, 5, k))
    lens = np.minimum(lens, c_hi - c_lo)
    rows = np.concatenate([np.full(ln, r) for r, ln in enumerate(lens)])
    cols = np.concatenate([c_lo + rng.choice(c_hi - c_lo, size=ln, replace=False) for ln in lens if ln > 0] or [np.zeros(0, int)])
    B = csr(sp.coo_matrix((rng.random(len(rows)) + 0.5, (rows, cols)), shape=(k, n)))
    alens = [1023, 1024, 2048, 2049, 4100]
    m = 5
    ar = np.concatenate([np.full(a, i) for i, a in enumerate(alens)])
    ac = np.concatenate([rng.choice(k, size=a, replace=False) for a in alens])
    A = csr(sp.coo_matrix((rng.random(len(ar)) + 0.5, (ar, ac)), shape=(m, k)))
    ref = oracle_d.spgemm(A, B)
    heavy = int((ref["row_nz"] > 5461).sum())
    assert heavy >= 3, ref["row_nz"]
    got, st = flat(A, B, numeric_again=True)
    assert st["num"][5] == heavy
    assert_parity(oracle_d, got, ref)
    assert np.array_equal(got["col_again"], got["col"])
    np.testing.assert_allclose(got["val_again"], got["val"], rtol=1e-9)
