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


def flat(A, B=None, prec="d", **env):
    d = experiments_lib_dir()
    if d is None:
        pytest.skip("no experiments variant library beside the one in use (__graft_entry__.build() makes lib_exp)")
    got, st = spgemm_subprocess(A, dict(NSPARSE_LIB_DIR=d, NSPARSE_HEAVY_FLAT="1", **env), prec=prec, B=B)
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
    """Rows of A with ~3000 entries (three batches of the flat walk, the last one partial) over a B that mixes rows of
    2-3 entries with rows of thousands, whose columns leave panels 2 and 3 of 6 empty (a tile without a product emits
    nothing and leaves the window clean)."""
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
    ar = np.repeat(np.arange(6), 3000)
    ac = np.concatenate([rng.choice(k, size=3000, replace=False) for _ in range(6)])
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
