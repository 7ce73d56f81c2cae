"""Inputs at the edges of what the loader can produce and the bins can take (round 5; first run on the CPU emulation,
tests/emu): duplicate column entries inside a row -- init_csr_matrix_from_file merges nothing (nsparse.cu:14-144,
SURVEY 8a2) -- sorted-with-repeats and unsorted; 40 M columns (beyond the 24 bits the lean hash looks at); a 200 K-entry
hub row on either side of the product; vectors; rows of B that are all empty.  Structure bit for bit, values by the
reference's rule (nsparse.cu:300-353), through both hash kernel families."""
import numpy as np
import pytest
import scipy.sparse as sp

from gpu_util import spgemm_subprocess

pytestmark = pytest.mark.gpu


def _csr(m, n, rows):
    rpt = np.zeros(m + 1, np.int32)
    col, val = [], []
    for i, (c, v) in enumerate(rows):
        rpt[i + 1] = rpt[i] + len(c)
        col += list(c)
        val += list(v)
    return dict(M=m, N=n, rpt=rpt, col=np.array(col, np.int32), val=np.array(val, np.float64))


def _from_scipy(S):
    S = S.tocsr()
    S.sort_indices()
    return dict(M=S.shape[0], N=S.shape[1], rpt=S.indptr.astype(np.int32), col=S.indices.astype(np.int32), val=S.data.copy())


def _dup_rows(rng, m, n, k, sort):
    rows = []
    for _ in range(m):
        c = rng.integers(0, n, size=int(rng.integers(0, k)))
        c = np.concatenate([c, c[: len(c) // 2]])  # every second entry once more
        c = np.sort(c) if sort else rng.permutation(c)
        rows.append((c, rng.random(len(c)) + 0.1))
    return _csr(m, n, rows)


def _cases():
    rng = np.random.default_rng(7)
    yield "duplicates, sorted with repeats", _dup_rows(rng, 300, 300, 12, True), None
    yield "duplicates, unsorted", _dup_rows(rng, 300, 300, 12, False), None
    yield "duplicates in hub-sized rows", _dup_rows(rng, 64, 4000, 900, True), _dup_rows(rng, 4000, 3000, 6, True)
    B = sp.random(2000, 40_000_000, density=3e-7, format="csr", random_state=rng, dtype=np.float64)
    A = sp.random(500, 2000, density=0.02, format="csr", random_state=rng, dtype=np.float64)
    yield "40 M columns", _from_scipy(A), _from_scipy(B)
    n = 250_000
    hub = _csr(3, n, [(np.sort(rng.choice(n, 200_000, replace=False)), rng.random(200_000)), (np.array([5]), [1.0]),
                      (np.array([], int), [])])
    yield "a row of A with 200 K entries", hub, _from_scipy(sp.random(n, 5000, density=2e-4, format="csr", random_state=rng, dtype=np.float64))
    yield "rows of B with 200 K entries", _from_scipy(sp.random(600, 3, density=0.3, format="csr", random_state=rng, dtype=np.float64)), hub
    # a thin heavy row of C (12 K non-zeros over 3 M columns: the ranked kernel) from a row of A with 6,000 entries -- more
    # than the 4 x 1024 cursors the kernel keeps in registers, the rest live in its global slice (heavy_ranked.h)
    nb, wide = 7000, 3_000_000
    Bthin = _csr(nb, wide, [(np.sort(rng.choice(wide, 2, replace=False)), rng.random(2) + 0.1) for _ in range(nb)])
    Along = _csr(4, nb, [(np.sort(rng.choice(nb, 6000, replace=False)), rng.random(6000) + 0.1), (np.array([3, 9]), [1.0, 2.0]),
                         (np.sort(rng.choice(nb, 4500, replace=False)), rng.random(4500) + 0.1), (np.array([], int), [])])
    yield "thin heavy rows from rows of A beyond the register cursors", Along, Bthin
    yield "1 x 1", _csr(1, 1, [(np.array([0]), [2.0])]), None
    col_v = _csr(400, 1, [(np.array([0]), [1.0 + i]) for i in range(400)])
    row_v = _csr(1, 400, [(np.arange(400), np.arange(400) + 1.0)])
    yield "column vector times row vector", col_v, row_v
    yield "row vector times column vector", row_v, col_v
    yield "every reached row of B empty", _csr(50, 60, [(np.array([i % 60]), [1.0]) for i in range(50)]), _csr(60, 10, [(np.array([], int), [])] * 60)


@pytest.mark.parametrize("lean", ["0", "3"])
def test_exotic_inputs(lean, oracle_d):
    for name, A, B in _cases():
        B = B or A
        ref = oracle_d.spgemm(A, B)
        got, st = spgemm_subprocess(A, {"NSPARSE_TB_LEAN": lean}, B=B)
        assert np.array_equal(got["rpt"], ref["rpt"]) and np.array_equal(got["col"], ref["col"]), name
        assert oracle_d.check_spgemm(got, ref) == 0, name
