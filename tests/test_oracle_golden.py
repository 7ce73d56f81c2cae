"""Pin the CPU oracle: reference fixture (data/test.mtx) + independent scipy vectors.

Not a product test: nothing here touches nsparse_amd.  If these fail, the checker is wrong."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, TEST_MTX, load_golden


def test_loader_test_mtx(oracle_d):
    A = oracle_d.load_mtx(os.path.join(GOLDEN, "test.mtx"))
    for k in ("M", "N", "nnz", "nnz_max"):
        assert A[k] == TEST_MTX[k]
    assert A["rpt"].tolist() == TEST_MTX["rpt"]
    assert A["col"].tolist() == TEST_MTX["col"]
    assert A["val"].tolist() == TEST_MTX["val"]


def test_spmv_test_mtx(oracle_d):
    A = oracle_d.load_mtx(os.path.join(GOLDEN, "test.mtx"))
    y = oracle_d.csr_spmv(A["rpt"], A["col"], A["val"], np.array(TEST_MTX["x"], float))
    assert y.tolist() == TEST_MTX["y"]


def test_spgemm_test_mtx(oracle_d):
    A = oracle_d.load_mtx(os.path.join(GOLDEN, "test.mtx"))
    rp, tot, mx = oracle_d.nprod(A["rpt"], A["col"], A["rpt"])
    assert rp.tolist() == TEST_MTX["row_prod"] and tot == TEST_MTX["n_prod"] and mx == 7
    Cm = oracle_d.spgemm(A, A)
    assert Cm["rpt"].tolist() == TEST_MTX["c_rpt"]
    assert Cm["col"].tolist() == TEST_MTX["c_col"]
    assert Cm["val"].tolist() == TEST_MTX["c_val"]
    # all rows land in bin 0 of both phases with the reference ladder (max n_prod 7 <= 32)
    assert oracle_d.bin_hist_ref(rp, 512, 32).tolist() == [5, 0, 0, 0, 0, 0, 0]
    assert oracle_d.bin_hist_ref(Cm["row_nz"], 256, 16).tolist() == [5, 0, 0, 0, 0, 0, 0]


def test_amb_test_mtx_layout(oracle_d):
    """AMB of test.mtx, seg_size 65536, block_size 1, chunk 32 -- SURVEY.md 8c."""
    A = oracle_d.load_mtx(os.path.join(GOLDEN, "test.mtx"))
    a = oracle_d.csr2amb(A, 65536, 1, 32)
    assert (a.pad_M, a.seg_num, a.c_size, a.nnz) == (32, 1, 1, 96)
    assert a.cs.tolist() == [0] and a.cl.tolist() == [2]
    col = a.sellcs_col.reshape(3, 32)
    val = a.sellcs_val.reshape(3, 32)
    assert col[:, 0].tolist() == [0, 2, 4] and val[:, 0].tolist() == [1, 30, 2]
    assert col[:, 1].tolist() == [0, 2, 4] and val[:, 1].tolist() == [10, 1, 0]
    assert col[:, 2].tolist() == [2, 4, 4] and val[:, 2].tolist() == [2, 50, 0]
    assert col[:, 3].tolist() == [1, 2, 4] and val[:, 3].tolist() == [20, 0, 0]
    assert col[:, 4].tolist() == [3, 2, 4] and val[:, 4].tolist() == [40, 0, 0]
    assert (col[:, 5:] == np.array([[0], [2], [4]])).all() and (val[:, 5:] == 0).all()
    assert a.s_write_permutation.tolist() == [2, 0, 4, 1, 3] + list(range(5, 32))
    assert a.s_write_permutation_offset.tolist() == [0]
    assert a.spmv(np.array(TEST_MTX["x"], float)).tolist() == TEST_MTX["y"]


@pytest.mark.parametrize("name", ["banded2k", "banded_signed1k", "rmat_s10"])
def test_against_scipy_vectors(oracle_d, name):
    g = load_golden(name)
    y = oracle_d.csr_spmv(g["rpt"], g["col"], g["val"], g["x"])
    assert oracle_d.ans_check(g["y"], y) == 0
    np.testing.assert_allclose(y, g["y"], rtol=1e-13)
    rp, tot, _ = oracle_d.nprod(g["rpt"], g["col"], g["rpt"])
    assert np.array_equal(rp, g["row_prod"]) and tot == int(g["row_prod"].sum())
    Cm = oracle_d.spgemm(g, g)
    assert Cm["nnz"] == len(g["c_col"])
    assert np.array_equal(Cm["rpt"], g["c_rpt"]) and np.array_equal(Cm["col"], g["c_col"])
    ref = dict(M=g["M"], nnz=len(g["c_col"]), rpt=g["c_rpt"], col=g["c_col"], val=g["c_val"])
    assert oracle_d.check_spgemm(Cm, ref) == 0
    # OpenMP variant (CPU baseline leg) is the same arithmetic
    Co = oracle_d.spgemm_omp(g, g)
    assert np.array_equal(Co["col"], Cm["col"]) and np.array_equal(Co["val"], Cm["val"])
    # ... and so is the fast all-cores form the bench times (window sweep instead of a sort for narrow rows), at
    # one thread and at all of them
    for nth in (1, 0):
        Ct, best, mean, used = oracle_d.spgemm_omp_timed(g, g, reps=1, threads=nth)
        assert used >= 1 and best > 0 and mean >= best
        assert np.array_equal(Ct["rpt"], Cm["rpt"]) and np.array_equal(Ct["col"], Cm["col"])
        assert np.array_equal(Ct["val"], Cm["val"])


def test_float_oracle(oracle_s):
    g = load_golden("banded2k")
    y = oracle_s.csr_spmv(g["rpt"], g["col"], g["val"], g["x"])
    np.testing.assert_allclose(y, g["y"], rtol=2e-5)
    Cm = oracle_s.spgemm(g, g)
    assert np.array_equal(Cm["col"], g["c_col"])
    np.testing.assert_allclose(Cm["val"], g["c_val"], rtol=2e-5)


@pytest.mark.parametrize("chunk", [32, 64])
@pytest.mark.parametrize("name,segs,blocks", [
    ("banded2k", [65536, 1024, 300], [1, 2, 3, 7, 20]),
    ("rmat_s10", [65536, 256], [1, 4]),
    ("wide_seg", [65536, 4096], [1, 2, 5]),
])
def test_amb_invariants(oracle_d, name, segs, blocks, chunk):
    """Format invariants: every stored entry appears exactly once with its value, all
    padding is zero, the AMB traversal reproduces the CSR SpMV."""
    g = load_golden(name)
    y_ref = g["y"]
    for seg in segs:
        for bs in blocks:
            a = oracle_d.csr2amb(g, seg, bs, chunk)
            v = a.sellcs_val
            assert np.count_nonzero(v) == g["nnz"]
            assert np.array_equal(np.sort(v[v != 0]), np.sort(g["val"]))
            assert a.nnz % (chunk * bs) == 0 and len(a.sellcs_col) == a.nnz // bs
            y = a.spmv(g["x"])
            assert oracle_d.ans_check(y_ref, y) == 0
            assert (a.y_pad[g["M"]:] == 0).all()
            if a.seg_num == 1:
                # one lane per row, same summation order with zeros interleaved: exact
                assert np.array_equal(y, oracle_d.csr_spmv(g["rpt"], g["col"], g["val"], g["x"]))


def test_amb_plan_model(oracle_d):
    g = load_golden("banded2k")
    seg, bs, by = oracle_d.amb_plan_model(g, 32)
    a = oracle_d.csr2amb(g, seg, bs, 32)
    assert a.footprint == by
    for s2, b2 in ((65536, 1), (1024, 3), (4096, 20)):
        assert oracle_d.csr2amb(g, s2, b2, 32).footprint >= by


@pytest.mark.parametrize("chunk", [32, 64])
@pytest.mark.parametrize("name,seg,bs,sigma", [
    ("banded2k", 65536, 1, 32768), ("banded2k", 1024, 3, 32768), ("banded2k", 300, 20, 32768),
    ("banded2k", 65536, 2, 512), ("rmat_s10", 65536, 1, 32768), ("rmat_s10", 256, 4, 32768),
    ("wide_seg", 4096, 5, 32768), ("wide_seg", 65536, 2, 1024), ("banded_signed1k", 65536, 2, 32768),
])
def test_amb_oracle_against_second_implementation(name, seg, bs, sigma, chunk, oracle_d):
    """The C oracle's CSR -> AMB conversion against tests/amb_numpy.py, an independent numpy restatement
    written from the reference's kernels: all seven arrays, the scalars, and the traversal."""
    import amb_numpy
    g = load_golden(name)
    ora = oracle_d.csr2amb(g, seg, bs, chunk, sigma)
    ref = amb_numpy.csr2amb(g, seg, bs, chunk, sigma)
    for k in ("c_size", "nnz", "pad_M", "seg_num"):
        assert ref[k] == getattr(ora, k), k
    for k in ("cs", "cl", "sellcs_col", "sellcs_val", "s_write_permutation", "s_write_permutation_offset",
              "write_permutation"):
        assert np.array_equal(ref[k], getattr(ora, k)), f"AMB array {k}: oracle and numpy restatement differ"
    y = amb_numpy.spmv(ref, g["x"])
    np.testing.assert_allclose(y, g["y"], rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(ora.spmv(g["x"]), y, rtol=1e-13, atol=1e-13)


def test_amb_second_implementation_on_test_mtx():
    """tests/amb_numpy.py on the reference's fixture: the hand-derived layout of SURVEY 8c."""
    import amb_numpy
    A = dict(M=5, N=5, rpt=np.array(TEST_MTX["rpt"]), col=np.array(TEST_MTX["col"]),
             val=np.array(TEST_MTX["val"], dtype=float))
    a = amb_numpy.csr2amb(A, 65536, 1, 32)
    assert a["c_size"] == 1 and a["nnz"] == 96 and a["cs"].tolist() == [0] and a["cl"].tolist() == [2]
    assert a["s_write_permutation"].tolist() == [2, 0, 4, 1, 3] + list(range(5, 32))
    assert a["sellcs_col"][:5].tolist() == [0, 0, 2, 1, 3] and a["sellcs_val"][:5].tolist() == [1, 10, 2, 20, 40]
    assert amb_numpy.spmv(a, np.array(TEST_MTX["x"], float)).tolist() == TEST_MTX["y"]
