"""Parity of the device CSR->AMB conversion (bit-exact arrays) and of the AMB SpMV kernel
(reference ans_check rule: 1e-8 double / 1e-5 float relative) with the CPU oracle."""
import ctypes as C
import os

import numpy as np
import pytest

import nsparse_amd as ns
from conftest import GOLDEN, TEST_MTX, load_golden
from gpu_util import DeviceAMB, synth, variant_env

pytestmark = pytest.mark.gpu

ARRAYS = ("cs", "cl", "sellcs_col", "sellcs_val", "s_write_permutation",
          "s_write_permutation_offset", "write_permutation")


def assert_same_format(dev, ora):
    for k in ("c_size", "nnz", "pad_M", "chunk", "block_size", "seg_num"):
        assert dev[k] == getattr(ora, k), k
    for k in ARRAYS:
        assert np.array_equal(dev[k], getattr(ora, k)), f"AMB array {k} differs from the oracle"


def test_test_mtx_layout_chunk32(lib_d, oracle_d):
    m = ns.sfCSR()
    lib_d.init_csr_matrix_from_file(C.byref(m), os.path.join(GOLDEN, "test.mtx").encode())
    A = lib_d.csr_host_to_numpy(m)
    lib_d.release_cpu_csr(m)
    d = DeviceAMB(lib_d, A, 65536, 1, chunk=32)
    arr = d.arrays()
    assert arr["cs"].tolist() == [0] and arr["cl"].tolist() == [2] and arr["nnz"] == 96
    assert arr["s_write_permutation"].tolist() == [2, 0, 4, 1, 3] + list(range(5, 32))
    assert_same_format(arr, oracle_d.csr2amb(A, 65536, 1, 32))
    assert d.spmv(np.array(TEST_MTX["x"], float)).tolist() == TEST_MTX["y"]
    d.close()


@pytest.mark.parametrize("chunk", [32, 64])
@pytest.mark.parametrize("prec", ["d", "s"])
@pytest.mark.parametrize("name,seg,bs", [
    ("banded2k", 65536, 1), ("banded2k", 1024, 3), ("banded2k", 300, 20), ("banded2k", 65536, 7),
    ("rmat_s10", 65536, 1), ("rmat_s10", 256, 4),
    ("wide_seg", 65536, 1), ("wide_seg", 4096, 5), ("wide_seg", 65536, 2),
    ("banded_signed1k", 65536, 2),
    # block sizes whose whole-row depth NB is derived from the register budget (spmv_amb.hip launch_bs, round 6)
    ("banded2k", 2048, 11), ("banded2k", 65536, 12), ("rmat_s10", 65536, 6), ("wide_seg", 65536, 9),
])
def test_manual_plan_bit_exact(name, seg, bs, prec, chunk, lib_d, lib_s, oracle_d, oracle_s):
    lib, orc = (lib_d, oracle_d) if prec == "d" else (lib_s, oracle_s)
    g = load_golden(name)
    g = dict(g, val=g["val"].astype(lib.real))
    d = DeviceAMB(lib, g, seg, bs, chunk=chunk)
    ora = orc.csr2amb(g, seg, bs, chunk)
    assert_same_format(d.arrays(), ora)
    assert lib.nsparse_amb_footprint_bytes(C.byref(d.amb)) == ora.footprint
    x = g["x"].astype(lib.real)
    y = d.spmv(x)
    y_ref = orc.csr_spmv(g["rpt"], g["col"], g["val"], x)
    if "signed" in name:
        # cancelling rows: the reference's purely relative rule is ill-posed; scale by sum |a x|
        mag = orc.csr_spmv(g["rpt"], g["col"], np.abs(g["val"]), np.abs(x))
        assert (np.abs(y - y_ref) <= (1e-13 if prec == "d" else 1e-5) * mag).all()
    else:
        assert orc.ans_check(y_ref, y) == 0
    # same summation order as the oracle traversal; the GPU fuses multiply-add, gcc does not
    if "signed" not in name:
        np.testing.assert_allclose(y, ora.spmv(x), rtol=1e-13 if prec == "d" else 1e-5)
    if ora.seg_num == 1:
        # single segment: plain stores instead of atomics => bit-reproducible run to run
        assert np.array_equal(y, d.spmv(x))
    d.close()


@pytest.mark.parametrize("name", ["banded2k", "wide_seg", "rmat_s10"])
def test_auto_plan_matches_footprint_model(name, lib_d, oracle_d):
    g = load_golden(name)
    d = DeviceAMB(lib_d, g, chunk=64)
    seg, bs, by = oracle_d.amb_plan_model(g, 64)
    assert (d.plan.isPlan, d.plan.seg_size, d.plan.block_size) == (1, seg, bs)
    assert lib_d.nsparse_amb_footprint_bytes(C.byref(d.amb)) == by
    assert d.plan.thread_block in (64, 128, 256, 512, 1024)
    assert_same_format(d.arrays(), oracle_d.csr2amb(g, seg, bs, 64))
    y = d.spmv(g["x"])
    assert oracle_d.ans_check(g["y"], y) == 0
    d.close()


def test_unsorted_rows_are_sorted_internally(lib_d, oracle_d):
    g = load_golden("banded2k")
    rng = np.random.default_rng(3)
    col, val = g["col"].copy(), g["val"].copy()
    for i in range(g["M"]):
        b, e = g["rpt"][i], g["rpt"][i + 1]
        p = rng.permutation(e - b)
        col[b:e], val[b:e] = col[b:e][p], val[b:e][p]
    d = DeviceAMB(lib_d, dict(g, col=col, val=val), 65536, 2)
    assert_same_format(d.arrays(), oracle_d.csr2amb(g, 65536, 2, 64))
    d.close()


def test_sigma_windows_and_16bit_permutation(lib_d, oracle_d):
    """M > 65536: several sigma windows, per-chunk high part of the permutation non-zero."""
    A = synth(lib_d, 2, 150000, 500000, 0, seed=9)
    d = DeviceAMB(lib_d, A, 65536, 1)
    arr = d.arrays()
    assert arr["s_write_permutation_offset"].max() == 2
    assert_same_format(arr, oracle_d.csr2amb(A, 65536, 1, 64))
    x = np.random.default_rng(1).random(A["N"])
    assert oracle_d.ans_check(oracle_d.csr_spmv(A["rpt"], A["col"], A["val"], x), d.spmv(x)) == 0
    d.close()


def test_full_size_cant_class(lib_d, oracle_d):
    A = synth(lib_d, 0, 9, 9, 257, seed=0x5EED0022)
    d = DeviceAMB(lib_d, A)  # auto plan
    x = np.zeros(A["N"])
    lib_d.nsparse_init_vector_seeded(x.ctypes.data_as(C.c_void_p), A["N"], 0x5EED0001)
    y = d.spmv(x)
    assert oracle_d.ans_check(oracle_d.csr_spmv(A["rpt"], A["col"], A["val"], x), y) == 0
    # linearity: A(2x) == 2 A(x) exactly (power-of-two scaling commutes with rounding)
    assert np.array_equal(d.spmv(2 * x), 2 * y)
    d.close()


def test_timed_plan_search(lib_d, oracle_d):
    """NSPARSE_AMB_TUNE=timed: the reference's default plan search (convert_amb.cu:18 `#define AT`,
    :556-600 evaluate_spmv, :878-925) -- every (segment size, block size) candidate is built and
    timed.  Which candidate wins depends on the clock; what must hold: the plan written back is one
    of the candidates, the arrays are bit-exact for THAT plan, and y is right."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, json, ctypes as C, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r);"
        "import nsparse_amd as ns; from gpu_util import DeviceAMB; from conftest import load_golden;"
        "lib = ns.load('d'); g = load_golden('banded2k'); d = DeviceAMB(lib, g);"
        "y = d.spmv(g['x']); arr = d.arrays();"
        "np.savez(sys.argv[1], y=y, **{k: arr[k] for k in ('cs','cl','sellcs_col','sellcs_val','s_write_permutation','s_write_permutation_offset','write_permutation')});"
        "print(json.dumps(dict(seg=int(d.plan.seg_size), bs=int(d.plan.block_size), isplan=int(d.plan.isPlan),"
        " tb=int(d.plan.thread_block), nnz=int(d.amb.nnz), c_size=int(d.amb.c_size))))"
    ) % (root, os.path.join(root, "tests"))
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "amb.npz")
        r = subprocess.run([sys.executable, "-c", code, out], cwd=root, capture_output=True, text=True,
                           env=dict(os.environ, NSPARSE_AMB_TUNE="timed"))
        assert r.returncode == 0, r.stderr[-2000:]
        info = json.loads(r.stdout.strip().splitlines()[-1])
        z = np.load(out)
        g = load_golden("banded2k")
        assert info["isplan"] == 1 and 1 <= info["bs"] <= 20 and info["tb"] in (64, 128, 256, 512, 1024)
        assert info["seg"] in (65536, 1024, 2048, 3072, 4096)  # sf_csr2amb's candidates for N < 128 K
        ora = oracle_d.csr2amb(g, info["seg"], info["bs"], 64)
        assert info["nnz"] == ora.nnz and info["c_size"] == ora.c_size
        for k in ARRAYS:
            assert np.array_equal(z[k], getattr(ora, k)), f"AMB array {k} differs for the timed plan"
        assert oracle_d.ans_check(g["y"], z["y"]) == 0


def test_no_rows(lib_d):
    """M = 0: an AMB with no chunks; conversion, SpMV and release are all no-ops that do not fail."""
    A = dict(M=0, N=500, rpt=np.zeros(1, np.int32), col=np.zeros(0, np.int32), val=np.zeros(0))
    d = DeviceAMB(lib_d, A)
    assert d.amb.c_size == 0 and d.plan.isPlan == 1 and lib_d.nsparse_last_error() == 0
    assert d.spmv(np.ones(500)).shape == (0,)
    d.close()


@pytest.mark.parametrize("split", ["1", "2", "8"])
def test_split_row_spmv_for_cache_resident_matrices(split, oracle_d):
    """NSPARSE_SPMV_SPLIT (round 4, opt-in): one chunk per workgroup of W wavefronts, wavefront w taking the blocks
    w, w + W, ... of every row, partial sums folded through LDS -- same y as the CPU loop (ans_check), bit-identical
    from run to run with one column segment, nothing written past row M, on a brick (rows of 81 entries: wider than
    the whole-row form holds) and on a power-law matrix (chunks of very different widths, several segments)."""
    import subprocess
    import sys
    from conftest import ROOT
    code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import nsparse_amd as ns; from gpu_util import synth, DeviceAMB; from oracle.oracle import Oracle\n"
            "lib, orc = ns.load('d'), Oracle('d')\n"
            "for kind, dims, seg in ((0, (6, 6, 40), None), (4, (30000, 95000, 0), None), (3, (13, 8, 0), (2048, 2))):\n"
            "    A = synth(lib, kind, *dims, seed=11); x = np.random.default_rng(2).random(A['N'])\n"
            "    d = DeviceAMB(lib, A) if seg is None else DeviceAMB(lib, A, seg[0], seg[1])\n"
            "    y1 = d.spmv(x); y2 = d.spmv(x)\n"
            "    ref = orc.csr_spmv(A['rpt'], A['col'], A['val'], x)\n"
            "    assert orc.ans_check(ref, y1[:A['M']]) == 0, kind\n"
            "    assert (y1[A['M']:] == 7.0).all(), 'wrote past row M'\n"
            "    if d.amb.seg_num == 1: assert y1.tobytes() == y2.tobytes(), 'not reproducible'\n"
            "    d.close()\n"
            "print('SPLIT_OK')\n" % (ROOT, os.path.join(ROOT, "tests")))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT,
                       env=dict(os.environ, **variant_env(dict(NSPARSE_SPMV_SPLIT=split))))
    assert r.returncode == 0 and "SPLIT_OK" in r.stdout, r.stderr[-2000:]
