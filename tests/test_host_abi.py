"""CPU-side tests of the C-ABI library: it loads, exports every symbol include/nsparse.h
declares, and its host functions (loader, plan, CPU SpMV, answer checks, generators) agree
with the oracle.  No device call is made here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import nsparse_amd as ns
from conftest import GOLDEN, ROOT, TEST_MTX, load_golden


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "nsparse.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"^[A-Za-z_][\w \*]*?\b(\w+)\s*\([^;{]*\)\s*;", src, flags=re.M)
    return sorted(set(n for n in names if n not in ("defined",)))


@pytest.mark.parametrize("prec", ["d", "s"])
def test_exports_every_declared_symbol(prec):
    names = _declared_functions()
    assert len(names) >= 30 and "spgemm_kernel_hash" in names and "sf_spmv_amb" in names
    dll = C.CDLL(os.path.join(ns.capi.LIB_DIR, f"libnsparse_{prec}.so"))
    for n in names:
        assert hasattr(dll, n), f"{n} declared in nsparse.h but not exported"
    # and the binding table covers the header exactly
    assert sorted(ns.SIGNATURES) == names


def test_struct_layout_matches_header(tmp_path):
    """Compile include/nsparse.h with gcc (as C) and compare sizeof / offsetof with capi.py."""
    import subprocess
    fields = {"sfPlan": ["thread_grid", "isPlan", "seg_size", "block_size"],
              "sfCSR": ["rpt", "d_val", "M", "nnz_max", "matrix_name"],
              "sfAMB": ["cs", "d_cs", "d_write_permutation", "block_size", "chunk", "c_size",
                        "seg_size", "matrix_name"]}
    prog = ["#include <stdio.h>", "#include <stddef.h>", '#include "nsparse.h"', "int main(void){"]
    for st, fl in fields.items():
        prog.append(f'printf("{st} %zu\\n", sizeof({st}));')
        for f in fl:
            prog.append(f'printf("{st}.{f} %zu\\n", offsetof({st}, {f}));')
    prog.append("return 0;}")
    src = tmp_path / "abi.c"
    src.write_text("\n".join(prog))
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    for st, fl in fields.items():
        cls = getattr(ns, st)
        assert int(got[st]) == C.sizeof(cls), st
        for f in fl:
            assert int(got[f"{st}.{f}"]) == getattr(cls, f).offset, f"{st}.{f}"


def _write(tmp_path, name, text):
    p = tmp_path / name
    p.write_text(text)
    return str(p).encode()


def _load(lib, path):
    m = ns.sfCSR()
    lib.init_csr_matrix_from_file(C.byref(m), path)
    out = lib.csr_host_to_numpy(m)
    lib.release_cpu_csr(m)
    return out


def test_loader_test_mtx(lib_d, oracle_d):
    A = _load(lib_d, os.path.join(GOLDEN, "test.mtx").encode())
    for k in ("M", "N", "nnz", "nnz_max"):
        assert A[k] == TEST_MTX[k]
    assert A["rpt"].tolist() == TEST_MTX["rpt"] and A["col"].tolist() == TEST_MTX["col"]
    assert A["val"].tolist() == TEST_MTX["val"]


CASES = {
    "general.mtx": "%%MatrixMarket matrix coordinate real general\n% c\n3 4 5\n1 1 1.5\n3 2 -2\n1 4 3e2\n2 2 4\n3 4 0.125\n",
    "symmetric.mtx": "%%MatrixMarket matrix coordinate real symmetric\n4 4 5\n1 1 1\n2 1 2\n4 1 3\n3 3 4\n4 3 5\n",
    "skew.mtx": "%%MatrixMarket matrix coordinate real skew-symmetric\n3 3 2\n2 1 2.5\n3 2 -1\n",
    "pattern.mtx": "%%MatrixMarket matrix coordinate pattern symmetric\n3 3 3\n1 1\n3 1\n3 3\n",
    "pattern_general.mtx": "%%MatrixMarket matrix coordinate pattern general\n2 3 3\n1 3\n2 1\n2 2\n",
    "complex.mtx": "%%MatrixMarket matrix coordinate complex general\n2 2 2\n1 1 1.5 9\n2 2 -3 7\n",
    "unsorted.mtx": "%%MatrixMarket matrix coordinate real general\n3 3 4\n3 3 1\n1 2 2\n1 1 3\n3 1 4\n",
    "emptyrow.mtx": "%%MatrixMarket matrix coordinate real general\n4 4 2\n1 1 1\n4 4 2\n",
}


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("prec", ["d", "s"])
def test_loader_quirks_match_oracle(tmp_path, name, prec, oracle_d, oracle_s, lib_d, lib_s):
    """general vs mirrored (same sign, also for skew), missing value => 1.0, complex => real part,
    in-row order = file order, no sorting, no duplicate merge (reference nsparse.cu:14-136)."""
    lib, orc = (lib_d, oracle_d) if prec == "d" else (lib_s, oracle_s)
    path = _write(tmp_path, name, CASES[name])
    A, O = _load(lib, path), orc.load_mtx(path.decode())
    for k in ("M", "N", "nnz", "nnz_max"):
        assert A[k] == O[k], k
    assert np.array_equal(A["rpt"], O["rpt"]) and np.array_equal(A["col"], O["col"])
    assert np.array_equal(A["val"], O["val"])
    if name == "skew.mtx":
        assert A["val"].tolist() == [2.5, 2.5, -1, -1]  # mirrored with the SAME sign
    if name == "pattern.mtx":
        assert set(A["val"].tolist()) == {1.0}


def test_loader_missing_file_exits(tmp_path):
    import subprocess, sys
    code = ("import ctypes as C, nsparse_amd as ns; L = ns.load('d'); m = ns.sfCSR();"
            "L.init_csr_matrix_from_file(C.byref(m), b'/nonexistent.mtx')")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 1 and "Cannot find file" in r.stdout


@pytest.mark.parametrize("prec", ["d", "s"])
def test_csr_kernel_and_checks(prec, oracle_d, oracle_s, lib_d, lib_s):
    lib, orc = (lib_d, oracle_d) if prec == "d" else (lib_s, oracle_s)
    g = load_golden("banded_signed1k")
    m = lib.csr_from_numpy(g["rpt"], g["col"], g["val"], g["N"])
    x = g["x"].astype(lib.real)
    y = np.zeros(g["M"], lib.real)
    lib.csr_kernel(y.ctypes.data_as(C.c_void_p), C.byref(m), x.ctypes.data_as(C.c_void_p))
    assert np.array_equal(y, orc.csr_spmv(g["rpt"], g["col"], g["val"], x))  # same loop order
    # ans_check rule: relative 1e-8 (double) / 1e-5 (float)
    tol = 1e-8 if prec == "d" else 1e-5
    bad = y.copy()
    bad[3] *= 1 + 4 * tol
    bad[7] *= 1 - 4 * tol
    ok = y.copy()
    ok[5] *= 1 + tol / 4
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    assert lib.nsparse_ans_check_count(p(y), p(bad), len(y)) == 2 == orc.ans_check(y, bad)
    assert lib.nsparse_ans_check_count(p(y), p(ok), len(y)) == 0 == orc.ans_check(y, ok)


def test_check_spgemm_rule(lib_d, oracle_d):
    g = load_golden("banded2k")
    ans = lib_d.csr_from_numpy(g["c_rpt"], g["c_col"], g["c_val"], g["N"])
    same = lib_d.csr_from_numpy(g["c_rpt"], g["c_col"], g["c_val"] * (1 + 1e-11), g["N"])
    assert lib_d.nsparse_check_spgemm_count(C.byref(same), C.byref(ans)) == 0
    off = g["c_val"].copy()
    off[10] *= 1 + 1e-8  # beyond 1e-9
    worse = lib_d.csr_from_numpy(g["c_rpt"], g["c_col"], off, g["N"])
    assert lib_d.nsparse_check_spgemm_count(C.byref(worse), C.byref(ans)) == 1
    col2 = g["c_col"].copy()
    col2[[0, 1]] = col2[[1, 0]]  # unsorted columns are a structural failure
    swapped = lib_d.csr_from_numpy(g["c_rpt"], col2, g["c_val"], g["N"])
    assert lib_d.nsparse_check_spgemm_count(C.byref(swapped), C.byref(ans)) == -3
    rpt2 = g["c_rpt"].copy()
    rpt2[5] += 1
    shifted = lib_d.csr_from_numpy(rpt2, g["c_col"], g["c_val"], g["N"])
    assert lib_d.nsparse_check_spgemm_count(C.byref(shifted), C.byref(ans)) == -2
    short = lib_d.csr_from_numpy(g["c_rpt"], g["c_col"], g["c_val"], g["N"])
    short.nnz -= 1
    assert lib_d.nsparse_check_spgemm_count(C.byref(short), C.byref(ans)) == -1


def test_plan_clamps(lib_d):
    p = ns.sfPlan()
    lib_d.init_plan(C.byref(p))
    assert p.isPlan == 0
    lib_d.set_plan(C.byref(p), 1 << 20, 0)
    assert (p.isPlan, p.seg_size, p.block_size) == (1, 65536, 1)
    lib_d.set_plan(C.byref(p), 2048, 21)
    assert (p.seg_size, p.block_size) == (2048, 1)
    lib_d.set_plan(C.byref(p), 4096, 20)
    assert (p.seg_size, p.block_size) == (4096, 20)


def test_seeded_vector(lib_d, lib_s):
    a, b = np.zeros(1000), np.zeros(1000)
    lib_d.nsparse_init_vector_seeded(a.ctypes.data_as(C.c_void_p), 1000, 0x5EED0001)
    lib_d.nsparse_init_vector_seeded(b.ctypes.data_as(C.c_void_p), 1000, 0x5EED0001)
    assert np.array_equal(a, b) and a.min() >= 0 and a.max() < 1 and 0.4 < a.mean() < 0.6
    f = np.zeros(1000, np.float32)
    lib_s.nsparse_init_vector_seeded(f.ctypes.data_as(C.c_void_p), 1000, 0x5EED0001)
    np.testing.assert_allclose(f, a, rtol=1e-6)


@pytest.mark.parametrize("kind,p,expect_m", [(0, (3, 4, 5), 180), (1, (6, 5, 4), 120),
                                             (2, (5000, 16000, 0), 5000), (3, (10, 8, 0), 1024)])
def test_synth_generators(lib_d, kind, p, expect_m):
    m = ns.sfCSR()
    lib_d.nsparse_synth_csr(C.byref(m), kind, p[0], p[1], p[2], 42, 0, 0)
    A = lib_d.csr_host_to_numpy(m)
    assert A["M"] == expect_m and A["rpt"][0] == 0 and A["rpt"][-1] == A["nnz"]
    assert A["col"].min() >= 0 and A["col"].max() < A["N"]
    for i in range(0, A["M"], max(1, A["M"] // 50)):  # strictly ascending columns
        c = A["col"][A["rpt"][i]:A["rpt"][i + 1]]
        assert (np.diff(c) > 0).all()
    # a row block equals the same rows of the full matrix (row-sharded generation)
    if kind in (0, 1, 2):
        lo, hi = expect_m // 3, 2 * expect_m // 3
        b = ns.sfCSR()
        lib_d.nsparse_synth_csr(C.byref(b), kind, p[0], p[1], p[2], 42, lo, hi)
        B = lib_d.csr_host_to_numpy(b)
        assert B["M"] == hi - lo and B["N"] == A["N"]
        assert np.array_equal(B["col"], A["col"][A["rpt"][lo]:A["rpt"][hi]])
        assert np.array_equal(B["val"], A["val"][A["rpt"][lo]:A["rpt"][hi]])
        lib_d.release_cpu_csr(b)
    lib_d.release_cpu_csr(m)


def test_binary_matrix_cache(tmp_path, lib_d, lib_s):
    """.csr.bin image == what the text loader produces; wrong-precision images are rejected;
    NSPARSE_BIN_CACHE=1 makes the loader write and then reuse the image."""
    import subprocess, sys
    path = _write(tmp_path, "sym.mtx", CASES["symmetric.mtx"])
    m = ns.sfCSR()
    lib_d.init_csr_matrix_from_file(C.byref(m), path)
    A = lib_d.csr_host_to_numpy(m)
    img = str(tmp_path / "a.bin").encode()
    assert lib_d.nsparse_save_csr_bin(C.byref(m), img) == 0
    lib_d.release_cpu_csr(m)
    b = ns.sfCSR()
    assert lib_d.nsparse_load_csr_bin(C.byref(b), img) == 0
    B = lib_d.csr_host_to_numpy(b)
    lib_d.release_cpu_csr(b)
    for k in ("M", "N", "nnz", "nnz_max"):
        assert A[k] == B[k]
    assert all(np.array_equal(A[k], B[k]) for k in ("rpt", "col", "val"))
    s = ns.sfCSR()
    assert lib_s.nsparse_load_csr_bin(C.byref(s), img) == -2
    assert lib_d.nsparse_load_csr_bin(C.byref(s), b"/nonexistent.bin") == -1
    code = ("import ctypes as C, nsparse_amd as ns; L = ns.load('d'); m = ns.sfCSR();"
            "L.init_csr_matrix_from_file(C.byref(m), %r); print(m.M, m.nnz, m.nnz_max)" % path)
    env = dict(os.environ, NSPARSE_BIN_CACHE="1")
    r1 = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, env=env)
    assert r1.returncode == 0 and os.path.exists(path.decode() + ".csr.bin")
    r2 = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, env=env)
    want = f"{A['M']} {A['nnz']} {A['nnz_max']}"  # C stdio and Python print interleave freely
    assert r2.returncode == 0 and want in r1.stdout.splitlines() and want in r2.stdout.splitlines()


def test_plan_file_round_trip(tmp_path, lib_d, lib_s):
    """nsparse_save_plan / nsparse_load_plan: every field back, isPlan TRUE; a plan written by the
    other precision build or for the other chunk size is refused and leaves the plan alone."""
    p = ns.sfPlan()
    lib_d.set_plan(C.byref(p), 4096, 7)
    p.thread_block, p.thread_grid, p.SIGMA, p.seg_num = 256, 1234, 32767, 16
    path = str(tmp_path / "m.mtx.plan").encode()
    assert lib_d.nsparse_save_plan(C.byref(p), path) == 0
    q = ns.sfPlan()
    lib_d.init_plan(C.byref(q))
    assert lib_d.nsparse_load_plan(C.byref(q), path) == 0
    assert (q.isPlan, q.seg_size, q.block_size, q.thread_block, q.thread_grid, q.SIGMA, q.seg_num) == \
        (1, 4096, 7, 256, 1234, 32767, 16)
    r = ns.sfPlan()
    lib_s.init_plan(C.byref(r))
    assert lib_s.nsparse_load_plan(C.byref(r), path) == -3 and r.isPlan == 0
    old = lib_d.nsparse_set_amb_chunk(32)
    try:
        assert lib_d.nsparse_load_plan(C.byref(r), path) == -3
    finally:
        lib_d.nsparse_set_amb_chunk(64)
    assert old == 32 and lib_d.nsparse_load_plan(C.byref(r), b"/nonexistent.plan") == -1
    open(path, "w").write("garbage\n")
    assert lib_d.nsparse_load_plan(C.byref(r), path) == -2
