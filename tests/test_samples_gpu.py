"""The four sample binaries with the reference's names and command lines
(cuda-c/Makefile:99-113; spgemm_hash.cu:79-94; spmv_amb.cu:75-118) on the reference's fixture:
they must print the reference's report lines and the 'Correct' verdict of its check functions."""
import os
import re
import subprocess

import pytest

from conftest import GOLDEN
import nsparse_amd as ns

pytestmark = pytest.mark.gpu
MTX = os.path.join(GOLDEN, "test.mtx")


def run(name, *args):
    exe = os.path.join(ns.capi.LIB_DIR, name)
    assert os.path.exists(exe), f"{exe} not built"
    r = subprocess.run([exe, *args], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


@pytest.mark.parametrize("prec", ["d", "s"])
def test_spgemm_hash_cli(prec):
    out = run(f"spgemm_hash_{prec}", MTX)
    assert out.count("Read mtx file:") == 2
    assert re.search(r"SpGEMM using CSR format \(Hash-based\): .*test\.mtx, [\d.]+\[GFLOPS\], [\d.]+\[ms\]", out)
    assert "(nnz of A): 9 =>" in out and "(Num of intermediate products): 19 =>" in out
    assert "(nnz of C): 11" in out
    assert "Calculation Result is Correct" in out


@pytest.mark.parametrize("prec", ["d", "s"])
@pytest.mark.parametrize("plan", [(), ("65536", "1"), ("3", "2")])
def test_amb_cli(prec, plan):
    out = run(f"amb_{prec}", MTX, *plan)
    m = re.search(r"Format Conversion Cost \(CSR=>AMB, (\d+)-(\d+)\): [\d.]+\[msec\]", out)
    assert m
    if plan:
        assert (m.group(1), m.group(2)) == plan
    assert re.search(r"SpMV using AMB format: .*test\.mtx, [\d.]+\[GFLOPS\], [\d.]+\[ms\]", out)
    assert "Calculation Result is Correct" in out


def test_amb_cli_keeps_the_plan_beside_the_matrix(tmp_path):
    """NSPARSE_BIN_CACHE=1: the first run writes <file>.plan (and <file>.csr.bin), the second run
    converts with that plan instead of searching; same layout, same verdict."""
    import shutil
    mtx = str(tmp_path / "test.mtx")
    shutil.copy(MTX, mtx)
    exe = os.path.join(ns.capi.LIB_DIR, "amb_d")
    env = dict(os.environ, NSPARSE_BIN_CACHE="1")
    runs = [subprocess.run([exe, mtx], capture_output=True, text=True, timeout=300, env=env) for _ in range(2)]
    assert all(r.returncode == 0 for r in runs), runs[-1].stderr[-2000:]
    assert os.path.exists(mtx + ".plan") and os.path.exists(mtx + ".csr.bin")
    assert "plan:" not in runs[0].stderr and "plan: " + mtx + ".plan" in runs[1].stderr
    conv = [re.search(r"CSR=>AMB, (\d+)-(\d+)\)", r.stdout).groups() for r in runs]
    assert conv[0] == conv[1]
    assert all("Calculation Result is Correct" in r.stdout for r in runs)


# ---- the loader and the sample drivers at scale (SuiteSparse files cannot be fetched: the stand-ins are
# written as Matrix Market files the way the collection ships them -- nsparse_write_mtx) ---------------------
def _write_standin(tmp_path, lib, kind, dims, flavour, name, seed=0x5EED0022):
    import ctypes as C
    from gpu_util import synth
    A = synth(lib, kind, *dims, seed=seed)
    m = lib.csr_from_numpy(A["rpt"], A["col"], A["val"], A["N"])
    path = str(tmp_path / name)
    assert lib.nsparse_write_mtx(C.byref(m), path.encode(), flavour) == 0
    return A, path


def test_cant_class_file_through_loader_and_spgemm_sample(tmp_path, oracle_d):
    """cant as SuiteSparse ships it: `real symmetric`, lower triangle, column-major sorted.  The loader must
    give back the generator's CSR exactly (mirroring, in-row order, nnz_max), and spgemm_hash_d must print the
    reference's report lines with the expected counts and the 'Correct' verdict
    (nsparse.cu:14-136, spgemm_hash.cu:79-94)."""
    import ctypes as C
    import numpy as np
    lib = ns.load("d")
    A, path = _write_standin(tmp_path, lib, 5, (9, 9, 257), 1, "cant_class.mtx")
    assert os.path.getsize(path) > 40e6
    m = ns.sfCSR()
    lib.init_csr_matrix_from_file(C.byref(m), path.encode())
    got = lib.csr_host_to_numpy(m)
    lib.release_cpu_csr(m)
    assert (got["M"], got["N"], got["nnz"]) == (A["M"], A["N"], A["nnz"]) and got["nnz_max"] == int(np.diff(A["rpt"]).max())
    assert np.array_equal(got["rpt"], A["rpt"]) and np.array_equal(got["col"], A["col"])
    assert np.array_equal(got["val"], A["val"])  # %.17g reads back bit-identical
    ref = oracle_d.spgemm_omp(A, A)
    n_prod = int(oracle_d.nprod(A["rpt"], A["col"], A["rpt"])[1])
    out = run("spgemm_hash_d", path)
    assert out.count("Read mtx file:") == 2
    mm = re.search(r"SpGEMM using CSR format \(Hash-based\): .*cant_class\.mtx, ([\d.]+)\[GFLOPS\], ([\d.]+)\[ms\]", out)
    assert mm and float(mm.group(1)) > 100
    assert f"(nnz of A): {A['nnz']} =>" in out and f"(nnz of C): {ref['nnz']}" in out
    assert f"(Num of intermediate products): {n_prod} =>" in out
    assert "Calculation Result is Correct" in out
    print(f"[sample] spgemm_hash_d cant-class file: {mm.group(1)} GFLOPS, {mm.group(2)} ms")


def test_cant_class_file_through_amb_sample(tmp_path):
    """amb_d on the cant-class file: conversion line, SpMV line, 'Correct' from ans_check against the CPU
    csr_kernel (spmv_amb.cu:75-118); float build on the regular brick written as `general`."""
    lib = ns.load("d")
    A, path = _write_standin(tmp_path, lib, 5, (9, 9, 257), 1, "cant_class.mtx")
    out = run("amb_d", path)
    assert re.search(r"Format Conversion Cost \(CSR=>AMB, \d+-\d+\): [\d.]+\[msec\]", out)
    mm = re.search(r"SpMV using AMB format: .*cant_class\.mtx, ([\d.]+)\[GFLOPS\], ([\d.]+)\[ms\]", out)
    assert mm and float(mm.group(1)) > 50
    assert "Calculation Result is Correct" in out
    libs = ns.load("s")
    As, paths = _write_standin(tmp_path, libs, 0, (6, 6, 40), 0, "brick_general.mtx")
    out = run("amb_s", paths)
    assert "Calculation Result is Correct" in out
    out = run("spgemm_hash_s", paths)
    assert f"(nnz of A): {As['nnz']} =>" in out and "Calculation Result is Correct" in out


def test_webbase_class_general_and_pattern_files(tmp_path, oracle_s):
    """A `general` file (webbase-1M is one) at 1 M rows through the float loader + C = A^2 through the library,
    and a `pattern symmetric` file (every value 1.0, mirrored) through the double loader."""
    import ctypes as C
    import numpy as np
    from gpu_util import spgemm
    libs = ns.load("s")
    A, path = _write_standin(tmp_path, libs, 4, (1000005, 3105536, 0), 0, "web_general.mtx")
    m = ns.sfCSR()
    libs.init_csr_matrix_from_file(C.byref(m), path.encode())
    got = libs.csr_host_to_numpy(m)
    libs.release_cpu_csr(m)
    assert np.array_equal(got["rpt"], A["rpt"]) and np.array_equal(got["col"], A["col"]) and np.array_equal(got["val"], A["val"])
    c, st = spgemm(libs, got)
    ref = oracle_s.spgemm_omp(A, A)
    assert np.array_equal(c["rpt"], ref["rpt"]) and np.array_equal(c["col"], ref["col"])
    lib = ns.load("d")
    B, pathb = _write_standin(tmp_path, lib, 5, (5, 5, 30), 3, "pattern_sym.mtx")
    lib.init_csr_matrix_from_file(C.byref(m), pathb.encode())
    gotb = lib.csr_host_to_numpy(m)
    lib.release_cpu_csr(m)
    assert np.array_equal(gotb["rpt"], B["rpt"]) and np.array_equal(gotb["col"], B["col"]) and (gotb["val"] == 1.0).all()


@pytest.mark.parametrize("prec", ["d", "s"])
def test_amb_dist_cli_on_one_gpu(prec, tmp_path):
    """amb_dist_{d,s}: the multi-GPU twin of amb_{d,s} (one thread per GPU, include/nsparse_dist.h) with one GPU:
    partition, row block, conversion, the native timed loops, 'Correct' from ans_check against csr_kernel."""
    lib = ns.load(prec)
    A, path = _write_standin(tmp_path, lib, 0, (6, 6, 30), 0, "brick_general.mtx")
    out = run(f"amb_dist_{prec}", path, "1")
    assert f"rank 0: rows [0, {A['M']})  nnz {A['nnz']}" in out
    assert re.search(r"SpMV using AMB format on 1 GPUs: .*brick_general\.mtx, [\d.]+\[GFLOPS\], [\d.]+\[ms\]", out)
    assert "Calculation Result is Correct" in out
    out = run(f"amb_dist_{prec}", MTX, "1", "65536", "1")
    assert "Calculation Result is Correct" in out
