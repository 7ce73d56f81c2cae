"""rocSPARSE (through its C API, libnsparse_vendor_{d,s}.so) as the THIRD oracle: the role cuSPARSE
plays for the reference -- spgemm_hash.cu:60-68 holds spgemm_kernel_hash against spgemm_cu_csr with
check_spgemm_answer (exact nnz / rpt / col, values to 1e-9 / 1e-6).  Here the same check runs between
the HIP path, rocSPARSE and the CPU oracle; and rocsparse csrmv stands in for spmv_cu_csr.cu:13-85."""
import ctypes as C

import numpy as np
import pytest

import nsparse_amd as ns
from gpu_util import synth

pytestmark = pytest.mark.gpu


def _vendor_product(lib, vl, A):
    a = lib.csr_from_numpy(A["rpt"], A["col"], A["val"], A["N"])
    b = lib.csr_from_numpy(A["rpt"], A["col"], A["val"], A["N"])
    lib.csr_memcpy(C.byref(a))
    lib.csr_memcpy(C.byref(b))
    c, cv = ns.sfCSR(), ns.sfCSR()
    lib.spgemm_kernel_hash(C.byref(a), C.byref(b), C.byref(c))
    ms = C.c_float()
    vl.nsparse_vendor_spgemm(C.byref(a), C.byref(b), C.byref(cv), C.byref(ms))
    assert vl.nsparse_vendor_last_error() == 0
    # the reference's own verdict function on the two device results, brought to the host
    lib.csr_memcpyDtH(C.byref(c))
    lib.csr_memcpyDtH(C.byref(cv))
    fails = lib.nsparse_check_spgemm_count(C.byref(c), C.byref(cv))
    got, ven = lib.csr_host_to_numpy(c), lib.csr_host_to_numpy(cv)
    lib.release_cpu_csr(c)
    lib.release_cpu_csr(cv)
    lib.release_csr(c)
    vl.nsparse_vendor_release_csr(cv)
    lib.release_csr(a)
    lib.release_csr(b)
    return got, ven, fails, ms.value


@pytest.mark.parametrize("name,kind,p,prec", [
    ("cant class (regular brick)", 0, (9, 9, 257), "d"),
    ("cant class (irregular)", 5, (9, 9, 257), "d"),
    ("R-MAT scale 14", 3, (14, 16, 0), "d"),
    ("webbase class, 60 K rows", 4, (60000, 190000, 0), "s"),
])
def test_hip_path_equals_rocsparse(name, kind, p, prec, lib_d, lib_s, oracle_d, oracle_s):
    lib, orc = (lib_d, oracle_d) if prec == "d" else (lib_s, oracle_s)
    vl = ns.load_vendor(prec)
    A = synth(lib, kind, *p, seed=0x5EED0022)
    got, ven, fails, ms = _vendor_product(lib, vl, A)
    print(f"[vendor] {name}: nnz(C)={got['nnz']} rocSPARSE device stages {ms:.3f} ms")
    assert ven["nnz"] == got["nnz"]
    assert np.array_equal(ven["rpt"], got["rpt"]), "rocSPARSE C.rpt differs"
    vcol, vval = ven["col"], ven["val"]
    if not np.array_equal(vcol, got["col"]):
        # rocSPARSE does not promise ascending columns inside a row: order its rows, then compare
        order = np.lexsort((vcol, np.repeat(np.arange(A["M"]), np.diff(ven["rpt"]))))
        vcol, vval = vcol[order], vval[order]
        fails = orc.check_spgemm(got, dict(ven, col=vcol, val=vval))
    assert np.array_equal(vcol, got["col"]), "rocSPARSE C.col differs"
    if prec == "d":
        assert fails == 0, "values: HIP path vs rocSPARSE outside the reference tolerance"
    else:  # both sum floats in their own order: the float rule holds against neither's order; 1e-5
        np.testing.assert_allclose(got["val"], vval, rtol=1e-5)
    # and both agree with the CPU oracle's structure
    ref = orc.spgemm(A, A)
    assert np.array_equal(ref["rpt"], got["rpt"]) and np.array_equal(ref["col"], got["col"])


def test_spgemm_cu_csr_name_is_exported(lib_d):
    """The reference's own name for the vendor product (nsparse.h:165) works as a drop-in."""
    vl = ns.load_vendor("d")
    A = synth(lib_d, 0, 4, 4, 8, seed=3)
    a = lib_d.csr_from_numpy(A["rpt"], A["col"], A["val"], A["N"])
    lib_d.csr_memcpy(C.byref(a))
    c, h = ns.sfCSR(), ns.sfCSR()
    vl.spgemm_cu_csr(C.byref(a), C.byref(a), C.byref(c))
    lib_d.spgemm_kernel_hash(C.byref(a), C.byref(a), C.byref(h))
    assert c.nnz == h.nnz and c.M == h.M
    lib_d.release_csr(c)  # hipMalloc'ed by the vendor library: the main library's release takes it
    lib_d.release_csr(h)
    lib_d.release_csr(a)


def test_rocsparse_csrmv_matches_amb(lib_d, oracle_d):
    from gpu_util import DeviceAMB
    vl = ns.load_vendor("d")
    A = synth(lib_d, 1, 60, 60, 60, seed=2)
    d = DeviceAMB(lib_d, A)
    x = np.random.default_rng(5).random(A["N"])
    y = d.spmv(x)
    d_y = lib_d.dmalloc((A["M"] + 64) * 8)
    ms = vl.nsparse_vendor_spmv_csr(d_y, C.byref(d.csr), d.d_x, 3)
    assert vl.nsparse_vendor_last_error() == 0 and ms > 0
    yv = lib_d.d2h(d_y, (A["M"],), np.float64)
    assert oracle_d.ans_check(yv, y) == 0
    assert oracle_d.ans_check(oracle_d.csr_spmv(A["rpt"], A["col"], A["val"], x), yv) == 0
    lib_d.dfree(d_y)
    d.close()
