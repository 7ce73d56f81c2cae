"""ctypes front end of the CPU oracle (oracle/nsparse_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by the
cpu_baseline leg of bench.py.  The product package (nsparse_amd/) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(force=False):
    """Compile liboracle_{d,s}.so with gcc (seconds)."""
    need = force or not all(
        os.path.exists(os.path.join(_HERE, f"liboracle_{p}.so")) for p in "ds")
    if not need:
        src = os.path.getmtime(os.path.join(_HERE, "nsparse_oracle.c"))
        need = any(os.path.getmtime(os.path.join(_HERE, f"liboracle_{p}.so")) < src for p in "ds")
    if need:
        subprocess.check_call(["make", "-C", _HERE, "-B", "-s"])


class _AMB(C.Structure):
    _fields_ = [
        ("cs", C.POINTER(C.c_int)),
        ("cl", C.POINTER(C.c_uint)),
        ("sellcs_col", C.POINTER(C.c_ushort)),
        ("sellcs_val", C.c_void_p),
        ("s_write_permutation", C.POINTER(C.c_ushort)),
        ("s_write_permutation_offset", C.POINTER(C.c_ushort)),
        ("write_permutation", C.POINTER(C.c_int)),
        ("block_size", C.c_int), ("nnz", C.c_int), ("M", C.c_int), ("N", C.c_int),
        ("pad_M", C.c_int), ("chunk", C.c_int), ("SIGMA", C.c_int),
        ("group_num_col", C.c_int), ("c_size", C.c_int),
        ("seg_size", C.c_longlong), ("seg_num", C.c_longlong),
        ("packed_cl", C.POINTER(C.c_int)),
        ("packed_cs", C.POINTER(C.c_int)),
        ("ell_col", C.POINTER(C.c_ushort)),
        ("ell_val", C.c_void_p),
        ("ell_nnz", C.c_int),
    ]


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


class AMBResult:
    """Host copy of an oracle AMB matrix (numpy arrays + scalars)."""

    def __init__(self, orc, st):
        real = orc.real
        n, cs, ch, bs = st.nnz, st.c_size, st.chunk, st.block_size
        self.cs = np.ctypeslib.as_array(st.cs, (max(cs, 1),))[:cs].copy()
        self.cl = np.ctypeslib.as_array(st.cl, (max(cs, 1),))[:cs].copy()
        self.sellcs_col = np.ctypeslib.as_array(st.sellcs_col, (max(n // bs, 1),))[:n // bs].copy()
        vbuf = (C.c_byte * (max(n, 1) * real().itemsize)).from_address(st.sellcs_val)
        self.sellcs_val = np.frombuffer(vbuf, dtype=real)[:n].copy()
        self.s_write_permutation = np.ctypeslib.as_array(
            st.s_write_permutation, (max(cs * ch, 1),))[:cs * ch].copy()
        self.s_write_permutation_offset = np.ctypeslib.as_array(
            st.s_write_permutation_offset, (max(cs, 1),))[:cs].copy()
        self.write_permutation = np.ctypeslib.as_array(
            st.write_permutation, (max(cs * ch, 1),))[:cs * ch].copy()
        for k in ("block_size", "nnz", "M", "N", "pad_M", "chunk", "SIGMA", "group_num_col",
                  "c_size", "seg_size", "seg_num", "ell_nnz"):
            setattr(self, k, int(getattr(st, k)))
        self.footprint = int(orc.lib.orc_amb_footprint(C.byref(st)))
        self._st = st
        self._orc = orc

    def spmv(self, x):
        """AMB traversal on the CPU; returns y[:M] (and keeps y_pad for tests)."""
        real = self._orc.real
        xp = np.zeros(self.N + 20, dtype=real)
        xp[:self.N] = x
        y = np.zeros(self.pad_M, dtype=real)
        self._orc.lib.orc_amb_spmv(C.byref(self._st), xp.ctypes.data_as(C.c_void_p),
                                   y.ctypes.data_as(C.c_void_p))
        self.y_pad = y
        return y[:self.M].copy()

    def __del__(self):
        try:
            self._orc.lib.orc_amb_free(C.byref(self._st))
        except Exception:
            pass


class Oracle:
    def __init__(self, precision="d"):
        assert precision in ("d", "s")
        build()
        self.precision = precision
        self.real = np.float64 if precision == "d" else np.float32
        self.lib = C.CDLL(os.path.join(_HERE, f"liboracle_{precision}.so"), mode=C.RTLD_LOCAL)
        L = self.lib
        L.orc_amb_footprint.restype = C.c_longlong
        L.orc_free.argtypes = [C.c_void_p]
        assert L.orc_sizeof_real() == self.real().itemsize

    # ---- loader -------------------------------------------------------------
    def load_mtx(self, path):
        M, N, nnz, nmax = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        rpt, col = C.POINTER(C.c_int)(), C.POINTER(C.c_int)()
        val = C.c_void_p()
        rc = self.lib.orc_load_mtx(path.encode(), C.byref(M), C.byref(N), C.byref(nnz),
                                   C.byref(nmax), C.byref(rpt), C.byref(col), C.byref(val))
        if rc != 0:
            raise IOError(f"orc_load_mtx({path}) -> {rc}")
        m, z = M.value, nnz.value
        r = np.ctypeslib.as_array(rpt, (m + 1,)).copy()
        c = np.ctypeslib.as_array(col, (max(z, 1),))[:z].copy()
        vb = (C.c_byte * (max(z, 1) * self.real().itemsize)).from_address(val.value)
        v = np.frombuffer(vb, dtype=self.real)[:z].copy()
        self.lib.orc_free(C.cast(rpt, C.c_void_p))
        self.lib.orc_free(C.cast(col, C.c_void_p))
        self.lib.orc_free(val)
        return dict(M=m, N=N.value, nnz=z, nnz_max=nmax.value, rpt=r, col=c, val=v)

    # ---- SpMV ---------------------------------------------------------------
    def csr_spmv(self, rpt, col, val, x, omp=False):
        M = len(rpt) - 1
        y = np.empty(M, dtype=self.real)
        val = np.ascontiguousarray(val, dtype=self.real)
        x = np.ascontiguousarray(x, dtype=self.real)
        fn = self.lib.orc_csr_spmv_omp if omp else self.lib.orc_csr_spmv
        fn(M, _ip(rpt), _ip(col), val.ctypes.data_as(C.c_void_p),
           x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p))
        return y

    def ans_check(self, ref, ans):
        ref = np.ascontiguousarray(ref, dtype=self.real)
        ans = np.ascontiguousarray(ans, dtype=self.real)
        return int(self.lib.orc_ans_check(ref.ctypes.data_as(C.c_void_p),
                                          ans.ctypes.data_as(C.c_void_p), len(ref)))

    # ---- SpGEMM -------------------------------------------------------------
    def nprod(self, arpt, acol, brpt):
        M = len(arpt) - 1
        rp = np.empty(M, dtype=np.int32)
        tot, mx = C.c_longlong(), C.c_int()
        self.lib.orc_spgemm_nprod(M, _ip(arpt), _ip(acol), _ip(brpt), _ip(rp),
                                  C.byref(tot), C.byref(mx))
        return rp, tot.value, mx.value

    def bin_hist_ref(self, n, mn, mmin):
        bins = np.zeros(7, dtype=np.int32)
        n = np.ascontiguousarray(n, dtype=np.int32)
        self.lib.orc_bin_hist_ref(len(n), _ip(n), mn, mmin, _ip(bins))
        return bins

    def bin_hist_thr(self, n, thr):
        thr = np.ascontiguousarray(thr, dtype=np.int32)
        bins = np.zeros(len(thr) + 1, dtype=np.int32)
        n = np.ascontiguousarray(n, dtype=np.int32)
        self.lib.orc_bin_hist_thr(len(n), _ip(n), len(thr), _ip(thr), _ip(bins))
        return bins

    def spgemm(self, A, B):
        """A, B: dicts with M, N, rpt, col, val.  Returns C dict (+ row_nz)."""
        M, Nc = A["M"], B["N"]
        row_nz = np.empty(M, dtype=np.int32)
        crpt = np.empty(M + 1, dtype=np.int32)
        nnz = self.lib.orc_spgemm_symbolic(M, Nc, _ip(A["rpt"]), _ip(A["col"]),
                                           _ip(B["rpt"]), _ip(B["col"]), _ip(row_nz), _ip(crpt))
        ccol = np.empty(max(nnz, 1), dtype=np.int32)
        cval = np.empty(max(nnz, 1), dtype=self.real)
        av = np.ascontiguousarray(A["val"], dtype=self.real)
        bv = np.ascontiguousarray(B["val"], dtype=self.real)
        self.lib.orc_spgemm_numeric(M, Nc, _ip(A["rpt"]), _ip(A["col"]), av.ctypes.data_as(C.c_void_p),
                                    _ip(B["rpt"]), _ip(B["col"]), bv.ctypes.data_as(C.c_void_p),
                                    _ip(crpt), _ip(ccol), cval.ctypes.data_as(C.c_void_p))
        return dict(M=M, N=Nc, nnz=nnz, rpt=crpt, col=ccol[:nnz], val=cval[:nnz], row_nz=row_nz)

    def spgemm_omp(self, A, B):
        M, Nc = A["M"], B["N"]
        crpt = np.empty(M + 1, dtype=np.int32)
        pc, pv = C.POINTER(C.c_int)(), C.c_void_p()
        av = np.ascontiguousarray(A["val"], dtype=self.real)
        bv = np.ascontiguousarray(B["val"], dtype=self.real)
        nnz = self.lib.orc_spgemm_omp(M, Nc, _ip(A["rpt"]), _ip(A["col"]), av.ctypes.data_as(C.c_void_p),
                                      _ip(B["rpt"]), _ip(B["col"]), bv.ctypes.data_as(C.c_void_p),
                                      _ip(crpt), C.byref(pc), C.byref(pv))
        col = np.ctypeslib.as_array(pc, (max(nnz, 1),))[:nnz].copy()
        vb = (C.c_byte * (max(nnz, 1) * self.real().itemsize)).from_address(pv.value)
        val = np.frombuffer(vb, dtype=self.real)[:nnz].copy()
        self.lib.orc_free(C.cast(pc, C.c_void_p))
        self.lib.orc_free(pv)
        return dict(M=M, N=Nc, nnz=nnz, rpt=crpt, col=col, val=val)

    def spgemm_omp_timed(self, A, B, reps=3, threads=0):
        """C = A B on all host cores, timed inside C (orc_spgemm_omp_timed: warm thread pool and output arrays, no
        Python copies in the time).  Returns (result dict, best seconds, mean seconds, threads)."""
        M, Nc = A["M"], B["N"]
        crpt = np.empty(M + 1, dtype=np.int32)
        pc, pv = C.POINTER(C.c_int)(), C.c_void_p()
        av = np.ascontiguousarray(A["val"], dtype=self.real)
        bv = np.ascontiguousarray(B["val"], dtype=self.real)
        best, mean, nth = C.c_double(), C.c_double(), C.c_int(int(threads))
        fn = self.lib.orc_spgemm_omp_timed
        fn.restype = C.c_int
        nnz = fn(M, Nc, _ip(A["rpt"]), _ip(A["col"]), av.ctypes.data_as(C.c_void_p), _ip(B["rpt"]), _ip(B["col"]),
                 bv.ctypes.data_as(C.c_void_p), _ip(crpt), C.byref(pc), C.byref(pv), int(reps), C.byref(best),
                 C.byref(mean), C.byref(nth))
        col = np.ctypeslib.as_array(pc, (max(nnz, 1),))[:nnz].copy()
        vb = (C.c_byte * (max(nnz, 1) * self.real().itemsize)).from_address(pv.value)
        val = np.frombuffer(vb, dtype=self.real)[:nnz].copy()
        self.lib.orc_free(C.cast(pc, C.c_void_p))
        self.lib.orc_free(pv)
        return dict(M=M, N=Nc, nnz=nnz, rpt=crpt, col=col, val=val), best.value, mean.value, nth.value

    def check_spgemm(self, c, ans):
        cv = np.ascontiguousarray(c["val"], dtype=self.real)
        av = np.ascontiguousarray(ans["val"], dtype=self.real)
        return int(self.lib.orc_check_spgemm(
            ans["M"], int(c["nnz"]), _ip(np.ascontiguousarray(c["rpt"], dtype=np.int32)),
            _ip(np.ascontiguousarray(c["col"], dtype=np.int32)), cv.ctypes.data_as(C.c_void_p),
            int(ans["nnz"]), _ip(np.ascontiguousarray(ans["rpt"], dtype=np.int32)),
            _ip(np.ascontiguousarray(ans["col"], dtype=np.int32)), av.ctypes.data_as(C.c_void_p)))

    # ---- AMB ----------------------------------------------------------------
    def csr2amb(self, A, seg_size=65536, block_size=1, chunk=32, sigma=32768):
        st = _AMB()
        v = np.ascontiguousarray(A["val"], dtype=self.real)
        self.lib.orc_csr2amb(A["M"], A["N"], _ip(A["rpt"]), _ip(A["col"]),
                             v.ctypes.data_as(C.c_void_p), C.c_longlong(seg_size), block_size,
                             chunk, sigma, C.byref(st))
        return AMBResult(self, st)

    def amb_plan_model(self, A, chunk=32, sigma=32768):
        seg, bs, by = C.c_longlong(), C.c_int(), C.c_longlong()
        v = np.ascontiguousarray(A["val"], dtype=self.real)
        self.lib.orc_amb_plan_model(A["M"], A["N"], _ip(A["rpt"]), _ip(A["col"]),
                                    v.ctypes.data_as(C.c_void_p), chunk, sigma,
                                    C.byref(seg), C.byref(bs), C.byref(by))
        return seg.value, bs.value, by.value
