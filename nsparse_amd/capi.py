"""ctypes view of the C-ABI in include/nsparse.h.

This is the binding a Python caller of the reference would write against nsparse.h; it is
plumbing for tests/ and bench.py, not the product.  The product is libnsparse_{d,s}.so.
The library must exist: there is NO fallback of any kind (no CPU path, no oracle import).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.environ.get("NSPARSE_LIB_DIR") or os.path.join(_HERE, "lib")  # override: A/B builds

c_int_p = C.POINTER(C.c_int)
c_uint_p = C.POINTER(C.c_uint)
c_ushort_p = C.POINTER(C.c_ushort)


class sfPlan(C.Structure):
    _fields_ = [("thread_grid", C.c_size_t), ("thread_block", C.c_size_t), ("isPlan", C.c_int),
                ("SIGMA", C.c_int), ("seg_size", C.c_size_t), ("seg_num", C.c_size_t),
                ("block_size", C.c_int)]


class sfCSR(C.Structure):
    _fields_ = [("rpt", c_int_p), ("col", c_int_p), ("val", C.c_void_p),
                ("d_rpt", C.c_void_p), ("d_col", C.c_void_p), ("d_val", C.c_void_p),
                ("M", C.c_int), ("N", C.c_int), ("nnz", C.c_int), ("nnz_max", C.c_int),
                ("matrix_name", C.c_char_p)]


class sfAMB(C.Structure):
    _fields_ = [("cs", c_int_p), ("cl", c_uint_p), ("sellcs_col", c_ushort_p), ("sellcs_val", C.c_void_p),
                ("s_write_permutation", c_ushort_p), ("s_write_permutation_offset", c_ushort_p),
                ("write_permutation", c_int_p),
                ("d_cs", C.c_void_p), ("d_cl", C.c_void_p), ("d_sellcs_col", C.c_void_p),
                ("d_sellcs_val", C.c_void_p), ("d_s_write_permutation", C.c_void_p),
                ("d_s_write_permutation_offset", C.c_void_p), ("d_write_permutation", C.c_void_p),
                ("block_size", C.c_int), ("nnz", C.c_int), ("M", C.c_int), ("N", C.c_int),
                ("pad_M", C.c_int), ("chunk", C.c_int), ("SIGMA", C.c_int), ("group_num_col", C.c_int),
                ("nnz_max", C.c_int), ("c_size", C.c_int), ("seg_size", C.c_size_t),
                ("seg_num", C.c_size_t), ("matrix_name", C.c_char_p)]


class SpgemmStats(C.Structure):
    _fields_ = [("n_prod", C.c_longlong), ("nnz_c", C.c_longlong), ("max_prod_row", C.c_int),
                ("max_nnz_row", C.c_int), ("sym_bin_size", C.c_int * 12), ("num_bin_size", C.c_int * 12),
                ("sym_fail_rows", C.c_int), ("ms_setup", C.c_float), ("ms_symbolic", C.c_float),
                ("ms_numeric", C.c_float), ("ms_total", C.c_float), ("ms_sym_bin", C.c_float * 12),
                ("ms_num_bin", C.c_float * 12), ("twin_rows", C.c_int)]


# every entry point declared in include/nsparse.h: name -> (restype, argtypes)
_P = C.POINTER
SIGNATURES = {
    "init_vector": (None, [C.c_void_p, C.c_int]),
    "init_csr_matrix_from_file": (None, [_P(sfCSR), C.c_char_p]),
    "csr_memcpy": (None, [_P(sfCSR)]),
    "csr_memcpyDtH": (None, [_P(sfCSR)]),
    "release_cpu_csr": (None, [sfCSR]),
    "release_cpu_amb": (None, [sfAMB]),
    "release_csr": (None, [sfCSR]),
    "release_amb": (None, [sfAMB]),
    "init_plan": (None, [_P(sfPlan)]),
    "set_plan": (None, [_P(sfPlan), C.c_size_t, C.c_int]),
    "sf_csr2amb": (None, [_P(sfAMB), _P(sfCSR), C.c_void_p, _P(sfPlan)]),
    "csr_kernel": (None, [C.c_void_p, _P(sfCSR), C.c_void_p]),
    "ans_check": (None, [C.c_void_p, C.c_void_p, C.c_int]),
    "sf_spmv_amb": (None, [C.c_void_p, _P(sfAMB), C.c_void_p, _P(sfPlan)]),
    "get_spgemm_flop": (None, [_P(sfCSR), _P(sfCSR), C.c_int, _P(C.c_longlong)]),
    "check_spgemm_answer": (None, [sfCSR, sfCSR]),
    "spgemm_kernel_hash": (None, [_P(sfCSR), _P(sfCSR), _P(sfCSR)]),
    "nsparse_last_error": (C.c_int, []),
    "nsparse_last_error_string": (C.c_char_p, []),
    "nsparse_build_info": (C.c_char_p, []),
    "nsparse_ans_check_count": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "nsparse_check_spgemm_count": (C.c_int, [_P(sfCSR), _P(sfCSR)]),
    "nsparse_init_vector_seeded": (None, [C.c_void_p, C.c_int, C.c_ulonglong]),
    "nsparse_set_amb_chunk": (C.c_int, [C.c_int]),
    "nsparse_amb_footprint_bytes": (C.c_longlong, [_P(sfAMB)]),
    "nsparse_spgemm_hash_numeric": (None, [_P(sfCSR), _P(sfCSR), _P(sfCSR)]),
    "nsparse_get_spgemm_stats": (None, [_P(SpgemmStats)]),
    "nsparse_spgemm_set_sorted": (C.c_int, [C.c_int]),
    "nsparse_set_deterministic": (C.c_int, [C.c_int]),
    "nsparse_trace_ranges": (C.c_int, []),
    "nsparse_get_spgemm_bins": (None, [c_int_p, c_int_p]),
    "nsparse_fused_state": (C.c_int, [c_int_p, c_int_p]),
    "nsparse_set_profiling": (None, [C.c_int]),
    "nsparse_set_bin_timing": (C.c_int, [C.c_int]),
    "nsparse_set_workspace_cache": (None, [C.c_int]),
    "nsparse_trim_workspace": (None, []),
    "nsparse_last_spmv_ms": (C.c_float, []),
    "nsparse_spmv_amb_async": (None, [C.c_void_p, _P(sfAMB), C.c_void_p, _P(sfPlan), C.c_void_p]),
    "nsparse_save_csr_bin": (C.c_int, [_P(sfCSR), C.c_char_p]),
    "nsparse_load_csr_bin": (C.c_int, [_P(sfCSR), C.c_char_p]),
    "nsparse_save_plan": (C.c_int, [_P(sfPlan), C.c_char_p]),
    "nsparse_load_plan": (C.c_int, [_P(sfPlan), C.c_char_p]),
    "nsparse_write_mtx": (C.c_int, [_P(sfCSR), C.c_char_p, C.c_int]),
    "nsparse_synth_csr": (None, [_P(sfCSR), C.c_int, C.c_longlong, C.c_longlong, C.c_longlong,
                                 C.c_ulonglong, C.c_longlong, C.c_longlong]),
}

# every entry point declared in include/nsparse_vendor.h (libnsparse_vendor_{d,s}.so: rocSPARSE baseline)
VENDOR_SIGNATURES = {
    "nsparse_vendor_spgemm": (None, [_P(sfCSR), _P(sfCSR), _P(sfCSR), _P(C.c_float)]),
    "nsparse_vendor_release_csr": (None, [sfCSR]),
    "spgemm_cu_csr": (None, [_P(sfCSR), _P(sfCSR), _P(sfCSR)]),
    "nsparse_vendor_spmv_csr": (C.c_float, [C.c_void_p, _P(sfCSR), C.c_void_p, C.c_int]),
    "nsparse_vendor_last_error": (C.c_int, []),
}

# every entry point declared in include/nsparse_dist.h (libnsparse_dist_{d,s}.so: row-sharded SpMV over RCCL)
c_double_p = C.POINTER(C.c_double)
DIST_SIGNATURES = {
    "nsparse_dist_partition_nnz": (C.c_int, [c_int_p, C.c_int, C.c_int, C.c_int, c_int_p]),
    "nsparse_dist_partition_work": (C.c_int, [C.POINTER(C.c_longlong), C.c_int, C.c_int, C.c_int, c_int_p]),
    "nsparse_dist_csr_row_block": (C.c_int, [_P(sfCSR), C.c_int, C.c_int, _P(sfCSR)]),
    "nsparse_dist_unique_id": (C.c_int, [C.c_char_p]),
    "nsparse_dist_init": (C.c_int, [_P(C.c_void_p), C.c_char_p, C.c_int, C.c_int]),
    "nsparse_dist_init_all": (C.c_int, [_P(C.c_void_p), C.c_int]),
    "nsparse_dist_destroy": (None, [C.c_void_p]),
    "nsparse_dist_spmv_setup": (C.c_int, [C.c_void_p, _P(sfCSR), c_int_p, C.c_void_p, _P(sfPlan)]),
    "nsparse_dist_y_elems": (C.c_longlong, [C.c_void_p]),
    "nsparse_dist_spmv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "nsparse_dist_capture": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "nsparse_dist_sync": (C.c_int, [C.c_void_p]),
    "nsparse_dist_close_gaps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "nsparse_dist_spmv_loop": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, c_double_p, c_double_p,
                                         c_double_p]),
    "nsparse_dist_amb": (_P(sfAMB), [C.c_void_p]),
    "nsparse_dist_plan": (_P(sfPlan), [C.c_void_p]),
    "nsparse_dist_stream": (C.c_void_p, [C.c_void_p]),
    "nsparse_dist_last_error": (C.c_int, []),
    "nsparse_dist_device_count": (C.c_int, []),
    "nsparse_dist_set_timeout": (C.c_double, [C.c_double]),
    "nsparse_dist_barrier": (C.c_int, [C.c_void_p]),
    "nsparse_dist_allreduce_f64": (C.c_int, [C.c_void_p, c_double_p, C.c_int, C.c_int]),
    "nsparse_dist_release_matrix": (C.c_int, [C.c_void_p]),
    "nsparse_dist_spgemm_row_work": (C.c_int, [_P(sfCSR), _P(sfCSR), C.POINTER(C.c_longlong)]),
    "nsparse_dist_spgemm": (C.c_int, [C.c_void_p, _P(sfCSR), _P(sfCSR), _P(sfCSR)]),
    "nsparse_dist_spgemm_gather": (C.c_int, [C.c_void_p, c_int_p, _P(sfCSR), _P(sfCSR)]),
    "nsparse_dist_release_gathered": (None, [sfCSR]),
}
DIST_ID_BYTES = 128

_libs = {}
_vendor = {}
_dist = {}


class Lib:
    """One precision build of the library ('d' = double, 's' = float)."""

    def __init__(self, precision):
        assert precision in ("d", "s")
        path = os.path.join(LIB_DIR, f"libnsparse_{precision}.so")
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C nsparse_amd/csrc`.  There is no fallback path.")
        self.precision = precision
        self.real = np.float64 if precision == "d" else np.float32
        self.path = path
        self.dll = C.CDLL(path, mode=C.RTLD_LOCAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(self.dll, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)
        # HIP runtime entry points used for plain device buffers in tests / bench.  They are
        # looked up through the library's own handle (dlsym searches its dependencies), so
        # they come from the SAME libamdhip64.so.7 the library is bound to: the system ROCm
        # one in a plain process, torch's bundled one (same SONAME) when torch was imported
        # first.  Never dlopen a second HIP runtime by path.
        self.hip = self.dll
        self.hip.hipMalloc.argtypes = [_P(C.c_void_p), C.c_size_t]
        self.hip.hipFree.argtypes = [C.c_void_p]
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.hip.hipDeviceSynchronize.argtypes = []

    # ---- small helpers -----------------------------------------------------------
    def csr_from_numpy(self, rpt, col, val, N, name=b"numpy"):
        """sfCSR whose HOST pointers alias numpy arrays (kept alive on the struct)."""
        rpt = np.ascontiguousarray(rpt, dtype=np.int32)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=self.real)
        m = sfCSR()
        m.rpt = rpt.ctypes.data_as(c_int_p)
        m.col = col.ctypes.data_as(c_int_p)
        m.val = val.ctypes.data_as(C.c_void_p)
        m.M = len(rpt) - 1
        m.N = int(N)
        m.nnz = int(rpt[-1])
        m.nnz_max = int(np.diff(rpt).max()) if len(rpt) > 1 else 0
        m.matrix_name = name
        m._keep = (rpt, col, val)
        return m

    def csr_host_to_numpy(self, m, copy=True):
        rpt = np.ctypeslib.as_array(m.rpt, (m.M + 1,))
        col = np.ctypeslib.as_array(m.col, (max(m.nnz, 1),))[:m.nnz]
        vb = (C.c_byte * (max(m.nnz, 1) * self.real().itemsize)).from_address(m.val)
        val = np.frombuffer(vb, dtype=self.real)[:m.nnz]
        if copy:
            rpt, col, val = rpt.copy(), col.copy(), val.copy()
        return dict(M=m.M, N=m.N, nnz=m.nnz, nnz_max=m.nnz_max, rpt=rpt, col=col, val=val)

    def dmalloc(self, nbytes):
        p = C.c_void_p()
        rc = self.hip.hipMalloc(C.byref(p), max(int(nbytes), 1))
        if rc != 0:
            raise RuntimeError(f"hipMalloc({nbytes}) -> {rc}")
        return p

    def dfree(self, p):
        self.hip.hipFree(p)

    def h2d(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        rc = self.hip.hipMemcpy(dptr, arr.ctypes.data_as(C.c_void_p), arr.nbytes, 1)
        assert rc == 0, rc

    def d2h(self, dptr, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        if out.nbytes:
            rc = self.hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), dptr, out.nbytes, 2)
            assert rc == 0, rc
        return out

    def amb_to_numpy(self, a):
        """Device arrays of an sfAMB -> numpy (for bit-exact comparison with the oracle)."""
        cs, ch, bs, n = a.c_size, a.chunk, a.block_size, a.nnz
        return dict(
            cs=self.d2h(a.d_cs, (cs,), np.int32), cl=self.d2h(a.d_cl, (cs,), np.uint32),
            sellcs_col=self.d2h(a.d_sellcs_col, (n // bs,), np.uint16),
            sellcs_val=self.d2h(a.d_sellcs_val, (n,), self.real),
            s_write_permutation=self.d2h(a.d_s_write_permutation, (cs * ch,), np.uint16),
            s_write_permutation_offset=self.d2h(a.d_s_write_permutation_offset, (cs,), np.uint16),
            write_permutation=self.d2h(a.d_write_permutation, (cs * ch,), np.int32),
            c_size=cs, chunk=ch, block_size=bs, nnz=n, pad_M=a.pad_M, seg_size=a.seg_size,
            seg_num=a.seg_num, M=a.M, N=a.N)


class VendorLib:
    """libnsparse_vendor_{d,s}.so: the reference's cuSPARSE comparison path on rocSPARSE (baseline and
    third oracle; the product library does not depend on it)."""

    def __init__(self, precision):
        assert precision in ("d", "s")
        path = os.path.join(LIB_DIR, f"libnsparse_vendor_{precision}.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it with `make -C nsparse_amd/csrc vendor`")
        self.precision = precision
        self.path = path
        self.dll = C.CDLL(path, mode=C.RTLD_LOCAL)
        for name, (res, args) in VENDOR_SIGNATURES.items():
            fn = getattr(self.dll, name)
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)


class DistLib:
    """libnsparse_dist_{d,s}.so: the native row-sharded multi-GPU SpMV (include/nsparse_dist.h).  It links the
    product library of the same precision (one instance per process: the loader finds the copy load() mapped)
    and RCCL."""

    def __init__(self, precision):
        assert precision in ("d", "s")
        load(precision)  # the product library first, so that both handles name one mapped object
        path = os.path.join(LIB_DIR, f"libnsparse_dist_{precision}.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it with `make -C nsparse_amd/csrc dist`")
        self.precision = precision
        self.path = path
        self.dll = C.CDLL(path, mode=C.RTLD_LOCAL)
        for name, (res, args) in DIST_SIGNATURES.items():
            fn = getattr(self.dll, name)
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)

    def partition_nnz(self, rpt, world, align=64):
        rpt = np.ascontiguousarray(rpt, dtype=np.int32)
        cuts = np.zeros(world + 1, dtype=np.int32)
        rc = self.nsparse_dist_partition_nnz(rpt.ctypes.data_as(c_int_p), len(rpt) - 1, world, align,
                                             cuts.ctypes.data_as(c_int_p))
        assert rc == 0, rc
        return cuts

    def partition_work(self, work, world, align=1):
        work = np.ascontiguousarray(work, dtype=np.int64)
        cuts = np.zeros(world + 1, dtype=np.int32)
        rc = self.nsparse_dist_partition_work(work.ctypes.data_as(C.POINTER(C.c_longlong)), len(work), world, align,
                                              cuts.ctypes.data_as(c_int_p))
        assert rc == 0, rc
        return cuts


def load_dist(precision="d"):
    if precision not in _dist:
        _dist[precision] = DistLib(precision)
    return _dist[precision]


def load_vendor(precision="d"):
    if precision not in _vendor:
        _vendor[precision] = VendorLib(precision)
    return _vendor[precision]


def load(precision="d"):
    if precision not in _libs:
        _libs[precision] = Lib(precision)
    return _libs[precision]
