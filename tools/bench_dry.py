"""NSPARSE_BENCH_DRYRUN=1: bench.py's whole control flow on a box WITHOUT a GPU -- a test of the harness, never a
measurement (the JSON line says so: "dry_run").

What runs for real: argument handling, the spawn of the rank processes (--gpus N), the rendezvous, the partition
helpers, the host side of the library (generator, loader, csr_kernel, the answer checks), the byte models, the
sub-process plumbing of the `configs` block (tools/bench_config.py answers with a canned record under the same
switch), the deadline handling and the assembly + json.dumps of the line.  What is replaced: everything that needs
the device.  "Device memory" is host memory, spgemm_kernel_hash is scipy's A @ B with made-up phase times,
the row-sharded SpMV is scipy's A @ x, barriers and reductions among the ranks go through the rendezvous socket,
the PMC passes return a canned counter table with the real kernel-name shapes.  Times are small fixed numbers: the
VALUES in a dry line mean nothing, its KEYS and TYPES are what tests/test_bench_dry_cpu.py checks.

Nothing here is imported by the measuring path: bench.py reads NSPARSE_BENCH_DRYRUN once and only then imports this.
"""
import ctypes as C
import time

import numpy as np

import nsparse_amd as ns


def _obj(x):
    """what a C.byref(...) / pointer argument of the real binding refers to"""
    return getattr(x, "_obj", x)


class DryHip:
    """hipMalloc & co over host memory: a 'device pointer' is the address of a ctypes buffer kept alive here."""

    def __init__(self):
        self.blocks = {}

    def hipMalloc(self, pp, nbytes):
        buf = C.create_string_buffer(max(int(nbytes), 1))
        addr = C.addressof(buf)
        self.blocks[addr] = buf
        _obj(pp).value = addr
        return 0

    def hipFree(self, p):
        self.blocks.pop(getattr(p, "value", p), None)
        return 0

    def hipMemcpy(self, dst, src, nbytes, kind):
        C.memmove(dst, src, int(nbytes))
        return 0

    def hipMemset(self, p, v, nbytes):
        C.memset(p, v, int(nbytes))
        return 0

    def hipDeviceSynchronize(self):
        return 0

    def hipSetDevice(self, d):
        return 0


def _csr(m, real):
    import scipy.sparse as sp
    rpt = np.ctypeslib.as_array(m.rpt, (m.M + 1,))
    nnz = int(rpt[-1])
    col = np.ctypeslib.as_array(m.col, (max(nnz, 1),))[:nnz]
    val = np.frombuffer((C.c_byte * (max(nnz, 1) * np.dtype(real).itemsize)).from_address(m.val), dtype=real)[:nnz]
    return sp.csr_matrix((val, col, rpt), shape=(m.M, m.N))


class DryLib(ns.Lib):
    """The product library with its device entry points replaced (host entry points are the real ones)."""
    dry = True

    def __init__(self, precision):
        super().__init__(precision)
        for name in list(vars(self)):  # the binding sets every entry point as an instance attribute: ours win
            if name in ns.SIGNATURES and name in DryLib.__dict__:
                delattr(self, name)
        self.hip = DryHip()
        self._last = None
        self._bin_timing = 0
        self._keep = {}
        for name in ("nsparse_set_workspace_cache", "nsparse_trim_workspace", "nsparse_set_profiling"):
            setattr(self, name, lambda *a: None)

    # -- device mirrors of a CSR: the host arrays themselves
    def csr_memcpy(self, pm):
        m = _obj(pm)
        m.d_rpt, m.d_col, m.d_val = C.cast(m.rpt, C.c_void_p), C.cast(m.col, C.c_void_p), C.c_void_p(m.val)

    def release_csr(self, m):
        self._keep.pop(getattr(m, "_dry_id", None), None)

    def get_spgemm_flop(self, pa, pb, M, pflop):
        a, b = _obj(pa), _obj(pb)
        blen = np.diff(np.ctypeslib.as_array(b.rpt, (b.M + 1,))).astype(np.int64)
        acol = np.ctypeslib.as_array(a.col, (max(a.nnz, 1),))[:a.nnz]
        _obj(pflop).value = int(2 * blen[acol].sum())

    def spgemm_kernel_hash(self, pa, pb, pc):
        a, b, c = _obj(pa), _obj(pb), _obj(pc)
        A, B = _csr(a, self.real), _csr(b, self.real)
        key = (a.rpt and C.addressof(a.rpt.contents), b.rpt and C.addressof(b.rpt.contents))
        if self._last is None or self._last[0] != key:
            Cm = (A @ B).tocsr()
            Cm.sort_indices()
            blen = np.diff(B.indptr).astype(np.int64)
            row_prod = np.add.reduceat(np.r_[blen[A.indices], 0], A.indptr[:-1])[:A.shape[0]] * (np.diff(A.indptr) > 0)
            self._last = (key, Cm, int(row_prod.sum()), int(row_prod.max(initial=0)))
        _, Cm, n_prod, max_prod = self._last
        rpt = np.ascontiguousarray(Cm.indptr, dtype=np.int32)
        col = np.ascontiguousarray(Cm.indices, dtype=np.int32)
        val = np.ascontiguousarray(Cm.data, dtype=self.real)
        c.M, c.N, c.nnz = A.shape[0], B.shape[1], int(rpt[-1])
        c.nnz_max = int(np.diff(rpt).max(initial=0))
        c.d_rpt, c.d_col, c.d_val = rpt.ctypes.data, col.ctypes.data, val.ctypes.data
        c._dry_id = id(rpt)
        self._keep[c._dry_id] = (rpt, col, val)
        self._stats = (n_prod, c.nnz, max_prod, c.nnz_max, A.shape[0])
        time.sleep(0.0005)

    def nsparse_get_spgemm_stats(self, pst):
        st = _obj(pst)
        n_prod, nnz_c, max_prod, max_nnz, M = self._stats
        st.n_prod, st.nnz_c, st.max_prod_row, st.max_nnz_row = n_prod, nnz_c, max_prod, max_nnz
        st.twin_rows = M // 3 * 2
        for i in range(12):
            st.sym_bin_size[i] = st.num_bin_size[i] = 0
            st.ms_sym_bin[i] = st.ms_num_bin[i] = 0.0
        st.sym_bin_size[6] = st.num_bin_size[6] = M
        if self._bin_timing:
            st.ms_sym_bin[6], st.ms_num_bin[6] = 0.05, 0.2
        st.ms_setup, st.ms_symbolic, st.ms_numeric, st.ms_total = 0.04, 0.06, 0.21, 0.33

    def nsparse_set_bin_timing(self, on):
        old, self._bin_timing = self._bin_timing, int(on)
        return old

    def nsparse_get_spgemm_bins(self, sym, num):
        # the shape of the library's ladders (spgemm/setup.h): six hash rungs, then the window spans
        for i, v in enumerate((32, 870, 1740, 6960, 27840, 1 << 30, 4096, 16384, 65536, 8192, 32768, 0, 0, 0, 0, 0, 0, 0)):
            sym[i] = v
        for i, v in enumerate((16, 170, 682, 2730, 5461, 1 << 30, 1536, 4096, 12288, 65536, 0, 0, 0, 0, 0, 0, 0, 0)):
            num[i] = v

    def nsparse_amb_footprint_bytes(self, pamb):
        return int(getattr(_obj(pamb), "_dry_footprint", 0))


class _Handle:
    pass


class DryDist:
    """include/nsparse_dist.h without a device: the collectives go through the rendezvous (attach it first)."""
    dry = True

    def __init__(self, lib):
        self.lib, self.rdv, self.err = lib, None, 0
        real = ns.load_dist(lib.precision)  # the real library must still load and export everything
        self.partition_nnz, self.partition_work = real.partition_nnz, real.partition_work
        self.path = real.path

    def attach(self, rdv):
        self.rdv = rdv

    def nsparse_dist_device_count(self):
        return 64  # (a dry run never refuses a world size)

    def nsparse_dist_set_timeout(self, s):
        return 60.0

    def nsparse_dist_unique_id(self, buf):
        buf.raw = bytes(range(128))
        return 0

    def nsparse_dist_init(self, ph, idb, rank, world):
        h = _Handle()
        h.rank, h.world, h.have_id, h.csr, h.graph = rank, world, idb is not None, None, False
        self._h = h
        _obj(ph).value = 1
        return 0

    def nsparse_dist_last_error(self):
        return self.err

    def nsparse_dist_barrier(self, h):
        self.rdv.barrier("dry barrier")
        return 0

    def nsparse_dist_allreduce_f64(self, h, vals, n, op):
        out = self.rdv.allreduce([vals[i] for i in range(n)], "sum" if op == 0 else "max", "dry allreduce")
        for i in range(n):
            vals[i] = out[i]
        return 0

    def nsparse_dist_spmv_setup(self, h, pcsr, cuts, d_x, pplan):
        hh, m, plan = self._h, _obj(pcsr), _obj(pplan)
        hh.A = _csr(m, self.lib.real)
        hh.cuts = [int(cuts[i]) for i in range(hh.world + 1)]
        plan.isPlan, plan.seg_size, plan.seg_num, plan.block_size, plan.thread_block = 1, 65536, 1, 1, 256
        hh.plan = plan
        amb = ns.sfAMB()
        amb.chunk, amb.seg_num, amb.block_size, amb.M, amb.N = 64, max(1, -(-m.N // 65536)), 1, m.M, m.N
        w = np.dtype(self.lib.real).itemsize
        amb._dry_footprint = int(hh.A.nnz * (w + 2) + m.M * (w + 8) + m.N * w)
        hh.amb = amb
        return 0

    def nsparse_dist_y_elems(self, h):
        hh = self._h
        return max(hh.cuts[-1], hh.world * max(1, max(b - a for a, b in zip(hh.cuts, hh.cuts[1:]))))

    def _spmv(self, d_y, d_x):
        hh = self._h
        w = np.dtype(self.lib.real).itemsize
        x = np.frombuffer((C.c_byte * (hh.A.shape[1] * w)).from_address(d_x.value), dtype=self.lib.real)
        y = np.ascontiguousarray(hh.A @ x, dtype=self.lib.real)
        if y.size:
            C.memmove(d_y.value + hh.cuts[hh.rank] * w, y.ctypes.data, y.nbytes)

    def nsparse_dist_spmv(self, h, d_y, d_x, gather):
        self._spmv(d_y, d_x)
        return 0

    def nsparse_dist_sync(self, h):
        return 0

    def nsparse_dist_capture(self, h, d_y, d_x, gather):
        return 0

    def nsparse_dist_spmv_loop(self, h, d_y, d_x, gather, iters, mw, me, us):
        self._spmv(d_y, d_x)
        for ref, v in ((mw, 0.013), (me, 0.012), (us, 2.5)):
            if ref is not None:
                _obj(ref).value = v
        return 0

    def nsparse_dist_amb(self, h):
        class _P:
            contents = self._h.amb
        return _P()

    def nsparse_dist_stream(self, h):
        return None

    def nsparse_dist_close_gaps(self, d_y, staged, d_cuts, world, rpr, M, stream):
        w = np.dtype(self.lib.real).itemsize
        cuts = np.frombuffer((C.c_byte * (4 * (world + 1))).from_address(d_cuts.value), dtype=np.int32)
        for r in range(world):
            n = int(cuts[r + 1] - cuts[r])
            if n > 0:
                C.memmove(d_y.value + int(cuts[r]) * w, staged.value + r * rpr * w, n * w)
        return 0

    def nsparse_dist_release_matrix(self, h):
        self._h.A = None
        return 0

    def nsparse_dist_destroy(self, h):
        return None


def pmc_traffic(workload):
    """the shape of bench.pmc_traffic's result: kernel names as rocprofv3 prints them"""
    names = ["void nsp::spgemm::k_num_block<128, 1536, 1, 2, true>(int const*, int const*, double const*)",
             "void nsp::spgemm::k_sym_dense<256, 4096>(int const*, int const*)",
             "void nsp::spmv::k_spmv_amb_row<1, false>(double*, double const*)",
             "void nsp::spmv::k_spmv_amb_row<1, true>(double*, double const*)"]
    return {k: {"fetch_bytes": 2.0e8, "write_bytes": 3.0e7, "hbm_bytes": 2.3e8} for k in names}


CANNED_CONFIG = {
    "baseline_config": None, "dtype": "f64", "library": "libnsparse_d.so", "M": 1000, "nnz_A": 27000, "n_prod": 729000,
    "nnz_C": 125000, "steps": 3, "ms": 1.0, "ms_first_call": 2.0, "gflops": 1.458,
    "phase_ms": {"setup": 0.1, "symbolic": 0.3, "numeric": 0.6}, "sym_bin_rows": [0] * 11, "num_bin_rows": [0] * 11,
    "sym_bins_ms": [0.0] * 11, "num_bins_ms": [0.0] * 11,
    "dominant": {"phase": "numeric", "bin": 1, "kernel": "k_num_tb / k_num_lean <64, 256>", "ms": 0.5},
    "roofline": {"bound": "hbm", "peak": 8000.0, "unit": "GB/s", "compulsory_bytes": 2148000, "achieved": 2.1, "frac": 0.0003,
                 "model": "(4+w)(nnz A + nnz B + nnz C) + 12 M: every array of the call once"},
    "gen_s": 0.0, "structure_check": {"against": "dry run", "nnz_equal": True, "rpt_equal": True},
}
