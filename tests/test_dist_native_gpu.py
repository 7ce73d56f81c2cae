"""libnsparse_dist_{d,s}.so on one GPU (include/nsparse_dist.h): a 1-rank RCCL communicator end to end (the
all-gather is a real ncclAllGather call), the hipGraph replay, the native timing loop, the ranks of a ragged
partition run one after the other (no communicator), and the gap-closing kernel by itself.  More than one
rank per GPU is refused by RCCL, so the multi-rank collective itself is covered by the gloo tests of the
partition logic (test_dist_cpu.py) and by the driver's multi-GPU run."""
import ctypes as C

import numpy as np
import pytest

import nsparse_amd as ns
from gpu_util import synth

pytestmark = pytest.mark.gpu


def _block(lib, dl, A, b, e):
    full = lib.csr_from_numpy(A["rpt"], A["col"], A["val"], A["N"])
    blk = ns.sfCSR()
    assert dl.nsparse_dist_csr_row_block(C.byref(full), int(b), int(e), C.byref(blk)) == 0
    lib.csr_memcpy(C.byref(blk))
    return blk


def _vec(lib, arr):
    p = lib.dmalloc(arr.nbytes)
    lib.h2d(p, arr)
    return p


@pytest.mark.parametrize("prec", ["d", "s"])
@pytest.mark.parametrize("kind,dims", [(0, (6, 6, 20)), (3, (12, 8, 0))])
def test_one_rank_rccl_end_to_end(prec, kind, dims, oracle_d, oracle_s):
    lib, dl, orc = ns.load(prec), ns.load_dist(prec), (oracle_d if prec == "d" else oracle_s)
    A = synth(lib, kind, *dims, seed=7)
    M, N = A["M"], A["N"]
    ident = C.create_string_buffer(ns.DIST_ID_BYTES)
    assert dl.nsparse_dist_unique_id(ident) == 0
    h = C.c_void_p()
    assert dl.nsparse_dist_init(C.byref(h), ident, 0, 1) == 0
    cuts = dl.partition_nnz(A["rpt"], 1)
    assert list(cuts) == [0, M]
    blk = _block(lib, dl, A, 0, M)
    x = np.random.default_rng(3).random(N + 20).astype(lib.real)
    d_x = _vec(lib, x)
    plan = ns.sfPlan()
    lib.init_plan(C.byref(plan))
    assert dl.nsparse_dist_spmv_setup(h, C.byref(blk), cuts.ctypes.data_as(ns.capi.c_int_p), d_x, C.byref(plan)) == 0
    assert plan.isPlan == 1 and dl.nsparse_dist_plan(h).contents.seg_size == plan.seg_size
    ny = int(dl.nsparse_dist_y_elems(h))
    assert ny >= M
    d_y = _vec(lib, np.full(ny + 64, 7.0, dtype=lib.real))
    ref = orc.csr_spmv(A["rpt"], A["col"], A["val"], x[:N])

    def check():
        assert dl.nsparse_dist_sync(h) == 0
        y = lib.d2h(d_y, (ny + 64,), lib.real)
        assert orc.ans_check(ref, y[:M]) == 0
        assert (y[ny:] == 7.0).all()
        lib.h2d(d_y, np.full(ny + 64, 7.0, dtype=lib.real))

    assert dl.nsparse_dist_spmv(h, d_y, d_x, 1) == 0          # kernel + ncclAllGather
    check()
    assert dl.nsparse_dist_spmv(h, d_y, d_x, 0) == 0          # local rows only
    check()
    assert dl.nsparse_dist_capture(h, d_y, d_x, 1) == 0       # the same sequence as a hipGraph
    lib.h2d(d_y, np.full(ny + 64, 7.0, dtype=lib.real))
    assert dl.nsparse_dist_spmv(h, d_y, d_x, 1) == 0          # one hipGraphLaunch
    check()
    ms_w, ms_e, us = C.c_double(), C.c_double(), C.c_double()
    assert dl.nsparse_dist_spmv_loop(h, d_y, d_x, 1, 20, C.byref(ms_w), C.byref(ms_e), C.byref(us)) == 0
    check()
    assert 0 < ms_e.value <= ms_w.value * 1.5 and us.value > 0
    print(f"[dist] {prec} kind {kind}: {ms_w.value * 1e3:.1f} us per SpMV (events {ms_e.value * 1e3:.1f}), host {us.value:.1f} us")
    dl.nsparse_dist_destroy(h)
    lib.release_csr(blk)
    lib.release_cpu_csr(blk)
    lib.dfree(d_x)
    lib.dfree(d_y)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_ranks_of_a_ragged_partition_one_after_the_other(world, oracle_d):
    """nnz-balanced cuts of a power-law matrix: every rank's handle (no communicator) computes its rows into the
    shared y at cuts[rank]; together they are the whole product."""
    lib, dl = ns.load("d"), ns.load_dist("d")
    A = synth(lib, 3, 13, 8, 0, seed=11)
    M, N = A["M"], A["N"]
    cuts = dl.partition_nnz(A["rpt"], world)
    assert cuts[0] == 0 and cuts[-1] == M and all(c % 64 == 0 for c in cuts[:-1])
    x = np.random.default_rng(4).random(N + 20)
    d_x = _vec(lib, x)
    d_y = _vec(lib, np.full(M + 64, 7.0))
    keep = []
    for r in range(world):
        h = C.c_void_p()
        assert dl.nsparse_dist_init(C.byref(h), None, r, world) == 0
        blk = _block(lib, dl, A, cuts[r], cuts[r + 1])
        plan = ns.sfPlan()
        lib.init_plan(C.byref(plan))
        assert dl.nsparse_dist_spmv_setup(h, C.byref(blk), cuts.ctypes.data_as(ns.capi.c_int_p), d_x, C.byref(plan)) == 0
        assert dl.nsparse_dist_spmv(h, d_y, d_x, 0) == 0
        assert dl.nsparse_dist_spmv(h, d_y, d_x, 1) == -4  # no communicator: the gather is refused
        assert dl.nsparse_dist_sync(h) == 0
        keep.append((h, blk))
    y = lib.d2h(d_y, (M + 64,), np.float64)
    assert oracle_d.ans_check(oracle_d.csr_spmv(A["rpt"], A["col"], A["val"], x[:N]), y[:M]) == 0
    assert (y[M:] == 7.0).all()
    for h, blk in keep:
        dl.nsparse_dist_destroy(h)
        lib.release_csr(blk)
        lib.release_cpu_csr(blk)
    lib.dfree(d_x)
    lib.dfree(d_y)


def test_close_gaps_kernel():
    lib, dl = ns.load("d"), ns.load_dist("d")
    rng = np.random.default_rng(9)
    for world, M in ((2, 1000), (5, 70001), (8, 64)):
        inner = np.sort(rng.integers(0, M + 1, world - 1))
        cuts = np.concatenate([[0], inner, [M]]).astype(np.int32)
        rpr = max(1, int(np.diff(cuts).max()))
        staged = rng.random(world * rpr)
        want = np.concatenate([staged[r * rpr:r * rpr + cuts[r + 1] - cuts[r]] for r in range(world)])
        d_s, d_c = _vec(lib, staged), _vec(lib, cuts)
        d_y = _vec(lib, np.full(M + 8, -1.0))
        assert dl.nsparse_dist_close_gaps(d_y, d_s, d_c, world, rpr, M, None) == 0
        lib.hip.hipDeviceSynchronize()
        y = lib.d2h(d_y, (M + 8,), np.float64)
        assert np.array_equal(y[:M], want) and (y[M:] == -1.0).all()
        for p in (d_s, d_c, d_y):
            lib.dfree(p)
