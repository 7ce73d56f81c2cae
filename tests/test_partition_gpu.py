"""SpGEMM by 1-D row blocks (stretch row), on ONE GPU playing every rank in turn: the
blocks C[rows_r, :] = A[rows_r, :] * B, stitched, must be the single-call product bit for bit in
rpt / col -- including the M < K set-up path (k_col_range: B-row records only for the stretch of B
the block's columns reach) -- and row-sharded AMB SpMV blocks must concatenate to the full y."""
import ctypes as C

import numpy as np
import pytest

from gpu_util import DeviceAMB, spgemm, synth
from nsparse_amd.dist import csr_row_block, row_partition_nnz, row_partition_work, row_products
from dist_driver import ShardedSpGEMM, make_gpu_local_spgemm

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,p,world", [(0, (9, 9, 40), 4), (5, (9, 9, 40), 3), (3, (13, 8, 0), 8),
                                          (4, (50000, 160000, 0), 2)])
def test_row_blocks_stitch_to_the_full_product(kind, p, world, lib_d, oracle_d):
    A = synth(lib_d, kind, *p, seed=0x5EED0022)
    full, st = spgemm(lib_d, A)
    local = make_gpu_local_spgemm(lib_d)
    rpt = np.zeros(A["M"] + 1, dtype=np.int64)
    cols, vals, prods = [], [], []
    blocks = row_partition_work(row_products(A, A["rpt"]), world)
    for r in range(world):
        op = ShardedSpGEMM(A, A, r, world, local, blocks=blocks)
        assert op.A_block["M"] < A["M"]  # every block takes the M < K set-up (k_col_range)
        c = op()
        b, e = blocks[r]
        rpt[b + 1:e + 1] = c["rpt"][1:] + rpt[b]
        cols.append(c["col"])
        vals.append(c["val"])
        prods.append(int(row_products(op.A_block, A["rpt"]).sum()))
    assert np.array_equal(rpt, full["rpt"]), "stitched C.rpt != single-call C.rpt"
    assert np.array_equal(np.concatenate(cols), full["col"]), "stitched C.col != single-call C.col"
    stitched = dict(full, val=np.concatenate(vals))
    assert oracle_d.check_spgemm(stitched, full) == 0
    assert sum(prods) == st.n_prod and max(prods) <= 1.35 * st.n_prod / world + st.max_prod_row


def test_row_sharded_amb_spmv_blocks(lib_d, oracle_d):
    """nnz-balanced row blocks, each converted to AMB against the full x on its own: the
    concatenation is y, and an empty block (more ranks than 64-row chunks) is a valid no-op."""
    A = synth(lib_d, 4, 40000, 125000, 0, seed=7)
    x = np.random.default_rng(3).random(A["N"])
    y_ref = oracle_d.csr_spmv(A["rpt"], A["col"], A["val"], x)
    for world in (3, 8):
        blocks = row_partition_nnz(A["rpt"], world)
        parts = []
        for b, e in blocks:
            blk = csr_row_block(A, b, e)
            d = DeviceAMB(lib_d, blk)
            parts.append(d.spmv(x))
            d.close()
        assert oracle_d.ans_check(y_ref, np.concatenate(parts)) == 0
    tiny = csr_row_block(A, 0, 100)
    blocks = row_partition_nnz(tiny["rpt"], 8)
    assert any(e == b for b, e in blocks)
    parts = []
    for b, e in blocks:
        d = DeviceAMB(lib_d, csr_row_block(tiny, b, e))
        assert d.amb.c_size >= 0
        parts.append(d.spmv(x))
        d.close()
    assert oracle_d.ans_check(y_ref[:100], np.concatenate(parts)) == 0


@pytest.mark.parametrize("prec", ["d", "s"])
@pytest.mark.parametrize("kind,p,world", [(5, (9, 9, 40), 3), (3, (13, 8, 0), 8)])
def test_native_row_partitioned_spgemm(prec, kind, p, world, oracle_d, oracle_s):
    """The C-ABI driver of the row-partitioned SpGEMM (include/nsparse_dist.h, round 4): per-row work from the host
    arrays = the numpy rule, product-balanced cuts, every rank's block through nsparse_dist_spgemm (one GPU playing
    the ranks in turn, handles without a communicator), stitched = the single-call product; and at world 1, with a
    real one-rank RCCL communicator, nsparse_dist_spgemm_gather assembles that product on the device."""
    import nsparse_amd as ns
    lib, dl, orc = ns.load(prec), ns.load_dist(prec), (oracle_d if prec == "d" else oracle_s)
    A = synth(lib, kind, *p, seed=0x5EED0022)
    full, st = spgemm(lib, A)
    host = lib.csr_from_numpy(A["rpt"], A["col"], A["val"], A["N"])
    work = np.zeros(A["M"], dtype=np.int64)
    assert dl.nsparse_dist_spgemm_row_work(C.byref(host), C.byref(host), work.ctypes.data_as(C.POINTER(C.c_longlong))) == 0
    assert np.array_equal(work, row_products(A, A["rpt"])) and int(work.sum()) == st.n_prod
    cuts = dl.partition_work(work, world)
    assert [(int(cuts[r]), int(cuts[r + 1])) for r in range(world)] == row_partition_work(work, world)
    b_dev = lib.csr_from_numpy(A["rpt"], A["col"], A["val"], A["N"])
    lib.csr_memcpy(C.byref(b_dev))
    rpt = np.zeros(A["M"] + 1, dtype=np.int64)
    cols, vals = [], []
    for r in range(world):
        h = C.c_void_p()
        assert dl.nsparse_dist_init(C.byref(h), None, r, world) == 0
        blk = ns.sfCSR()
        assert dl.nsparse_dist_csr_row_block(C.byref(host), int(cuts[r]), int(cuts[r + 1]), C.byref(blk)) == 0
        lib.csr_memcpy(C.byref(blk))
        c = ns.sfCSR()
        assert dl.nsparse_dist_spgemm(h, C.byref(blk), C.byref(b_dev), C.byref(c)) == 0
        assert c.M == cuts[r + 1] - cuts[r]
        lib.csr_memcpyDtH(C.byref(c))
        got = lib.csr_host_to_numpy(c)
        lib.release_cpu_csr(c)
        # without a communicator a gather among several ranks is refused, not attempted
        if world > 1:
            cf = ns.sfCSR()
            assert dl.nsparse_dist_spgemm_gather(h, cuts.ctypes.data_as(ns.capi.c_int_p), C.byref(c), C.byref(cf)) == -4
        lib.release_csr(c)
        lib.release_csr(blk)
        lib.release_cpu_csr(blk)
        dl.nsparse_dist_destroy(h)
        b, e = int(cuts[r]), int(cuts[r + 1])
        rpt[b + 1:e + 1] = got["rpt"][1:] + rpt[b]
        cols.append(got["col"])
        vals.append(got["val"])
    assert np.array_equal(rpt, full["rpt"]) and np.array_equal(np.concatenate(cols), full["col"])
    assert orc.check_spgemm(dict(full, val=np.concatenate(vals)), full) == 0
    # world 1 with a real communicator: the gather path end to end (copies instead of broadcasts, the same offsets)
    ident = C.create_string_buffer(ns.DIST_ID_BYTES)
    assert dl.nsparse_dist_unique_id(ident) == 0
    h = C.c_void_p()
    assert dl.nsparse_dist_init(C.byref(h), ident, 0, 1) == 0
    a_dev = lib.csr_from_numpy(A["rpt"], A["col"], A["val"], A["N"])
    lib.csr_memcpy(C.byref(a_dev))
    c, cf = ns.sfCSR(), ns.sfCSR()
    assert dl.nsparse_dist_spgemm(h, C.byref(a_dev), C.byref(b_dev), C.byref(c)) == 0
    one = np.array([0, A["M"]], dtype=np.int32)
    assert dl.nsparse_dist_spgemm_gather(h, one.ctypes.data_as(ns.capi.c_int_p), C.byref(c), C.byref(cf)) == 0
    assert cf.M == A["M"] and cf.nnz == full["nnz"]
    assert np.array_equal(lib.d2h(cf.d_rpt, (cf.M + 1,), np.int32), full["rpt"])
    assert np.array_equal(lib.d2h(cf.d_col, (cf.nnz,), np.int32), full["col"])
    assert orc.check_spgemm(dict(full, val=lib.d2h(cf.d_val, (cf.nnz,), lib.real)), full) == 0
    dl.nsparse_dist_release_gathered(cf)
    lib.release_csr(c)
    lib.release_csr(a_dev)
    lib.release_csr(b_dev)
    dl.nsparse_dist_destroy(h)
