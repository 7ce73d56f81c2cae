"""SpGEMM by 1-D row blocks (stretch row), on ONE GPU playing every rank in turn: the
blocks C[rows_r, :] = A[rows_r, :] * B, stitched, must be the single-call product bit for bit in
rpt / col -- including the M < K set-up path (k_col_range: B-row records only for the stretch of B
the block's columns reach) -- and row-sharded AMB SpMV blocks must concatenate to the full y."""
import ctypes as C

import numpy as np
import pytest

from gpu_util import DeviceAMB, spgemm, synth
from nsparse_amd.dist import (ShardedSpGEMM, csr_row_block, make_gpu_local_spgemm, row_partition_nnz,
                              row_partition_work, row_products)

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
