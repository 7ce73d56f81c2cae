"""Row-partition arithmetic of the multi-GPU paths, in numpy: where the 1-D row blocks of the row-sharded
SpMV and of the row-partitioned SpGEMM are cut.

The NATIVE rule lives in libnsparse_dist (csrc/dist_spmv.hip: nsparse_dist_partition_nnz / _work, nsparse_dist_row_block);
these are its Python twins, cut for cut (tests/test_host_abi.py compares them), used by bench.py to build each rank's
synthetic row block and by the tests.  No torch, no device: the one-process-per-GPU torch.distributed driver that used to
live here (a test vehicle since round 3, when the ranks became native) is tests/dist_driver.py.
"""

import numpy as np


def row_partition(M, world_size, align=64):
    """Equal row blocks (the nnz-balanced cut of a matrix whose rows all have the same length), each a
    multiple of `align` rows (AMB chunk), last one ragged.
    Returns (rows_per_rank, [(begin, end)] * world_size); blocks past M are empty."""
    rpr = -(-M // world_size)
    rpr = -(-rpr // align) * align
    return rpr, [(min(r * rpr, M), min((r + 1) * rpr, M)) for r in range(world_size)]


def row_partition_nnz(rpt, world_size, align=64):
    """Row blocks balanced by NON-ZEROS: rank r ends at the first row boundary, rounded
    to a multiple of `align` rows (the AMB chunk), where the running nnz reaches (r+1)/P of the
    total.  Returns [(begin, end)] * world_size, contiguous, covering [0, M); blocks may be empty
    when the matrix has fewer than `align` rows per rank."""
    rpt = np.asarray(rpt, dtype=np.int64)
    M = len(rpt) - 1
    total = int(rpt[-1])
    cuts = [0]
    for r in range(1, world_size):
        row = int(np.searchsorted(rpt, total * r / world_size, side="left"))
        row = min(M, max(cuts[-1], int(round(row / align)) * align))
        cuts.append(row)
    cuts.append(M)
    return [(cuts[r], cuts[r + 1]) for r in range(world_size)]


def csr_row_block(A, begin, end):
    """Rows [begin, end) of a host CSR dict as a CSR dict over the same columns."""
    lo, hi = int(A["rpt"][begin]), int(A["rpt"][end])
    return dict(M=end - begin, N=A["N"], rpt=(A["rpt"][begin:end + 1] - lo).astype(np.int32),
                col=A["col"][lo:hi], val=A["val"][lo:hi], nnz=hi - lo)


def row_partition_work(work_per_row, world_size, align=1):
    """Row blocks balanced by per-row WORK (intermediate products: what set_intprod_num counts,
    kernel_spgemm_hash_d.cu:70-86).  Same contract as row_partition_nnz."""
    w = np.concatenate([[0], np.cumsum(np.asarray(work_per_row, dtype=np.int64))])
    return row_partition_nnz(w, world_size, align)


def row_products(A, B_rpt):
    """Intermediate products of every row of A * B (host, numpy): sum of the B row lengths it meets."""
    blen = np.diff(np.asarray(B_rpt, dtype=np.int64))
    per_entry = blen[np.asarray(A["col"], dtype=np.int64)] if len(A["col"]) else np.zeros(0, np.int64)
    out = np.zeros(A["M"], dtype=np.int64)
    rpt = np.asarray(A["rpt"], dtype=np.int64)
    nz = rpt[1:] > rpt[:-1]
    if per_entry.size:
        out[nz] = np.add.reduceat(per_entry, rpt[:-1][nz])
    return out
