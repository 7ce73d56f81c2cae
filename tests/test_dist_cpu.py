"""World-size-2 test of the row-sharded SpMV driver on CPU (gloo).

The GPU kernel cannot run here, so the per-rank compute is injected (a numpy CSR product on the
rank's row block); what is under test is everything around it: the row partition, the row-block
extraction, the all-gather layout and that every rank ends with the full, rank-count-independent y."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from nsparse_amd.dist import csr_row_block, row_partition
    from dist_driver import ShardedSpMV
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = load_golden(name)
    M = g["M"]
    _, blocks = row_partition(M, world)
    blk = csr_row_block(g, *blocks[rank])

    def local_spmv(x_full, y_out):
        x = x_full.numpy()
        y = np.zeros(blk["M"])
        for i in range(blk["M"]):
            lo, hi = blk["rpt"][i], blk["rpt"][i + 1]
            y[i] = np.dot(blk["val"][lo:hi], x[blk["col"][lo:hi]])
        y_out[:blk["M"]] = torch.from_numpy(y)

    op = ShardedSpMV(M, rank, world, local_spmv, lambda n: torch.zeros(n, dtype=torch.float64),
                     lambda o, i: dist.all_gather_into_tensor(o, i))
    x = torch.from_numpy(g["x"].copy())
    y = op(x).clone()  # op returns a view of its gather buffer, reused by the next call
    # a second product with y fed back as x (iteration): every rank must hold the full vector
    if g["N"] == M:
        y2 = op(y.clone())
        out[rank] = (y.numpy().copy(), y2.numpy().copy())
    else:
        out[rank] = (y.numpy().copy(), None)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["banded2k", "wide_seg"])
def test_row_sharded_spmv_world2(name):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, name, out), nprocs=world, join=True)
        res = dict(out)
    g = load_golden(name)
    for r in range(world):
        np.testing.assert_allclose(res[r][0], g["y"], rtol=1e-12)
    assert np.array_equal(res[0][0], res[1][0])
    if res[0][1] is not None:
        import scipy.sparse as sp
        A = sp.csr_matrix((g["val"], g["col"], g["rpt"]), shape=(g["M"], g["N"]))
        np.testing.assert_allclose(res[0][1], A @ g["y"], rtol=1e-11)
        assert np.array_equal(res[0][1], res[1][1])


def _worker_ragged(rank, world, port, name, out):
    """nnz-balanced (unequal) row blocks: padded all-gather + one concatenation."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import scipy.sparse as sp
    import torch
    import torch.distributed as dist
    from nsparse_amd.dist import csr_row_block, row_partition_nnz
    from dist_driver import ShardedSpMV
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = load_golden(name)
    blocks = row_partition_nnz(g["rpt"], world, align=64)
    blk = csr_row_block(g, *blocks[rank])
    Ab = sp.csr_matrix((blk["val"], blk["col"], blk["rpt"]), shape=(blk["M"], g["N"]))

    def local_spmv(x_full, y_out):
        y_out[:blk["M"]] = torch.from_numpy(Ab @ x_full.numpy())

    op = ShardedSpMV(g["M"], rank, world, local_spmv, lambda n: torch.zeros(n, dtype=torch.float64),
                     lambda o, i: dist.all_gather_into_tensor(o, i), blocks=blocks, compact=torch.cat)
    y = op(torch.from_numpy(g["x"].copy())).clone()
    out[rank] = (y.numpy().copy(), blocks, op.ragged)
    dist.barrier()
    dist.destroy_process_group()


def test_nnz_balanced_row_shards_world2():
    """R-MAT rows are skewed: the nnz-balanced cut is far from M/2, the blocks are unequal, and every
    rank must still end with the full y in row order."""
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker_ragged, args=(world, port, "rmat_s10", out), nprocs=world, join=True)
        res = dict(out)
    g = load_golden("rmat_s10")
    blocks = res[0][1]
    assert res[0][2], "expected unequal blocks"
    nnz_blocks = [int(g["rpt"][e] - g["rpt"][b]) for b, e in blocks]
    assert max(nnz_blocks) < 0.6 * g["nnz"] and blocks[0][1] % 64 == 0 and blocks[0][1] != g["M"] // 2
    for r in range(world):
        np.testing.assert_allclose(res[r][0], g["y"], rtol=1e-12)


def _worker_spgemm(rank, world, port, name, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import scipy.sparse as sp
    import torch.distributed as dist
    from dist_driver import ShardedSpGEMM
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = load_golden(name)
    A = dict(M=g["M"], N=g["N"], rpt=g["rpt"], col=g["col"], val=g["val"])

    def local_spgemm(Ab, B):  # the rank's product, injected (the GPU call cannot run here)
        a = sp.csr_matrix((Ab["val"], Ab["col"], Ab["rpt"]), shape=(Ab["M"], B["M"]))
        b = sp.csr_matrix((B["val"], B["col"], B["rpt"]), shape=(B["M"], B["N"]))
        c = (a @ b).tocsr()
        c.sort_indices()
        return dict(rpt=c.indptr.astype(np.int32), col=c.indices.astype(np.int32), val=c.data)

    op = ShardedSpGEMM(A, A, rank, world, local_spgemm)
    c_loc = op()
    full = op.gather(c_loc)
    out[rank] = (full["rpt"], full["col"], full["val"], op.blocks, c_loc["nnz"])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["banded2k", "rmat_s10"])
def test_row_partitioned_spgemm_world2(name):
    """SpGEMM by 1-D row blocks of A balanced by products, B replicated: the blocks
    C[rows_r, :] stitched together are the golden C = A^2 -- rpt and col exactly, on every rank."""
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker_spgemm, args=(world, port, name, out), nprocs=world, join=True)
        res = dict(out)
    g = load_golden(name)
    for r in range(world):
        rpt, col, val, blocks, nnz_loc = res[r]
        assert np.array_equal(rpt, g["c_rpt"]) and np.array_equal(col, g["c_col"])
        np.testing.assert_allclose(val, g["c_val"], rtol=1e-9, atol=1e-12)
    # balanced by products: neither rank carries more than 60 % of them
    prod = g["row_prod"].astype(np.int64)
    share = [int(prod[b:e].sum()) for b, e in res[0][3]]
    assert max(share) <= 0.6 * sum(share)
    assert res[0][4] + res[1][4] == len(g["c_col"])


def test_row_partition_properties():
    from nsparse_amd.dist import csr_row_block, row_partition
    for M in (1, 63, 64, 65, 62451, 3542400):
        for P in (1, 2, 4, 8):
            rpr, blocks = row_partition(M, P)
            assert rpr % 64 == 0 and rpr * P >= M
            assert blocks[0][0] == 0 and blocks[-1][1] == M
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(P - 1))
            assert all(0 <= e - b <= rpr for b, e in blocks)
    from nsparse_amd.dist import row_partition_nnz, row_partition_work, row_products
    for name in ("banded2k", "rmat_s10", "wide_seg"):
        g = load_golden(name)
        for P in (1, 2, 3, 8):
            blocks = row_partition_nnz(g["rpt"], P)
            assert blocks[0][0] == 0 and blocks[-1][1] == g["M"]
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(P - 1))
            assert all(b % 64 == 0 for b, _ in blocks)
            nn = [int(g["rpt"][e] - g["rpt"][b]) for b, e in blocks]
            # no block exceeds its share by more than the rows one alignment step can hold
            step = int(np.diff(g["rpt"]).max()) * 64
            assert max(nn) <= g["nnz"] / P + step
        if g["M"] == g["N"]:
            assert np.array_equal(row_products(g, g["rpt"]), g["row_prod"])
            wb = row_partition_work(g["row_prod"], 4)
            assert wb[0][0] == 0 and wb[-1][1] == g["M"]
    # a matrix with fewer rows than ranks * align: trailing blocks are empty, never negative
    tiny = np.arange(0, 3 * 40 + 1, 3)
    bl = row_partition_nnz(tiny, 8)
    assert bl[-1][1] == 40 and all(e >= b for b, e in bl)
    g = load_golden("banded2k")
    _, blocks = row_partition(g["M"], 4)
    parts = [csr_row_block(g, b, e) for b, e in blocks]
    assert sum(p["nnz"] for p in parts) == g["nnz"]
    assert np.array_equal(np.concatenate([p["col"] for p in parts]), g["col"])
