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
    from nsparse_amd.dist import ShardedSpMV, csr_row_block, row_partition
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


def test_row_partition_properties():
    from nsparse_amd.dist import csr_row_block, row_partition
    for M in (1, 63, 64, 65, 62451, 3542400):
        for P in (1, 2, 4, 8):
            rpr, blocks = row_partition(M, P)
            assert rpr % 64 == 0 and rpr * P >= M
            assert blocks[0][0] == 0 and blocks[-1][1] == M
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(P - 1))
            assert all(0 <= e - b <= rpr for b, e in blocks)
    g = load_golden("banded2k")
    _, blocks = row_partition(g["M"], 4)
    parts = [csr_row_block(g, b, e) for b, e in blocks]
    assert sum(p["nnz"] for p in parts) == g["nnz"]
    assert np.array_equal(np.concatenate([p["col"] for p in parts]), g["col"])
