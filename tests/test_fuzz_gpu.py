"""Randomised parity sweep through the C-ABI: many small A (M x K) * B (K x N) of mixed shape,
density and row-length distribution against the CPU oracle -- structure exact, values to the
reference tolerance.  Catches the corners the targeted tests do not name (empty rows / columns,
1 x N, N x 1, hub rows next to empty ones, every mix of bins in one call)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from gpu_util import oracle_fp64_accumulated, spgemm

pytestmark = pytest.mark.gpu


def _rand_csr(rng, m, n, kind):
    if kind == 0:      # uniform density
        d = rng.choice([0.0, 0.002, 0.01, 0.05, 0.3])
        a = sp.random(m, n, density=d, format="csr", random_state=rng, dtype=np.float64)
    elif kind == 1:    # power-law row lengths, hub columns
        lens = np.minimum((rng.pareto(1.2, m) * 2).astype(np.int64), n)
        rows = np.repeat(np.arange(m), lens)
        cols = np.minimum((rng.pareto(0.8, rows.size) * 3).astype(np.int64), n - 1)
        a = sp.csr_matrix((rng.random(rows.size) + 0.1, (rows, cols)), shape=(m, n))
        a.sum_duplicates()
    elif kind == 3:    # runs of rows with one column pattern (twin rows), different values
        base = sp.random(max(m // 3, 1), n, density=rng.choice([0.003, 0.02, 0.2]), format="csr",
                         random_state=rng, dtype=np.float64)
        pick = np.sort(rng.integers(0, base.shape[0], size=m))
        a = base[pick].tocsr()
        a.data = rng.random(a.data.size) + 0.1
    else:              # banded
        bw = int(rng.integers(1, 40))
        offs = [o for o in range(-bw, bw + 1) if -m < o < n]
        diags = [rng.random(max(m, n)) + 0.1 for _ in offs]
        a = sp.diags(diags, offs, shape=(m, n), format="csr")
    a.sort_indices()
    return dict(M=m, N=n, rpt=a.indptr.astype(np.int32), col=a.indices.astype(np.int32),
                val=a.data.astype(np.float64))


# NSPARSE_FUZZ_SEEDS / NSPARSE_FUZZ_BASE / NSPARSE_FUZZ_PREC=s: a longer soak from other seeds, or
# through the float build (default: 40 cases from seed 1000, double)
@pytest.mark.parametrize("seed", range(int(os.environ.get("NSPARSE_FUZZ_SEEDS", "40"))))
def test_random_products(seed, lib_d, oracle_d, lib_s, oracle_s):
    if os.environ.get("NSPARSE_FUZZ_PREC", "d") == "s":  # the float build (soak runs)
        lib_d, oracle_d = lib_s, oracle_s
    rng = np.random.default_rng(int(os.environ.get("NSPARSE_FUZZ_BASE", "1000")) + seed)
    m, k, n = (int(rng.choice([1, 2, 7, 63, 64, 65, 300, 1500, 4000, 20000])) for _ in range(3))
    A = _rand_csr(rng, m, k, int(rng.integers(0, 4)))
    B = _rand_csr(rng, k, n, int(rng.integers(0, 4)))
    A["val"], B["val"] = A["val"].astype(lib_d.real), B["val"].astype(lib_d.real)
    ref = oracle_d.spgemm(A, B)
    got, st = spgemm(lib_d, A, B)
    assert got["nnz"] == ref["nnz"]
    assert np.array_equal(got["rpt"], ref["rpt"]) and np.array_equal(got["col"], ref["col"])
    if lib_d.real == np.float32:
        # float build: the reference's 1e-6 rule against the fp64-accumulated oracle (the float oracle's
        # own CSR-order float sums are the noisier side for rows with thousands of products)
        from oracle.oracle import Oracle
        assert oracle_d.check_spgemm(got, oracle_fp64_accumulated(Oracle("d"), A, B)) == 0
    else:
        assert oracle_d.check_spgemm(got, ref) == 0
    assert sum(st.sym_bin_size) + st.twin_rows == m and sum(st.num_bin_size) == m


def _node_block_square(rng):
    """A square matrix with d unknowns per node of a random banded node graph (every unknown of a node has
    the node's column pattern: twin rows in classes of d), renumbered not at all / inside bands / globally."""
    nodes = int(rng.choice([5, 40, 300, 1500]))
    d = int(rng.choice([1, 2, 3, 4, 6]))
    bw = min(int(rng.integers(1, 30)), nodes - 1)
    g = sp.random(nodes, nodes, density=min(1.0, rng.choice([2.0, 6.0, 20.0]) / nodes), format="csr",
                  random_state=rng, dtype=np.float64)
    g = (g + sp.diags([np.ones(nodes)] * 3, [-bw, 0, bw], shape=(nodes, nodes), format="csr")).tocsr()
    a = sp.kron(g, np.ones((d, d)), format="csr")
    n = a.shape[0]
    mode = int(rng.integers(0, 3))
    perm = np.arange(n)
    if mode == 1:      # shuffled inside bands of a few nodes
        band = d * int(rng.integers(2, 12))
        for s in range(0, n, band):
            perm[s:s + band] = s + rng.permutation(min(band, n - s))
    elif mode == 2:    # anywhere
        perm = rng.permutation(n)
    a = a[perm][:, perm].tocsr()
    a.sort_indices()
    a.data = rng.random(a.data.size) + 0.1
    return dict(M=n, N=n, rpt=a.indptr.astype(np.int32), col=a.indices.astype(np.int32), val=a.data.astype(np.float64))


@pytest.mark.parametrize("seed", range(int(os.environ.get("NSPARSE_FUZZ_SQ_SEEDS", "24"))))
def test_random_node_block_squares(seed, lib_d, oracle_d, lib_s, oracle_s):
    """C = A * A with two copies of A (as the reference's sample calls it) on node-block matrices: twin rows
    by pattern in classes of 1..6, groups of three, keyed runs when the numbering scatters the classes."""
    if os.environ.get("NSPARSE_FUZZ_PREC", "d") == "s":
        lib_d, oracle_d = lib_s, oracle_s
    rng = np.random.default_rng(int(os.environ.get("NSPARSE_FUZZ_BASE", "1000")) + 7919 * seed)
    A = _node_block_square(rng)
    A["val"] = A["val"].astype(lib_d.real)
    ref = oracle_d.spgemm(A, A)
    got, st = spgemm(lib_d, A, numeric_again=True)
    assert np.array_equal(got["rpt"], ref["rpt"]) and np.array_equal(got["col"], ref["col"])
    if lib_d.real == np.float32:
        from oracle.oracle import Oracle
        assert oracle_d.check_spgemm(got, oracle_fp64_accumulated(Oracle("d"), A, A)) == 0
    else:
        assert oracle_d.check_spgemm(got, ref) == 0
    assert np.array_equal(got["col_again"], got["col"])
    np.testing.assert_allclose(got["val_again"], got["val"], rtol=1e-9 if lib_d.real == np.float64 else 2e-6)
    assert sum(st.sym_bin_size) + st.twin_rows == A["M"] and sum(st.num_bin_size) == A["M"]


@pytest.mark.parametrize("shuffled", [False, True])
@pytest.mark.parametrize("d", [2, 6])
def test_bricks_with_two_and_six_unknowns_per_node(d, shuffled, lib_d, oracle_d):
    """A 27-point brick of 5 x 5 x 12 nodes with d = 2 and d = 6 unknowns per node (the cant-class stand-in has 3): twin
    classes of d rows.  The node-block numeric kernel works on groups of at most THREE rows (a leader and its lowest
    and highest follower), so a class of 2 is one short group and a class of 6 is split -- whatever the split, all
    but one row of every node must be found as twins, and C must be the oracle's, with the natural numbering and with
    the unknowns shuffled inside bands (no twin is a neighbour then: keyed runs)."""
    nx, ny, nz = 5, 5, 12
    idx = np.arange(nx * ny * nz).reshape(nx, ny, nz)
    rows, cols = [], []
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                a = idx[max(0, -dx):nx - max(0, dx), max(0, -dy):ny - max(0, dy), max(0, -dz):nz - max(0, dz)]
                b = idx[max(0, dx):nx - max(0, -dx), max(0, dy):ny - max(0, -dy), max(0, dz):nz - max(0, -dz)]
                rows.append(a.ravel())
                cols.append(b.ravel())
    g = sp.csr_matrix((np.ones(sum(len(r) for r in rows)), (np.concatenate(rows), np.concatenate(cols))),
                      shape=(idx.size, idx.size))
    a = sp.kron(g, np.ones((d, d)), format="csr")
    n = a.shape[0]
    rng = np.random.default_rng(600 + d)
    if shuffled:
        perm = np.arange(n)
        band = 9 * d
        for s0 in range(0, n, band):
            perm[s0:s0 + band] = s0 + rng.permutation(min(band, n - s0))
        a = a[perm][:, perm].tocsr()
    a.sort_indices()
    a.data = rng.random(a.data.size) + 0.1
    A = dict(M=n, N=n, rpt=a.indptr.astype(np.int32), col=a.indices.astype(np.int32), val=a.data.astype(np.float64))
    ref = oracle_d.spgemm(A, A)
    got, st = spgemm(lib_d, A, numeric_again=True)
    assert np.array_equal(got["rpt"], ref["rpt"]) and np.array_equal(got["col"], ref["col"])
    assert oracle_d.check_spgemm(got, ref) == 0
    assert np.array_equal(got["col_again"], got["col"])
    np.testing.assert_allclose(got["val_again"], got["val"], rtol=1e-9)
    assert st.twin_rows == n - n // d, (st.twin_rows, n, d)
    assert sum(st.sym_bin_size) + st.twin_rows == n and sum(st.num_bin_size) == n


@pytest.mark.parametrize("seed", range(int(os.environ.get("NSPARSE_FUZZ_AB_SEEDS", "16"))))
def test_random_node_block_products_with_another_b(seed, lib_d, oracle_d, lib_s, oracle_s):
    """A * B on node-block matrices with B != A: a row block of A times the whole A (what a rank of the partitioned
    product computes) and A times a second node-block matrix of the same size -- the keyed runs of the node-block
    kernel then take the pattern leaders of B's rows from k_b_twins (setup.h), not from A's own map."""
    if os.environ.get("NSPARSE_FUZZ_PREC", "d") == "s":
        lib_d, oracle_d = lib_s, oracle_s
    rng = np.random.default_rng(int(os.environ.get("NSPARSE_FUZZ_BASE", "1000")) + 104729 * seed)
    A = _node_block_square(rng)
    A["val"] = A["val"].astype(lib_d.real)
    n = A["M"]
    lo = int(rng.integers(0, max(1, n // 2)))
    hi = int(rng.integers(lo + 1, n + 1))
    z0, z1 = int(A["rpt"][lo]), int(A["rpt"][hi])
    blk = dict(M=hi - lo, N=n, rpt=(A["rpt"][lo:hi + 1] - z0).astype(np.int32), col=A["col"][z0:z1], val=A["val"][z0:z1])
    # a second matrix with the same number of unknowns: the node graph of A under another renumbering and with
    # other values (so that its rows come in pattern classes too, in other places)
    perm = rng.permutation(n)
    S = sp.csr_matrix((A["val"].astype(np.float64), A["col"], A["rpt"]), shape=(n, n))[perm][:, perm].tocsr()
    S.sort_indices()
    B = dict(M=n, N=n, rpt=S.indptr.astype(np.int32), col=S.indices.astype(np.int32),
             val=(rng.random(S.data.size) + 0.1).astype(lib_d.real))
    for X, Y in ((blk, A), (A, B)):
        ref = oracle_d.spgemm(X, Y)
        got, st = spgemm(lib_d, X, Y)
        assert np.array_equal(got["rpt"], ref["rpt"]) and np.array_equal(got["col"], ref["col"])
        if lib_d.real == np.float32:
            from oracle.oracle import Oracle
            assert oracle_d.check_spgemm(got, oracle_fp64_accumulated(Oracle("d"), X, Y)) == 0
        else:
            assert oracle_d.check_spgemm(got, ref) == 0


@pytest.mark.parametrize("seed", range(int(os.environ.get("NSPARSE_FUZZ_AMB_SEEDS", "24"))))
def test_random_amb_plans(seed, lib_d, oracle_d, lib_s, oracle_s):
    """sf_csr2amb + sf_spmv_amb under random plans (segment size, block size 1..20, chunk 32 / 64, both precisions) on
    random matrices of every shape the SpGEMM sweep uses, unsorted rows included: the seven AMB arrays bit for bit
    against the oracle (convert_amb.cu:22-929), the footprint model, and y against csr_kernel by the reference's rule
    (kernel_spmv_amb.cu:21-79; nsparse.cu:261-297).  Round 5: written and first run on the CPU emulation."""
    import ctypes as C

    from gpu_util import DeviceAMB
    from test_amb_gpu import assert_same_format
    rng = np.random.default_rng(int(os.environ.get("NSPARSE_FUZZ_BASE", "1000")) + 15485863 * seed)
    lib, orc = (lib_d, oracle_d) if rng.integers(0, 2) == 0 else (lib_s, oracle_s)
    m = int(rng.choice([1, 5, 63, 64, 65, 300, 1000, 2500]))
    n = int(rng.choice([1, 7, 64, 300, 5000, 70000, 140000]))
    A = _rand_csr(rng, m, n, int(rng.choice([0, 1, 3, 4])))
    if rng.integers(0, 4) == 0 and A["rpt"][-1] > 0:  # unsorted rows: the conversion sorts internally
        col, val = A["col"].copy(), A["val"].copy()
        for i in range(m):
            lo, hi = A["rpt"][i], A["rpt"][i + 1]
            p = rng.permutation(hi - lo)
            col[lo:hi], val[lo:hi] = col[lo:hi][p], val[lo:hi][p]
        A = dict(A, col=col, val=val)
    A = dict(A, val=(A["val"] * rng.choice([-1.0, 1.0], size=A["val"].size) if rng.integers(0, 3) == 0 else A["val"]).astype(lib.real))
    seg = int(rng.choice([1, 3, 64, 300, 1024, 4096, 65536]))
    # the segment number lives in the upper 16 bits of `cl` (nsparse.h: SCL_BORDER; convert_amb.cu:313-346): a plan with
    # more than 65536 segments wraps -- in the reference, in the oracle and here alike (the arrays still agree; y is then
    # not A x).  The reference's own search never gets there (seg_size 65536, or 1..4 when N < 100); neither does this.
    seg = max(seg, -(-n // 65536))
    bs = int(rng.integers(1, 21))
    chunk = int(rng.choice([32, 64]))
    d = DeviceAMB(lib, A, seg, bs, chunk=chunk)
    try:
        srt = A
        if m and A["rpt"][-1] > 0:  # the oracle is handed sorted rows (what the conversion works on)
            S = sp.csr_matrix((A["val"], A["col"], A["rpt"]), shape=(m, n))
            S.sort_indices()
            srt = dict(A, col=S.indices.astype(np.int32), val=S.data.astype(lib.real))
        ora = orc.csr2amb(srt, int(d.plan.seg_size), int(d.plan.block_size), chunk)
        assert_same_format(d.arrays(), ora)
        assert lib.nsparse_amb_footprint_bytes(C.byref(d.amb)) == ora.footprint
        x = (rng.random(n) + 0.5).astype(lib.real)
        y = d.spmv(x)
        y_ref = orc.csr_spmv(srt["rpt"], srt["col"], srt["val"], x)
        mag = orc.csr_spmv(srt["rpt"], srt["col"], np.abs(srt["val"]), np.abs(x))
        assert (np.abs(y - y_ref) <= (1e-12 if lib.real == np.float64 else 2e-5) * (mag + 1e-300)).all()
    finally:
        d.close()
