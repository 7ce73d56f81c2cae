#!/usr/bin/env python3
"""Generate tests/golden/*.npz: known-answer vectors from an INDEPENDENT implementation
(scipy.sparse), used to pin oracle/nsparse_oracle.c and, through it, the HIP path.

The reference (EBD-CREST/nsparse) holds one fixture, data/test.mtx, and no expected
outputs; it cannot be compiled here (CUDA-only).  So the vectors are made with scipy:
  y      = A @ x                    (CSR order summation, same as the reference's csr_kernel)
  C      = A @ A with sorted indices (what cuSPARSE csrgemm, the reference's own SpGEMM
           oracle, returns: structural product, ascending columns)
  C_pat  = pattern product (all-ones values) -> structure that is immune to scipy's
           dropping of exact-zero sums.
Run from the repo root:  python tools/make_golden.py
"""
import os

import numpy as np
import scipy.sparse as sp

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def banded(M, half_bw, rng, signed=False):
    rows, cols = [], []
    for d in range(-half_bw, half_bw + 1):
        i = np.arange(max(0, -d), min(M, M - d))
        keep = rng.random(len(i)) < 0.8 if d != 0 else np.ones(len(i), bool)
        rows.append(i[keep])
        cols.append(i[keep] + d)
    r, c = np.concatenate(rows), np.concatenate(cols)
    v = rng.random(len(r)) + 0.1
    if signed:
        v *= rng.choice([-1.0, 1.0], len(r))
    return sp.csr_matrix((v, (r, c)), shape=(M, M))


def rmat(scale, ef, rng, a=0.57, b=0.19, c=0.19):
    n = 1 << scale
    m = n * ef
    r = np.zeros(m, np.int64)
    cidx = np.zeros(m, np.int64)
    for bit in range(scale):
        u = rng.random(m)
        rb = u >= a + b
        cb = ((u >= a) & (u < a + b)) | (u >= a + b + c)
        r |= rb.astype(np.int64) << bit
        cidx |= cb.astype(np.int64) << bit
    v = rng.random(m) + 0.1
    A = sp.coo_matrix((v, (r, cidx)), shape=(n, n)).tocsr()  # duplicates summed
    A.sort_indices()
    return A


def pack(name, A, rng):
    A = A.tocsr()
    A.sort_indices()
    A.sum_duplicates()
    M, N = A.shape
    x = rng.random(N)
    y = A @ x
    Cm = (A @ A).tocsr()
    Cm.sort_indices()
    P = A.copy()
    P.data[:] = 1.0
    Cp = (P @ P).tocsr()
    Cp.sort_indices()
    assert Cp.nnz == Cm.nnz, "scipy dropped a cancelled entry; change the seed"
    assert np.array_equal(Cp.indices, Cm.indices) and np.array_equal(Cp.indptr, Cm.indptr)
    row_prod = np.asarray(P @ np.diff(P.indptr).astype(np.float64)).astype(np.int64)
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        M=M, N=N, rpt=A.indptr.astype(np.int32), col=A.indices.astype(np.int32),
        val=A.data.astype(np.float64), x=x, y=y,
        c_rpt=Cm.indptr.astype(np.int32), c_col=Cm.indices.astype(np.int32),
        c_val=Cm.data.astype(np.float64), row_prod=row_prod.astype(np.int32))
    print(name, "M", M, "nnz", A.nnz, "n_prod", int(row_prod.sum()), "nnzC", Cm.nnz)


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(0x5EED0001)
    pack("banded2k", banded(2048, 6, rng), rng)
    pack("banded_signed1k", banded(1000, 9, rng, signed=True), rng)
    pack("rmat_s10", rmat(10, 6, rng), rng)
    # wide matrix: N > 65536 so that AMB needs several column segments
    M, N = 1500, 150000
    r = np.repeat(np.arange(M), 12)
    c = rng.integers(0, N, len(r))
    v = rng.random(len(r)) + 0.1
    W = sp.coo_matrix((v, (r, c)), shape=(M, N)).tocsr()
    W.sum_duplicates()
    W.sort_indices()
    xw = rng.random(N)
    np.savez_compressed(os.path.join(OUT, "wide_seg.npz"), M=M, N=N,
                        rpt=W.indptr.astype(np.int32), col=W.indices.astype(np.int32),
                        val=W.data.astype(np.float64), x=xw, y=W @ xw)
    print("wide_seg", W.shape, W.nnz)


if __name__ == "__main__":
    main()
