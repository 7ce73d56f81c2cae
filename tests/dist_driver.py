"""One process per GPU over torch.distributed: the Python driver of the row-sharded SpMV and the row-partitioned SpGEMM.
TEST INFRASTRUCTURE (tests/test_dist_cpu.py: gloo world 2 on the CPU; tests/test_partition_gpu.py): since round 3 the
measured multi-GPU path is native (libnsparse_dist: RCCL all-gather on the rank's stream, no Python in the loop), and
this module is its layout twin -- same partition, same gather layout, same gap closing -- with torch as plumbing (device
buffers, the current stream, backend "nccl" = RCCL on ROCm, "gloo" for the CPU tests).

Design it mirrors: 1-D row blocks, every rank converts ITS block to AMB against the full x (replicated),
computes y_local with the single-GPU kernel, then ONE all-gather with the whole shard as the message puts the full y on
every rank; the shards are disjoint, so there is no cross-GPU reduction and the result does not depend on the rank count.
"""
import ctypes as C

import numpy as np

from nsparse_amd.dist import csr_row_block, row_partition, row_partition_nnz, row_partition_work, row_products  # noqa: F401


class ShardedSpMV:
    """y = A x with A row-sharded over the ranks of `group`.

    local_spmv(x_full, y_local_out) computes this rank's rows.  On a GPU box it is the AMB
    kernel launched on torch's current stream (make_gpu_local); the CPU tests inject their own.
    """

    def __init__(self, M, rank, world_size, local_spmv, make_buffer, all_gather, blocks=None, compact=None):
        self.M, self.rank, self.world = M, rank, world_size
        if blocks is None:
            self.rpr, self.blocks = row_partition(M, world_size)
        else:  # e.g. row_partition_nnz: unequal blocks, the collective moves the longest one per rank
            self.blocks = [(int(b), int(e)) for b, e in blocks]
            assert self.blocks[0][0] == 0 and self.blocks[-1][1] == M
            assert all(self.blocks[r][1] == self.blocks[r + 1][0] for r in range(world_size - 1))
            self.rpr = max(1, max(e - b for b, e in self.blocks))
        self.begin, self.end = self.blocks[rank]
        self.local_spmv = local_spmv
        self.all_gather = all_gather
        self.y_full = make_buffer(self.rpr * world_size)
        self.y_local = make_buffer(self.rpr)
        # equal blocks land in place; ragged ones leave gaps that one concatenation closes
        self.ragged = any(b != r * self.rpr for r, (b, e) in enumerate(self.blocks) if e > b)
        self.compact = compact

    def __call__(self, x_full, gather=True):
        self.local_spmv(x_full, self.y_local)
        if gather and self.world > 1:
            self.all_gather(self.y_full, self.y_local)
            if self.ragged:
                return self.compact([self.y_full[r * self.rpr:r * self.rpr + (e - b)]
                                     for r, (b, e) in enumerate(self.blocks)])
            return self.y_full[:self.M]
        if self.world == 1:
            return self.y_local[:self.M]
        return self.y_local


def make_gpu_sharded_spmv(lib, A_local, M_global, rank, world_size, device, plan_args=None, blocks=None):
    """Build the GPU pipeline for this rank's row block `A_local` (host CSR dict)."""
    import torch
    import torch.distributed as dist

    import nsparse_amd as ns

    tdtype = torch.float64 if lib.precision == "d" else torch.float32
    csr = lib.csr_from_numpy(A_local["rpt"], A_local["col"], A_local["val"], A_local["N"])
    lib.csr_memcpy(C.byref(csr))
    plan = ns.sfPlan()
    if plan_args is None:
        lib.init_plan(C.byref(plan))
    else:
        lib.set_plan(C.byref(plan), *plan_args)
    x_tune = torch.zeros(A_local["N"] + 20, dtype=tdtype, device=device)
    torch.cuda.synchronize()
    amb = ns.sfAMB()
    lib.sf_csr2amb(C.byref(amb), C.byref(csr), C.c_void_p(x_tune.data_ptr()), C.byref(plan))
    m_local = A_local["M"]

    def local_spmv(x_full, y_out):
        stream = torch.cuda.current_stream().cuda_stream
        lib.nsparse_spmv_amb_async(C.c_void_p(y_out.data_ptr()), C.byref(amb),
                                   C.c_void_p(x_full.data_ptr()), C.byref(plan), C.c_void_p(stream))

    def make_buffer(n):
        return torch.zeros(n, dtype=tdtype, device=device)

    def all_gather(out, inp):
        if dist.get_backend() == "nccl":
            dist.all_gather_into_tensor(out, inp)  # RCCL, device buffers, in stream order
        else:  # smoke-test backends (gloo) gather through host memory
            torch.cuda.synchronize()
            parts = [torch.empty(inp.numel(), dtype=inp.dtype) for _ in range(world_size)]
            dist.all_gather(parts, inp.cpu())
            out.copy_(torch.cat(parts).to(out.device))

    op = ShardedSpMV(M_global, rank, world_size, local_spmv, make_buffer, all_gather, blocks=blocks,
                     compact=torch.cat)
    assert m_local == op.end - op.begin
    op._keep = (csr, amb, plan, x_tune)
    op.amb, op.plan, op.csr = amb, plan, csr
    return op


# ---------------------------------------------------------------------------------------------
# SpGEMM: 1-D row partition of A, B replicated (stretch row).  No exchange inside the
# algorithm: rank r computes C[rows_r, :] = A[rows_r, :] * B with the single-GPU call (an A with
# fewer rows than B takes the library's column-range set-up, k_col_range).  The result stays
# distributed; gather() assembles the full CSR on every rank when a caller wants it.
# ---------------------------------------------------------------------------------------------
class ShardedSpGEMM:
    """C = A B with A cut into row blocks (balanced by products), B whole on every rank.

    local_spgemm(A_block, B) -> dict(rpt, col, val) for the block; on a GPU box it is
    spgemm_kernel_hash through the C-ABI (make_gpu_local_spgemm), the CPU tests inject their own.
    """

    def __init__(self, A, B, rank, world_size, local_spgemm, blocks=None):
        self.rank, self.world = rank, world_size
        self.M, self.N = A["M"], B["N"]
        self.blocks = blocks if blocks is not None else row_partition_work(row_products(A, B["rpt"]), world_size)
        self.begin, self.end = self.blocks[rank]
        self.A_block = csr_row_block(A, self.begin, self.end)
        self.B = B
        self.local_spgemm = local_spgemm

    def __call__(self):
        c = self.local_spgemm(self.A_block, self.B)
        return dict(M=self.end - self.begin, N=self.N, rpt=np.asarray(c["rpt"]), col=np.asarray(c["col"]),
                    val=np.asarray(c["val"]), nnz=int(c["rpt"][-1]))

    def gather(self, c_local, device="cpu"):
        """Full C on every rank: one all-gather of the block sizes, then one padded all-gather each
        for rpt / col / val (all_gather_into_tensor wants equal shares; pads are cut off)."""
        import torch
        import torch.distributed as dist
        if self.world == 1:
            return c_local
        sizes = torch.zeros(2 * self.world, dtype=torch.int64, device=device)
        mine = torch.tensor([c_local["M"], c_local["nnz"]], dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(sizes, mine)
        sizes = sizes.cpu().numpy().reshape(self.world, 2)
        m_max, z_max = int(sizes[:, 0].max()), max(1, int(sizes[:, 1].max()))

        def padded_gather(arr, n, dtype):
            t = torch.zeros(n, dtype=dtype, device=device)
            t[:len(arr)] = torch.as_tensor(np.ascontiguousarray(arr), dtype=dtype, device=device)
            out = torch.empty(n * self.world, dtype=dtype, device=device)
            dist.all_gather_into_tensor(out, t)
            return out.cpu().numpy().reshape(self.world, n)

        val_t = torch.float32 if np.asarray(c_local["val"]).dtype == np.float32 else torch.float64
        rpts = padded_gather(c_local["rpt"], m_max + 1, torch.int32)
        cols = padded_gather(c_local["col"], z_max, torch.int32)
        vals = padded_gather(c_local["val"], z_max, val_t)
        rpt = np.zeros(self.M + 1, dtype=np.int32)
        col_parts, val_parts, off, row = [], [], 0, 0
        for r in range(self.world):
            m, z = int(sizes[r, 0]), int(sizes[r, 1])
            rpt[row + 1:row + m + 1] = rpts[r, 1:m + 1] + off
            col_parts.append(cols[r, :z])
            val_parts.append(vals[r, :z])
            off += z
            row += m
        assert row == self.M
        return dict(M=self.M, N=self.N, nnz=off, rpt=rpt, col=np.concatenate(col_parts), val=np.concatenate(val_parts))


def make_gpu_local_spgemm(lib):
    """local_spgemm for ShardedSpGEMM on a GPU box: csr_memcpy / spgemm_kernel_hash / csr_memcpyDtH."""
    import nsparse_amd as ns

    def local_spgemm(A_block, B):
        if A_block["M"] == 0:
            return dict(rpt=np.zeros(1, np.int32), col=np.zeros(0, np.int32), val=np.zeros(0, lib.real))
        a = lib.csr_from_numpy(A_block["rpt"], A_block["col"], A_block["val"], B["M"])
        b = lib.csr_from_numpy(B["rpt"], B["col"], B["val"], B["N"])
        c = ns.sfCSR()
        lib.csr_memcpy(C.byref(a))
        lib.csr_memcpy(C.byref(b))
        lib.spgemm_kernel_hash(C.byref(a), C.byref(b), C.byref(c))
        lib.csr_memcpyDtH(C.byref(c))
        out = lib.csr_host_to_numpy(c)
        lib.release_cpu_csr(c)
        lib.release_csr(c)
        lib.release_csr(a)
        lib.release_csr(b)
        return out

    return local_spgemm
