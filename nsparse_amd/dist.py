"""Row-sharded SpMV across the GPUs of one node: one process per GPU, RCCL all-gather of y.

Not in the reference (single GPU).  North-star design: 1-D row blocks, every
rank converts ITS row block to AMB against the full x (replicated, N*w bytes), computes
y_local with the same single-GPU kernel, then one all-gather puts the full y on every rank for
the next iteration.  The y shards are disjoint, so there is no cross-GPU reduction and the
result does not depend on the number of ranks (bit-identical to the 1-GPU run when the matrix
has one column segment).

xGMI is point to point; the y shard of a rank is small (M/P*w bytes, 3.5 MB for nlpkkt120 at
P=8), so ONE all_gather per SpMV with the whole shard as the message is used -- no bucketing,
nothing to pipeline against: the collective needs the finished y_local.

torch is plumbing here: device buffers, the current stream, and torch.distributed (backend
"nccl" is RCCL on ROCm; "gloo" for the CPU tests).  The SpMV itself is the C-ABI library.
"""
import ctypes as C

import numpy as np


def row_partition(M, world_size, align=64):
    """Equal row blocks, each a multiple of `align` rows (AMB chunk), last one ragged.
    Returns (rows_per_rank, [(begin, end)] * world_size); blocks past M are empty."""
    rpr = -(-M // world_size)
    rpr = -(-rpr // align) * align
    return rpr, [(min(r * rpr, M), min((r + 1) * rpr, M)) for r in range(world_size)]


def csr_row_block(A, begin, end):
    """Rows [begin, end) of a host CSR dict as a CSR dict over the same columns."""
    lo, hi = int(A["rpt"][begin]), int(A["rpt"][end])
    return dict(M=end - begin, N=A["N"], rpt=(A["rpt"][begin:end + 1] - lo).astype(np.int32),
                col=A["col"][lo:hi], val=A["val"][lo:hi], nnz=hi - lo)


class ShardedSpMV:
    """y = A x with A row-sharded over the ranks of `group`.

    local_spmv(x_full, y_local_out) computes this rank's rows.  On a GPU box it is the AMB
    kernel launched on torch's current stream (make_gpu_local); the CPU tests inject their own.
    """

    def __init__(self, M, rank, world_size, local_spmv, make_buffer, all_gather):
        self.M, self.rank, self.world = M, rank, world_size
        self.rpr, self.blocks = row_partition(M, world_size)
        self.begin, self.end = self.blocks[rank]
        self.local_spmv = local_spmv
        self.all_gather = all_gather
        self.y_full = make_buffer(self.rpr * world_size)
        self.y_local = make_buffer(self.rpr)

    def __call__(self, x_full, gather=True):
        self.local_spmv(x_full, self.y_local)
        if gather and self.world > 1:
            self.all_gather(self.y_full, self.y_local)
            return self.y_full[:self.M]
        if self.world == 1:
            return self.y_local[:self.M]
        return self.y_local


def make_gpu_sharded_spmv(lib, A_local, M_global, rank, world_size, device, plan_args=None):
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

    op = ShardedSpMV(M_global, rank, world_size, local_spmv, make_buffer, all_gather)
    assert m_local == op.end - op.begin
    op._keep = (csr, amb, plan, x_tune)
    op.amb, op.plan, op.csr = amb, plan, csr
    return op
