// amb_dist_{s,d} <file.mtx> [ngpus [seg_size block_size]]  --  y = A x with the AMB format, row-sharded over
// the GPUs of one node (include/nsparse_dist.h).  The multi-GPU twin of amb_{s,d} (reference driver
// cuda-c/src/sample/spmv/spmv_amb.cu:15-118, single GPU): same loader, same CPU check (csr_kernel +
// ans_check), same protocol (TRI_NUM runs, the first discarded).  One process, one thread per GPU
// (ncclCommInitAll): rows cut by non-zeros on 64-row boundaries, x replicated, one RCCL all-gather of y
// per SpMV.  Every rank ends up with the whole y; rank 0's copy is checked.
#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>

#include "nsparse_dist.h"

struct RankOut {
    double ms_gather = 0, ms_compute = 0, us_host = 0;
    long long footprint = 0;
    int rc = 0;
};

// The rank threads agree on success BEFORE the first collective: a rank that failed its set-up must not leave the
// others waiting for it in ncclAllGather (they would sit there until the library's watchdog gives up).
struct Agree {
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0, world = 0;
    bool failed = false;
    // every rank calls this once with its own verdict; returns true when ALL ranks are fine
    bool all_ok(bool mine_ok)
    {
        std::unique_lock<std::mutex> lk(m);
        failed |= !mine_ok;
        if (++arrived == world) cv.notify_all();
        else cv.wait(lk, [&] { return arrived == world; });
        return !failed;
    }
};

static void rank_main(int r, int world, nsparse_dist_t h, const sfCSR *full, const int *cuts, const real *x,
                      const sfPlan *plan_in, real *y_out, RankOut *out, Agree *agree)
{
    sfCSR blk = {};
    real *d_x = nullptr, *d_y = nullptr;
    bool have_blk = false;
    out->rc = hipSetDevice(r) == hipSuccess ? 0 : -1;
    if (!out->rc) out->rc = nsparse_dist_csr_row_block(full, cuts[r], cuts[r + 1], &blk);
    if (!out->rc) {
        have_blk = true;
        csr_memcpy(&blk);
        if (hipMalloc((void **)&d_x, sizeof(real) * (full->N + MAX_BLOCK_SIZE)) != hipSuccess) out->rc = -2;
    }
    if (!out->rc) {
        hipMemset(d_x, 0, sizeof(real) * (full->N + MAX_BLOCK_SIZE));
        hipMemcpy(d_x, x, sizeof(real) * full->N, hipMemcpyHostToDevice);
        sfPlan plan = *plan_in;
        out->rc = nsparse_dist_spmv_setup(h, &blk, cuts, d_x, &plan);
    }
    if (!out->rc && hipMalloc((void **)&d_y, sizeof(real) * (size_t)(nsparse_dist_y_elems(h) + WARP)) != hipSuccess) out->rc = -2;
    if (!agree->all_ok(out->rc == 0)) {  // some rank failed: nobody enters a collective
        if (!out->rc) out->rc = -100;    // (this rank was fine)
        if (d_x) hipFree(d_x);
        if (d_y) hipFree(d_y);
        if (have_blk) {
            release_csr(blk);
            release_cpu_csr(blk);
        }
        return;
    }
    hipMemset(d_y, 0, sizeof(real) * (size_t)(nsparse_dist_y_elems(h) + WARP));
    // compute only, then compute + all-gather: TRI_NUM runs each, the first discarded (spmv_amb.cu:46-58)
    out->rc = nsparse_dist_spmv_loop(h, d_y, d_x, 0, 1, nullptr, nullptr, nullptr);
    if (!out->rc) out->rc = nsparse_dist_spmv_loop(h, d_y, d_x, 0, TRI_NUM - 1, &out->ms_compute, nullptr, nullptr);
    if (!out->rc) out->rc = nsparse_dist_spmv_loop(h, d_y, d_x, 1, 1, nullptr, nullptr, nullptr);
    if (!out->rc) out->rc = nsparse_dist_spmv_loop(h, d_y, d_x, 1, TRI_NUM - 1, &out->ms_gather, nullptr, &out->us_host);
    if (blk.M > 0) out->footprint = nsparse_amb_footprint_bytes(const_cast<sfAMB *>(nsparse_dist_amb(h)));
    if (r == 0 && !out->rc) hipMemcpy(y_out, d_y, sizeof(real) * full->M, hipMemcpyDeviceToHost);
    hipFree(d_x);
    hipFree(d_y);
    release_csr(blk);
    release_cpu_csr(blk);
}

int main(int argc, char **argv)
{
    if (argc < 2) {
        fprintf(stderr, "usage: %s matrix.mtx [ngpus [seg_size block_size]]\n", argv[0]);
        return 1;
    }
    int ndev = 0;
    hipGetDeviceCount(&ndev);
    int world = argc >= 3 ? atoi(argv[2]) : ndev;
    if (world < 1 || world > ndev) {
        fprintf(stderr, "%d GPUs asked for, %d visible\n", world, ndev);
        return 1;
    }
    sfCSR mat;
    init_csr_matrix_from_file(&mat, argv[1]);
    std::vector<real> x((size_t)mat.N), y((size_t)mat.M), csr_y((size_t)mat.M);
    init_vector(x.data(), mat.N);
    csr_kernel(csr_y.data(), &mat, x.data());
    sfPlan plan;
    if (argc >= 5) set_plan(&plan, (size_t)atoi(argv[3]), atoi(argv[4]));
    else init_plan(&plan);

    std::vector<int> cuts((size_t)world + 1);
    nsparse_dist_partition_nnz(mat.rpt, mat.M, world, WARP, cuts.data());
    std::vector<nsparse_dist_t> h((size_t)world, nullptr);
    if (nsparse_dist_init_all(h.data(), world)) return 2;
    std::vector<RankOut> out((size_t)world);
    std::vector<std::thread> th;
    Agree agree;
    agree.world = world;
    for (int r = 0; r < world; r++)
        th.emplace_back(rank_main, r, world, h[r], &mat, cuts.data(), x.data(), &plan, y.data(), &out[r], &agree);
    for (auto &t : th) t.join();
    double ms_g = 0, ms_c = 0, us = 0;
    long long fp = 0;
    for (int r = 0; r < world; r++) {
        if (out[r].rc) {
            for (int q = 0; q < world; q++)
                if (out[q].rc && out[q].rc != -100) fprintf(stderr, "rank %d failed: %d\n", q, out[q].rc);
            for (int q = 0; q < world; q++) nsparse_dist_destroy(h[q]);
            return 3;
        }
        ms_g = out[r].ms_gather > ms_g ? out[r].ms_gather : ms_g;
        ms_c = out[r].ms_compute > ms_c ? out[r].ms_compute : ms_c;
        us = out[r].us_host > us ? out[r].us_host : us;
        fp += out[r].footprint;
        printf("rank %d: rows [%d, %d)  nnz %d\n", r, cuts[r], cuts[r + 1], mat.rpt[cuts[r + 1]] - mat.rpt[cuts[r]]);
    }
    printf("SpMV using AMB format on %d GPUs: %s, %f[GFLOPS], %f[ms] (compute only %f[ms], host %f[us] per SpMV)\n",
           world, mat.matrix_name, (float)((double)mat.nnz * 2 / 1000 / 1000 / ms_g), ms_g, ms_c, us);
    printf("AMB footprint: %lld bytes/SpMV => %f[GB/s]\n", fp, (float)((double)fp / 1e6 / ms_g));
    for (int r = 0; r < world; r++) nsparse_dist_destroy(h[r]);
    ans_check(csr_y.data(), y.data(), mat.M);
    release_cpu_csr(mat);
    return 0;
}
