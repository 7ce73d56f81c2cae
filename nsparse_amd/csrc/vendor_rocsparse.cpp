// vendor_rocsparse.cpp -- the vendor library behind the reference's comparison samples, on ROCm.
//
// Replaces (reference file:line):
//   spgemm_cu_csr / spgemm_kernel_cu_csr   cuda-c/src/kernel/kernel_spgemm_cu_csr.cu:59-203
//       (cusparseXcsrgemmNnz + cusparse{S,D}csrgemm: the answer check_spgemm_answer is held against,
//        cuda-c/src/sample/spgemm/spgemm_hash.cu:60-68, and the "vs vendor" GFLOPS line)
//   csr_ans / the cusparse{S,D}csrmv loop   cuda-c/src/sample/spmv/spmv_cu_csr.cu:13-85
//
// Built as its own shared object, libnsparse_vendor_{d,s}.so, so that libnsparse_{d,s}.so does not
// depend on rocSPARSE.  It is a BASELINE and a third oracle for the tests, never part of the product
// path.  Same sfCSR in / out as the rest of the ABI; C's device arrays are plain hipMalloc blocks,
// so release_csr of the main library (or nsparse_vendor_release_csr) frees them.
#include <hip/hip_runtime.h>
#include <rocsparse/rocsparse.h>

#include <cstdio>
#include <cstdlib>

#include "nsparse.h"

namespace {

int g_err = 0;

#define V_HIP(expr)                                                                         \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            fprintf(stderr, "nsparse vendor: %s -> %s\n", #expr, hipGetErrorString(e_));    \
            g_err = (int)e_;                                                                \
        }                                                                                   \
    } while (0)
#define V_RS(expr)                                                                          \
    do {                                                                                    \
        rocsparse_status s_ = (expr);                                                       \
        if (s_ != rocsparse_status_success) {                                               \
            fprintf(stderr, "nsparse vendor: %s -> rocsparse status %d\n", #expr, (int)s_); \
            g_err = 1000 + (int)s_;                                                         \
        }                                                                                   \
    } while (0)

rocsparse_handle handle()
{
    static rocsparse_handle h = nullptr;
    if (!h) V_RS(rocsparse_create_handle(&h));
    return h;
}

#if NSPARSE_REAL_IS_FLOAT
#define RS_CSRGEMM_BUFFER rocsparse_scsrgemm_buffer_size
#define RS_CSRGEMM rocsparse_scsrgemm
#define RS_CSRMV_ANALYSIS rocsparse_scsrmv_analysis
#define RS_CSRMV rocsparse_scsrmv
#else
#define RS_CSRGEMM_BUFFER rocsparse_dcsrgemm_buffer_size
#define RS_CSRGEMM rocsparse_dcsrgemm
#define RS_CSRMV_ANALYSIS rocsparse_dcsrmv_analysis
#define RS_CSRMV rocsparse_dcsrmv
#endif

}  // namespace

extern "C" {

int nsparse_vendor_last_error(void) { return g_err; }

// C = A * B with rocsparse csrgemm (buffer size, nnz, compute): allocates c->d_rpt / d_col / d_val,
// sets c->M, N, nnz.  Synchronous on return.  *ms (optional) = device time of the three stages,
// allocations excluded, by HIP events -- the quantity kernel_spgemm_cu_csr.cu:186-199 times.
void nsparse_vendor_spgemm(sfCSR *a, sfCSR *b, sfCSR *c, float *ms)
{
    g_err = 0;
    rocsparse_handle h = handle();
    rocsparse_mat_descr da, db, dc, dd;
    rocsparse_mat_info info;
    V_RS(rocsparse_create_mat_descr(&da));
    V_RS(rocsparse_create_mat_descr(&db));
    V_RS(rocsparse_create_mat_descr(&dc));
    V_RS(rocsparse_create_mat_descr(&dd));
    V_RS(rocsparse_create_mat_info(&info));
    const real alpha = (real)1;
    const rocsparse_int m = a->M, n = b->N, k = a->N;
    c->M = m;
    c->N = n;
    hipEvent_t e0, e1, e2, e3;
    V_HIP(hipEventCreate(&e0));
    V_HIP(hipEventCreate(&e1));
    V_HIP(hipEventCreate(&e2));
    V_HIP(hipEventCreate(&e3));
    size_t buf_bytes = 0;
    V_RS(RS_CSRGEMM_BUFFER(h, rocsparse_operation_none, rocsparse_operation_none, m, n, k, &alpha, da, a->nnz,
                           a->d_rpt, a->d_col, db, b->nnz, b->d_rpt, b->d_col, (const real *)nullptr, dd, 0,
                           (const rocsparse_int *)nullptr, (const rocsparse_int *)nullptr, info, &buf_bytes));
    void *buf = nullptr;
    V_HIP(hipMalloc(&buf, buf_bytes ? buf_bytes : 1));
    V_HIP(hipMalloc((void **)&c->d_rpt, sizeof(int) * (size_t)(m + 1)));
    rocsparse_int nnz_c = 0;
    V_HIP(hipEventRecord(e0, 0));
    V_RS(rocsparse_csrgemm_nnz(h, rocsparse_operation_none, rocsparse_operation_none, m, n, k, da, a->nnz, a->d_rpt,
                               a->d_col, db, b->nnz, b->d_rpt, b->d_col, dd, 0, (const rocsparse_int *)nullptr,
                               (const rocsparse_int *)nullptr, dc, c->d_rpt, &nnz_c, info, buf));
    V_HIP(hipEventRecord(e1, 0));
    V_HIP(hipDeviceSynchronize());
    c->nnz = nnz_c;
    V_HIP(hipMalloc((void **)&c->d_col, sizeof(int) * (size_t)(nnz_c > 0 ? nnz_c : 1)));
    V_HIP(hipMalloc((void **)&c->d_val, sizeof(real) * (size_t)(nnz_c > 0 ? nnz_c : 1)));
    V_HIP(hipEventRecord(e2, 0));
    V_RS(RS_CSRGEMM(h, rocsparse_operation_none, rocsparse_operation_none, m, n, k, &alpha, da, a->nnz, a->d_val,
                    a->d_rpt, a->d_col, db, b->nnz, b->d_val, b->d_rpt, b->d_col, (const real *)nullptr, dd, 0,
                    (const real *)nullptr, (const rocsparse_int *)nullptr, (const rocsparse_int *)nullptr, dc,
                    c->d_val, c->d_rpt, c->d_col, info, buf));
    V_HIP(hipEventRecord(e3, 0));
    V_HIP(hipDeviceSynchronize());
    if (ms) {
        float t0 = 0, t1 = 0;
        V_HIP(hipEventElapsedTime(&t0, e0, e1));
        V_HIP(hipEventElapsedTime(&t1, e2, e3));
        *ms = t0 + t1;
    }
    V_HIP(hipFree(buf));
    V_HIP(hipEventDestroy(e0));
    V_HIP(hipEventDestroy(e1));
    V_HIP(hipEventDestroy(e2));
    V_HIP(hipEventDestroy(e3));
    V_RS(rocsparse_destroy_mat_info(info));
    V_RS(rocsparse_destroy_mat_descr(da));
    V_RS(rocsparse_destroy_mat_descr(db));
    V_RS(rocsparse_destroy_mat_descr(dc));
    V_RS(rocsparse_destroy_mat_descr(dd));
}

void spgemm_cu_csr(sfCSR *a, sfCSR *b, sfCSR *c) { nsparse_vendor_spgemm(a, b, c, nullptr); }

void nsparse_vendor_release_csr(sfCSR c)
{
    V_HIP(hipFree(c.d_rpt));
    V_HIP(hipFree(c.d_col));
    V_HIP(hipFree(c.d_val));
}

// y = A x with rocsparse csrmv (adaptive algorithm, analysis outside the timed loop -- the vendor
// library's best case), `reps` launches after one warm-up; returns the mean ms per SpMV by HIP events
// (spmv_cu_csr.cu:46-62 times the same loop with cudaEvents).
float nsparse_vendor_spmv_csr(real *d_y, sfCSR *a, real *d_x, int reps)
{
    g_err = 0;
    rocsparse_handle h = handle();
    rocsparse_mat_descr d;
    rocsparse_mat_info info;
    V_RS(rocsparse_create_mat_descr(&d));
    V_RS(rocsparse_create_mat_info(&info));
    V_RS(RS_CSRMV_ANALYSIS(h, rocsparse_operation_none, a->M, a->N, a->nnz, d, a->d_val, a->d_rpt, a->d_col, info));
    const real one = (real)1, zero = (real)0;
    V_RS(RS_CSRMV(h, rocsparse_operation_none, a->M, a->N, a->nnz, &one, d, a->d_val, a->d_rpt, a->d_col, info, d_x,
                  &zero, d_y));
    hipEvent_t e0, e1;
    V_HIP(hipEventCreate(&e0));
    V_HIP(hipEventCreate(&e1));
    V_HIP(hipDeviceSynchronize());
    V_HIP(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; i++)
        V_RS(RS_CSRMV(h, rocsparse_operation_none, a->M, a->N, a->nnz, &one, d, a->d_val, a->d_rpt, a->d_col, info,
                      d_x, &zero, d_y));
    V_HIP(hipEventRecord(e1, 0));
    V_HIP(hipEventSynchronize(e1));
    float ms = 0;
    V_HIP(hipEventElapsedTime(&ms, e0, e1));
    V_HIP(hipEventDestroy(e0));
    V_HIP(hipEventDestroy(e1));
    V_RS(rocsparse_destroy_mat_info(info));
    V_RS(rocsparse_destroy_mat_descr(d));
    return reps > 0 ? ms / (float)reps : 0.f;
}

}  // extern "C"
