/*
 * nsparse_vendor.h -- the vendor-library baseline of the reference's comparison samples, on ROCm.
 *
 * The reference ships its hash SpGEMM and AMB SpMV next to cuSPARSE wrappers with the same sfCSR
 * interface (reference: cuda-c/inc/nsparse.h:160-166 spgemm_kernel_cu_csr / spgemm_cu_csr;
 * cuda-c/src/kernel/kernel_spgemm_cu_csr.cu:59-203; cuda-c/src/sample/spmv/spmv_cu_csr.cu:13-85)
 * and prints its own numbers beside the vendor's; check_spgemm_answer compares against the vendor
 * result.  libnsparse_vendor_{d,s}.so plays that role with rocSPARSE through its C API
 * (rocsparse_csrgemm_nnz / rocsparse_{s,d}csrgemm / rocsparse_{s,d}csrmv).  It is a BASELINE and a
 * third oracle for the tests: the product libraries libnsparse_{d,s}.so neither link nor call it.
 */
#ifndef NSPARSE_AMD_NSPARSE_VENDOR_H
#define NSPARSE_AMD_NSPARSE_VENDOR_H

#include "nsparse.h"

#ifdef __cplusplus
extern "C" {
#endif

/* C = A * B by rocSPARSE csrgemm.  a / b: device arrays valid.  Allocates c->d_rpt / d_col / d_val
 * (hipMalloc; free with nsparse_vendor_release_csr or release_csr), sets c->M, N, nnz.  Synchronous.
 * *ms (may be NULL): device time of the nnz + compute stages, allocations excluded
 * (kernel_spgemm_cu_csr.cu:186-199).                                                            */
void nsparse_vendor_spgemm(sfCSR *a, sfCSR *b, sfCSR *c, float *ms);
void nsparse_vendor_release_csr(sfCSR c);
/* The reference's own name for the same thing (nsparse.h:165, kernel_spgemm_cu_csr.cu:175-203):
 * C = A * B by the vendor library, c's device arrays allocated here.                            */
void spgemm_cu_csr(sfCSR *a, sfCSR *b, sfCSR *c);

/* y = A x by rocSPARSE csrmv (adaptive, analysis done once outside the loop): one warm-up, then
 * `reps` launches between two HIP events; returns the mean ms per SpMV (spmv_cu_csr.cu:46-62).  */
float nsparse_vendor_spmv_csr(real *d_y, sfCSR *a, real *d_x, int reps);

/* 0, a HIP error code, or 1000 + rocsparse_status of the last call. */
int nsparse_vendor_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
