/*
 * nsparse_dist.h -- row-sharded AMB SpMV across the GPUs of one node, native (C ABI, RCCL).
 *
 * The reference is single-GPU; this is the multi-GPU row of the hot path that
 * BASELINE.json's north_star / SURVEY 8e name: 1-D row blocks balanced by non-zeros, every rank
 * converts ITS block to AMB against the full x (replicated), computes its part of y with the same
 * kernel as sf_spmv_amb (reference kernel_spmv_amb.cu:10-104), then ONE ncclAllGather puts the
 * whole y on every rank.  The shards of y are disjoint: no cross-GPU reduction, the result does
 * not depend on the rank count.
 *
 * libnsparse_dist_{d,s}.so links libnsparse_{d,s}.so and RCCL; the product libraries themselves
 * stay RCCL-free.  One rank per GPU: one process per GPU (nsparse_dist_init with an id the
 * caller broadcasts: MPI_Bcast, a file, torch.distributed ...) or one thread per GPU of one
 * process (nsparse_dist_init_all).  Everything here is plain pointers and sizes.
 *
 * Per SpMV on one stream, nothing else: [memset of the local y when the matrix has several
 * column segments] -> k_spmv_amb_row -> ncclAllGather (in place, recvcount = rows of the longest
 * block) -> [one copy kernel that closes the gaps when the blocks are unequal].  No host
 * synchronisation, no allocation, no Python.  nsparse_dist_capture() records that sequence once
 * into a hipGraph; afterwards a SpMV is one hipGraphLaunch.
 */
#ifndef NSPARSE_AMD_NSPARSE_DIST_H
#define NSPARSE_AMD_NSPARSE_DIST_H

#include "nsparse.h"

#ifdef __cplusplus
extern "C" {
#endif

#define NSPARSE_DIST_ID_BYTES 128 /* sizeof(ncclUniqueId) */

typedef struct nsparse_dist *nsparse_dist_t;

/* ---- partition (host, no device): SURVEY 8e ------------------------------------------------
 * cuts[0] = 0 <= cuts[1] <= ... <= cuts[world] = M; rank r owns rows [cuts[r], cuts[r+1]).
 * Rank r ends at the row boundary, rounded to a multiple of `align` rows (the AMB chunk: 64),
 * where the running non-zero count reaches (r+1)/world of the total.  Blocks may be empty.
 * Returns 0, or -1 on bad arguments.                                                           */
int nsparse_dist_partition_nnz(const int *rpt, int M, int world, int align, int *cuts);
/* Same rule on 64-bit per-row work (SpGEMM: intermediate products per row, what the reference's
 * set_intprod_num counts, kernel_spgemm_hash_d.cu:70-86).                                      */
int nsparse_dist_partition_work(const long long *work_per_row, int M, int world, int align, int *cuts);
/* Rows [begin, end) of a HOST sfCSR as a new host sfCSR over the same columns (malloc'd arrays,
 * free with release_cpu_csr).                                                                  */
int nsparse_dist_csr_row_block(const sfCSR *full, int begin, int end, sfCSR *block);

/* ---- communicator --------------------------------------------------------------------------
 * Rank 0 makes an id, the caller hands it to every rank, every rank calls init with its GPU
 * current (hipSetDevice before).  Collective.  0 on success, else 2000 + ncclResult_t or a HIP
 * error code (also in nsparse_dist_last_error()).                                              */
int nsparse_dist_unique_id(char id[NSPARSE_DIST_ID_BYTES]);
/* id == NULL: no communicator -- the handle computes its own rows (gather = 0) only; that is how the
 * tests run the ranks of a partition one after the other on a single GPU.                       */
int nsparse_dist_init(nsparse_dist_t *h, const char id[NSPARSE_DIST_ID_BYTES], int rank, int world);
/* One process, `world` GPUs (devices 0 .. world-1), one handle per GPU: ncclCommInitAll.  The
 * caller then drives handle r from its own thread with device r current.                       */
int nsparse_dist_init_all(nsparse_dist_t *handles, int world);
void nsparse_dist_destroy(nsparse_dist_t h);
/* GPUs this process sees (hipGetDeviceCount; 0 on error): a launcher compares it with its rank count
 * BEFORE any collective, so that "more ranks than GPUs" is a message and not a hang.              */
int nsparse_dist_device_count(void);
/* Watchdog (round 4).  No call of this library waits for a peer for ever: nsparse_dist_init gives
 * ncclCommInitRank, and nsparse_dist_sync / _barrier / _allreduce_f64 / _spmv_loop give the handle's
 * stream, this many seconds (default 60, NSPARSE_DIST_TIMEOUT_S); then the communicator is aborted
 * and the call returns -7 (stream) or -8 (communicator creation).  Returns the previous value.   */
double nsparse_dist_set_timeout(double seconds);
/* All ranks: returns when every rank has reached the call and the handle's stream is idle (a
 * one-int ncclAllReduce; world 1 or a handle without communicator: the stream only).             */
int nsparse_dist_barrier(nsparse_dist_t h);
/* vals[0, n) <- sum (op 0) or max (op 1) over the ranks; host memory in and out, n <= 64.  With
 * _barrier this is all a launcher needs around a timed loop: no second communication library.    */
int nsparse_dist_allreduce_f64(nsparse_dist_t h, double *vals, int n, int op);

/* ---- SpMV ----------------------------------------------------------------------------------
 * a_local: this rank's row block, device arrays valid (csr_memcpy), M = cuts[rank+1] - cuts[rank]
 * rows, N = columns of the whole matrix.  Converts it to AMB (sf_csr2amb; *plan as there: isPlan
 * FALSE lets the library choose and writes the choice back).  cuts: the world+1 row cuts, equal
 * on every rank.  d_x_any: device vector of N + MAX_BLOCK_SIZE elements for the plan search.     */
int nsparse_dist_spmv_setup(nsparse_dist_t h, sfCSR *a_local, const int *cuts, real *d_x_any, sfPlan *plan);
/* Drop the handle's matrix (AMB arrays, staging, recorded graph); the communicator stays and the
 * handle takes the next nsparse_dist_spmv_setup.                                                 */
int nsparse_dist_release_matrix(nsparse_dist_t h);
/* Elements the caller must allocate for the gathered y: world * (rows of the longest block),
 * at least M.  (The all-gather is in place with equal shares.)                                  */
long long nsparse_dist_y_elems(nsparse_dist_t h);
/* y[0, M) = A x on every rank.  Asynchronous on the handle's stream.  gather = 0: only this
 * rank's rows, at d_y + cuts[rank] (compute-only timing).                                       */
int nsparse_dist_spmv(nsparse_dist_t h, real *d_y, const real *d_x, int gather);
/* Record the sequence for (d_y, d_x, gather) into a hipGraph; later nsparse_dist_spmv calls with
 * the same three arguments replay it with one hipGraphLaunch.  0, or an error (plain launches
 * keep working).                                                                                */
int nsparse_dist_capture(nsparse_dist_t h, real *d_y, const real *d_x, int gather);
int nsparse_dist_sync(nsparse_dist_t h);
/* The gap-closing step by itself (unequal blocks are gathered in equal shares of `rpr` elements):
 * y[cuts[r] + i] = staged[r * rpr + i].  d_cuts: world + 1 ints on the device.  Exposed so that
 * the step can be checked on one GPU.                                                           */
int nsparse_dist_close_gaps(real *d_y, const real *d_staged, const int *d_cuts, int world, int rpr, int M, void *stream);
/* `iters` back-to-back SpMVs from a native loop: *ms_wall = host clock from the first enqueue to
 * the end of the last one (per SpMV), *ms_events = the same by HIP events on the handle's
 * stream, *us_host = host time per SpMV spent enqueueing (no wait included).  Any may be NULL. */
int nsparse_dist_spmv_loop(nsparse_dist_t h, real *d_y, const real *d_x, int gather, int iters,
                           double *ms_wall, double *ms_events, double *us_host);
/* ---- SpGEMM by 1-D row blocks (stretch row) ---------------------------------------
 * C[rows of rank r, :] = A[rows of rank r, :] * B, B whole on every rank, no collective inside the
 * algorithm.  work[i] = intermediate products of row i of A * B from the HOST arrays (what the
 * reference's set_intprod_num counts, kernel_spgemm_hash_d.cu:70-86): feed it to
 * nsparse_dist_partition_work for product-balanced cuts.                                        */
int nsparse_dist_spgemm_row_work(const sfCSR *a_host, const sfCSR *b_host, long long *work);
/* This rank's block: a_block (its rows of A; nsparse_dist_csr_row_block + csr_memcpy) and b with
 * device arrays valid -> c_block, device arrays only (release_csr), exactly spgemm_kernel_hash on
 * the block.  0, or the product library's error code.                                            */
int nsparse_dist_spgemm(nsparse_dist_t h, sfCSR *a_block, sfCSR *b, sfCSR *c_block);
/* The whole C on every rank: block sizes by one all-reduce, then every rank's rpt / col / val
 * stretch broadcast straight to its place (no padding).  c_full: device arrays only, release with
 * nsparse_dist_release_gathered.  Collective; -40 when nnz(C) does not fit int.                  */
int nsparse_dist_spgemm_gather(nsparse_dist_t h, const int *cuts, const sfCSR *c_block, sfCSR *c_full);
void nsparse_dist_release_gathered(sfCSR c_full);

/* The rank's AMB matrix and plan (footprint model, tests); owned by the handle.                 */
const sfAMB *nsparse_dist_amb(nsparse_dist_t h);
const sfPlan *nsparse_dist_plan(nsparse_dist_t h);
void *nsparse_dist_stream(nsparse_dist_t h); /* hipStream_t */

int nsparse_dist_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
