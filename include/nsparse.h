/*
 * nsparse.h -- C-ABI boundary of the MI355X-native hash-SpGEMM / AMB-SpMV library.
 *
 * This header is the drop-in replacement for the reference's public header
 * (reference: cuda-c/inc/nsparse.h:1-172).  Every struct below keeps the
 * reference's field order and types so that code compiled against either header
 * sees the same ABI, and every entry point keeps the reference's name, argument
 * list and ownership rules.  The reference links these symbols with C++ linkage
 * out of .cu files; here they are exported `extern "C"` from
 *     libnsparse_d.so   (built with -DDOUBLE, real = double)
 *     libnsparse_s.so   (built with -DFLOAT,  real = float)
 * exactly mirroring the reference's two-binaries-per-precision build
 * (reference: cuda-c/Makefile:20-22,99-113).
 *
 * What is NOT here, and why:
 *   - spgemm_kernel_cu_csr / spgemm_cu_csr / the cuSPARSE typed prototypes
 *     (reference nsparse.h:160-166): vendor baseline, out of scope.
 *   - sfBIN (reference nsparse.h:110-121): private to spgemm_kernel_hash; the
 *     MI355X build keeps its binning state in an internal workspace.
 *   - csr_ans_check (reference nsparse.h:150): declared but never defined upstream.
 *
 * Device pointers (the d_* members, d_x, d_y) are plain HIP device pointers.
 * No torch / thrust / rocprim type appears in any signature.
 */
#ifndef NSPARSE_AMD_NSPARSE_H
#define NSPARSE_AMD_NSPARSE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- precision switch (reference nsparse.h:3-11) ------------------------- */
#if defined(FLOAT)
typedef float real;
#define NSPARSE_REAL_IS_FLOAT 1
#else
typedef double real; /* -DDOUBLE or nothing: double, as upstream */
#define NSPARSE_REAL_IS_FLOAT 0
#endif

#define div_round_up(a, b) (((a) % (b) == 0) ? (a) / (b) : (a) / (b) + 1)

/* ---- hardware constants (reference nsparse.h:16-19, re-derived for CDNA4) -
 * The reference hard-codes a 32-lane warp.  gfx950 executes 64-lane wavefronts,
 * so WARP is 64 here.  Callers that size `d_y` as M + WARP (reference
 * spmv_amb.cu:33) therefore still over-allocate enough; the library itself never
 * writes past y[M-1] and never reads past x[N-1] (see sf_spmv_amb below).      */
#define WARP_BIT 6
#define WARP 64
#define MAX_LOCAL_THREAD_NUM 1024
#define MAX_THREAD_BLOCK (MAX_LOCAL_THREAD_NUM / WARP)

/* iteration counts of the sample drivers (reference nsparse.h:22-26) */
#define TRI_NUM 101
#define TEST_NUM 2
#define SPGEMM_TRI_NUM 11

/* AMB format constants (reference nsparse.h:29-38) */
#define sfFLT_MAX 1000000000
#define SHORT_MAX 32768
#define SHORT_MAX_BIT 15
#define USHORT_MAX 65536
#define USHORT_MAX_BIT 16
#define SCL_BORDER 16
#define SCL_BIT ((1 << SCL_BORDER) - 1)
#define MAX_BLOCK_SIZE 20

/* self-check switch of the samples (reference nsparse.h:41) */
#define sfDEBUG

typedef enum { FALSE, TRUE } BOOL;

/* Tuning record for AMB SpMV (reference nsparse.h:50-59). */
typedef struct {
    size_t thread_grid;   /* workgroups of the SpMV launch                      */
    size_t thread_block;  /* threads per workgroup (multiple of 64)             */
    BOOL isPlan;          /* FALSE: sf_csr2amb chooses seg_size / block_size    */
    int SIGMA;            /* sorting window of SELL-C-sigma                     */
    size_t seg_size;      /* columns per segment, <= 65536                      */
    size_t seg_num;       /* ceil(N / seg_size)                                 */
    int block_size;       /* consecutive columns per stored column id, 1..20    */
} sfPlan;

/* CSR container with host and device mirrors (reference nsparse.h:62-75). */
typedef struct {
    int *rpt;   /* host   row pointers  [M+1] */
    int *col;   /* host   column ids    [nnz] */
    real *val;  /* host   values        [nnz] */
    int *d_rpt; /* device row pointers  [M+1] */
    int *d_col; /* device column ids    [nnz] */
    real *d_val;/* device values        [nnz] */
    int M;
    int N;
    int nnz;
    int nnz_max;        /* longest row */
    char *matrix_name;  /* borrowed from the caller, never freed here */
} sfCSR;

/* AMB container (reference nsparse.h:78-107).  Only the d_* members are filled
 * by sf_csr2amb, as upstream; the host mirrors stay untouched.               */
typedef struct {
    int *cs;
    unsigned int *cl;
    unsigned short *sellcs_col;
    real *sellcs_val;
    unsigned short *s_write_permutation;
    unsigned short *s_write_permutation_offset;
    int *write_permutation;
    int *d_cs;                 /* [c_size]   first value slot of each chunk          */
    unsigned int *d_cl;        /* [c_size]   low16 = blocks per lane - 1, high16 = segment */
    unsigned short *d_sellcs_col; /* [nnz/block_size] block base column inside the segment */
    real *d_sellcs_val;        /* [nnz]      values, zero padded                      */
    unsigned short *d_s_write_permutation;        /* [c_size*chunk] output row % 65536 */
    unsigned short *d_s_write_permutation_offset; /* [c_size]       output row / 65536 */
    int *d_write_permutation;  /* [c_size*chunk] output row, uncompressed             */
    int block_size;
    int nnz;            /* padded value count */
    int M;
    int N;
    int pad_M;          /* chunk * ceil(M / chunk) */
    int chunk;          /* rows per chunk: 32 (upstream layout) or 64 (one wavefront) */
    int SIGMA;
    int group_num_col;  /* == seg_num */
    int nnz_max;
    int c_size;         /* non-empty chunks */
    size_t seg_size;
    size_t seg_num;
    char *matrix_name;
} sfAMB;

/* ========================================================================== */
/*  Entry points that replace the reference one-for-one                       */
/* ========================================================================== */

/* x[i] = drand48(), seeded with time(NULL)      (reference nsparse.cu:190-199) */
void init_vector(real *x, int row);

/* MatrixMarket coordinate file -> host CSR      (reference nsparse.cu:14-144)
 * Mirrors the upstream loader: "general" in the banner line => stored as is,
 * anything else => off-diagonal entries mirrored with the same sign; a missing
 * third token => value 1.0; no duplicate merging; in-row order = file order.
 * Exits with "Cannot find file" like upstream when the file is missing.       */
void init_csr_matrix_from_file(sfCSR *mat, char *file_name);

/* hipMalloc d_rpt/d_col/d_val and copy H2D      (reference nsparse.cu:146-156) */
void csr_memcpy(sfCSR *mat);
/* malloc rpt/col/val and copy D2H               (reference nsparse.cu:158-168) */
void csr_memcpyDtH(sfCSR *mat);

/* frees (structs by value, as upstream)         (reference nsparse.cu:202-235) */
void release_cpu_csr(sfCSR mat);
void release_cpu_amb(sfAMB mat);
void release_csr(sfCSR mat);
void release_amb(sfAMB mat);

/* plan handling                                 (reference nsparse.cu:171-187) */
void init_plan(sfPlan *plan);
void set_plan(sfPlan *plan, size_t seg_size, int block_size);

/* CSR (device arrays valid) -> AMB; allocates every mat->d_*; fills *plan and
 * sets plan->isPlan = TRUE                      (reference convert_amb.cu:835-929)
 * d_x must hold N + MAX_BLOCK_SIZE elements only if the caller later runs a
 * foreign kernel on the AMB arrays; this library's own kernel clamps reads.   */
void sf_csr2amb(sfAMB *mat, sfCSR *csr_mat, real *d_x, sfPlan *plan);

/* CPU scalar CSR SpMV, the reference's only CPU path (reference nsparse.cu:240-259) */
void csr_kernel(real *csr_ans, sfCSR *cpu_mat, real *rhs_vec);
/* relative-error check, 1e-5 (float) / 1e-8 (double); prints the verdict line
 *                                               (reference nsparse.cu:261-298) */
void ans_check(real *csr_ans, real *ans_vec, int N);

/* y = A x from AMB; synchronous on return       (reference kernel_spmv_amb.cu:98-104) */
void sf_spmv_amb(real *d_y, sfAMB *mat, real *d_x, sfPlan *plan);

/* flop = 2 * (number of intermediate products)  (reference kernel_spgemm_cu_csr.cu:35-57) */
void get_spgemm_flop(sfCSR *a, sfCSR *b, int M, long long int *flop);
/* exact nnz / rpt / col, values to 1e-6 (float) / 1e-9 (double)
 *                                               (reference nsparse.cu:300-353) */
void check_spgemm_answer(sfCSR c, sfCSR ans);
/* C = A B with the hash algorithm; allocates c->d_rpt/d_col/d_val, sets
 * c->M/N/nnz; synchronous on return             (reference kernel_spgemm_hash_d.cu:1035-1075)
 * A product with nnz(C) >= 2^31 does not fit sfCSR's int row pointers: error -40, c->d_* NULL
 * (upstream wraps silently).  Rows of B should have ascending columns (true for the loader's
 * output on SuiteSparse files); other orders are accepted and take the slower global-table path
 * for rows beyond the LDS tables.                                                              */
void spgemm_kernel_hash(sfCSR *a, sfCSR *b, sfCSR *c);

/* ========================================================================== */
/*  Extensions (prefix nsparse_): not in the reference; additive only          */
/* ========================================================================== */

/* 0 when the last library call succeeded; otherwise a HIP error code or a
 * negative library code.  The reference aborts on every error; so does this
 * library unless NSPARSE_NO_ABORT=1 is set in the environment.                */
int nsparse_last_error(void);
const char *nsparse_last_error_string(void);

/* Verdicts of the two check functions as return values (number of offending
 * entries, or a negative code for a structural mismatch) for test harnesses;
 * ans_check / check_spgemm_answer print and call these.                        */
int nsparse_ans_check_count(const real *csr_ans, const real *ans_vec, int N);
int nsparse_check_spgemm_count(const sfCSR *c, const sfCSR *ans);

/* "gfx950 <double|float> <hash>": target, precision and the content hash of the library's sources at
 * build time (csrc/Makefile); tests recompute the hash from the tree to catch a stale prebuilt .so.   */
const char *nsparse_build_info(void);

/* Deterministic replacement for init_vector: splitmix64 stream, U[0,1).       */
void nsparse_init_vector_seeded(real *x, int row, unsigned long long seed);

/* Rows per AMB chunk used by the next sf_csr2amb: 32 = upstream layout,
 * 64 = one CDNA wavefront per chunk (default).  Returns the value in force.   */
int nsparse_set_amb_chunk(int chunk);

/* AMB bytes moved by one SpMV according to the reference's own footprint model
 * (reference convert_amb.cu:785-791), evaluated on a converted matrix.        */
long long nsparse_amb_footprint_bytes(const sfAMB *mat);

/* Numeric-only re-run on an existing structure: c->d_rpt / c->d_col stay, c->d_val
 * is recomputed (the cuda-cpp tree's SpGEMM_Hash_Numeric, HashSpGEMM_volta.hpp:1018-1031). */
void nsparse_spgemm_hash_numeric(sfCSR *a, sfCSR *b, sfCSR *c);

/* 1 (default): columns of every C row ascend, as check_spgemm_answer requires.  0: rows that
 * went through a hash table are written in table order (the cuda-cpp tree's template<bool sort>,
 * HashSpGEMM_volta.hpp:585-604); dense-window rows stay ordered.  rpt and the SET of columns of
 * each row are unchanged.  Numeric-only re-runs need the sorted structure.  Returns the old value. */
int nsparse_spgemm_set_sorted(int on);

/* 1: C.val of every later spgemm_kernel_hash / nsparse_spgemm_hash_numeric is summed in ONE FIXED ORDER -- per entry
 * of C, the products in the order of the A entries of its row -- so two runs on the same inputs give the same bytes
 * (the default accumulates with LDS atomics in whatever order the lanes arrive, like the reference:
 * kernel_spgemm_hash_d.cu:829-927, README "relies on warp lock-step").  A debugging aid (hard part 1): the
 * values are recomputed by a second pass over the finished structure, one binary search per (A entry, C entry).
 * Returns the previous setting.                                                                                */
int nsparse_set_deterministic(int on);

/* 1 when roctx ranges are being emitted around the phases of the calls ("nsparse:spgemm" > setup / symbolic /
 * numeric, "nsparse:csr2amb", "nsparse:spmv_amb"): a profiler is in the process (rocprofv3 --marker-trace) or
 * NSPARSE_ROCTX=1.  The marker library is dlopen'ed then; the product library never links it.               */
int nsparse_trace_ranges(void);

/* Statistics of the last spgemm_kernel_hash call. */
typedef struct {
    long long n_prod;         /* intermediate products                              */
    long long nnz_c;          /* 64-bit: reported even when it does not fit sfCSR's int (error -40) */
    int max_prod_row;         /* longest row of intermediate products               */
    int max_nnz_row;          /* longest row of C                                   */
    int sym_bin_size[12];     /* rows per symbolic bin (0 tiny, 1-5 hash, 6-8 dense window, 9-10 bit window);
                               * twin rows (below) are in none                        */
    int num_bin_size[12];     /* rows per numeric bin                               */
    int sym_fail_rows;        /* rows that overflowed LDS and went to the global table */
    float ms_setup;           /* products + binning   (HIP events; the four phase times are 0 unless
                               * nsparse_set_bin_timing(1) or profiling mode)               */
    float ms_symbolic;        /* all symbolic kernels + scan     (HIP events)       */
    float ms_numeric;         /* numeric binning + all numeric kernels              */
    float ms_total;           /* whole call                                         */
    float ms_sym_bin[12];      /* per-bin kernel time, HIP events on the bin's own stream */
    float ms_num_bin[12];
    int twin_rows;            /* rows with the column pattern of another row (found by pattern): not in
                               * the symbolic bins, they take their pattern leader's structure */
} nsparse_spgemm_stats;
void nsparse_get_spgemm_stats(nsparse_spgemm_stats *out);

/* Bin ladder of the symbolic / numeric phase, 18 ints each: tiny, hash_t[4], dense_span[3],
 * dense_ratio, bits_span[2], bits_ratio, bits_min, bits_wide_min, bits_wide_span, rank_span, rank_ratio,
 * rank_max_nz.
 * Row (n, span) -> bin: n <= tiny: 0;
 * span <= dense_span[2] and (span <= dense_ratio * n or 4 * span <= dense_ratio * products of the
 * row -- the numeric phase bins by nnz, n, but a window also pays off by its products):
 * 6 + #(dense_span < span);
 * n > bits_min and span <= bits_span[1] and span <= bits_ratio * n: 9 + (span > bits_span[0]);
 * bits_wide_min > 0 and n > bits_wide_min and span <= bits_wide_span: 10 (window in pieces);
 * rank_span > 0 (numeric ladder) and span <= rank_span and n <= rank_max_nz and (span <= rank_ratio * n or
 * span <= 4 * products): 9 (ranked window: accumulators addressed by bitmap rank);
 * else 1 + #(hash_t < n).                                                               */
void nsparse_get_spgemm_bins(int *sym_thresholds, int *num_thresholds);

/* The fused helper kernels of spgemm_kernel_hash (matrices of up to 256 K rows) hold a grid barrier inside an
 * ordinary launch, so every workgroup of their grid must be resident at once.  *coresident: how many
 * 1024-thread workgroups the census at first use found resident together on the current device AS THIS PROCESS
 * SEES IT (HSA_CU_MASK / ROC_GLOBAL_CU_MASK, partitions; -1: no census yet) -- larger grids take the kernel
 * chains.  *fallbacks: calls that were repeated with the chains because a barrier timed out after all (CUs
 * taken away since the census).  Returns 1 while the device's context still fuses, 0 once it has given up.
 * Either pointer may be NULL.                                                                          */
int nsparse_fused_state(int *coresident, int *fallbacks);

/* 1: serialise the row bins on one stream (clean per-kernel durations for roofline work
 *    and rocprof, per-bin timings on); 0 (default): bins overlap on their own streams.  */
void nsparse_set_profiling(int on);

/* Times in nsparse_spgemm_stats -- per bin (ms_sym_bin / ms_num_bin: HIP events around the kernels of
 * every bin) and per phase (ms_setup ... ms_total: four event records per call).  Off by default: every
 * event record is a runtime call on the launch path and a packet on the stream (cant class: 0.337 ->
 * 0.325 ms without the four phase events).  Profiling mode implies it.  Returns the previous setting. */
int nsparse_set_bin_timing(int on);

/* 1 (default): device blocks released by release_csr/release_amb and the internal
 * workspaces are kept in a cache and reused; 0: every call hipMalloc/hipFree's
 * like upstream ("reference-compatible timing"); 2: no cache either, but the
 * blocks come from the runtime's stream-ordered allocator (hipMallocAsync /
 * hipFreeAsync, default pool set to keep freed memory) -- an allocation call per
 * array inside every call as upstream, without hipFree's device-wide wait.     */
void nsparse_set_workspace_cache(int on);
/* Return every cached device block to the driver. */
void nsparse_trim_workspace(void);

/* Duration in ms of the last sf_spmv_amb kernel sequence measured with HIP
 * events on the launch stream (valid when profiling is on).                    */
float nsparse_last_spmv_ms(void);

/* Asynchronous form of sf_spmv_amb on a caller stream (hipStream_t passed as
 * void*), no device synchronisation: used by the row-sharded multi-GPU driver
 * to overlap with RCCL.                                                        */
void nsparse_spmv_amb_async(real *d_y, sfAMB *mat, real *d_x, sfPlan *plan, void *stream);

/* Binary image of a host CSR (exactly what init_csr_matrix_from_file produces).  0 on success.
 * With NSPARSE_BIN_CACHE=1 the loader reads / writes `<file>.csr.bin` by itself.  An image
 * written by the other precision build is rejected (-2).                                   */
int nsparse_save_csr_bin(const sfCSR *mat, const char *path);
int nsparse_load_csr_bin(sfCSR *mat, const char *path);

/* The plan sf_csr2amb wrote back (reference convert_amb.cu:906), kept beside the matrix so that
 * the next run skips the search (up to 100 format builds + 500 timed SpMVs with
 * NSPARSE_AMB_TUNE=timed).  A one-line text file; load returns 0 and sets isPlan = TRUE, or
 * non-zero (missing, other precision build, other chunk size) and leaves *plan alone.  The sample
 * driver amb_{s,d} uses `<file>.plan` when NSPARSE_BIN_CACHE=1 and no plan is given.          */
int nsparse_save_plan(const sfPlan *plan, const char *path);
int nsparse_load_plan(sfPlan *plan, const char *path);

/* Synthetic stand-ins for the SuiteSparse inputs named in BASELINE.md (there is
 * no network on the GPU box).  Each fills the HOST side of *mat with malloc'd
 * arrays (free with release_cpu_csr); columns ascend inside every row.
 *   kind 0: 3-dof 27-point brick  nx*ny*nz nodes  (cant class;   p0,p1,p2 = nx,ny,nz)
 *   kind 1: scalar 27-point grid  nx*ny*nz        (nlpkkt class; p0,p1,p2 = nx,ny,nz)
 *   kind 2: power-law web graph   p0 rows, ~p1 nnz              (webbase class)
 *   kind 3: R-MAT scale p0, edge factor p1 (or exactly p2 edges when p2 > 0, for fractional
 *           factors), duplicates merged                           (config 5)
 *   kind 4: web graph of 32-page sites, p0 rows, ~p1 nnz: index pages, popular directories,
 *           template sites -- the SuiteSparse statistics of webbase-1M (config 3)
 *   kind 5: kind 0 renumbered inside bands of three mesh planes, 7.4 % of the node couplings
 *           dropped (3 x 3 dof at once): the SuiteSparse statistics of cant, irregular numbering
 *   kind 6: kind 5 + scalar perturbations: p2 = nz + permille * 2^32; that share of the nodes
 *           gets one dof constrained (row = diagonal, column gone) or loses one scalar
 *           coupling, so its rows no longer share one column pattern
 * Rows [row_begin,row_end) only (row_end <= 0: all rows) so that one rank of a
 * row-sharded run can generate just its block.                                  */
void nsparse_synth_csr(sfCSR *mat, int kind, long long p0, long long p1, long long p2,
                       unsigned long long seed, long long row_begin, long long row_end);

/* *mat (host arrays) as a Matrix Market file in SuiteSparse conventions (1-based coordinate
 * format, entries sorted by column then row): flavour 0 real general, 1 real symmetric (lower
 * triangle only; the matrix must be symmetric), 2 pattern general, 3 pattern symmetric.  Values
 * carry enough digits to read back bit-identical.  0 on success.  Lets the loader and the sample
 * drivers run at full size on a box without network (the stand-ins as files).                  */
int nsparse_write_mtx(const sfCSR *mat, const char *path, int flavour);

#ifdef __cplusplus
}
#endif
#endif /* NSPARSE_AMD_NSPARSE_H */
