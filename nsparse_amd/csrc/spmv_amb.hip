// spmv_amb.hip -- y = A x from the AMB format, for gfx950.
//
// Replaces (reference file:line):
//   kernel_spmv_init_ans            cuda-c/src/kernel/kernel_spmv_amb.cu:10-19
//   kernel_spmv_amb_atomic<1..20>   cuda-c/src/kernel/kernel_spmv_amb.cu:21-79
//   sf_spmv_amb (+ the template recursion dispatcher)            :81-104
//
// Same traversal as the reference (one lane per (chunk, row slot); per block one 16-bit
// column + block_size values, values laid out column-major inside the chunk so that a
// wavefront reads contiguous memory), re-thought for CDNA4:
//   * a chunk is 64 rows = one wavefront (chunk 32 also supported: two chunks per wave);
//   * the value / column streams are read once -> nontemporal loads, x stays cacheable;
//   * when the matrix has a single column segment every output row is owned by exactly one
//     lane, so the result is stored, not atomically added, and is bit-reproducible; with
//     several segments the native fp64/fp32 global atomic add is used (the reference spins
//     on a 64-bit CAS, kernel_spmv_amb.cu:70-76);
//   * workgroup b runs on XCD b % 8 (observed dispatch order); for matrices that fit the Infinity
//     Cache the block index is remapped so that each XCD walks one contiguous eighth of the chunks
//     and its private L2 sees one contiguous window of x instead of eight interleaved ones; matrices
//     that stream from HBM keep the natural order (see launch_bs);
//   * reads of x are clamped to N-1 and rows >= M are not written, so neither the
//     N + MAX_BLOCK_SIZE / M + WARP over-allocation of the reference's callers
//     (spmv_amb.cu:32-33) nor the uninitialised tail of x can influence the result.
#include <cstdlib>

#include "internal.h"

namespace nsp {
namespace spmv {

static float g_last_ms = 0.f;

template <int BSZ, int C, bool ATOMIC, bool NT>
__global__ __launch_bounds__(1024) void k_spmv_amb(real *__restrict__ y, const real *__restrict__ val,
                                                   const unsigned short *__restrict__ col,
                                                   const unsigned int *__restrict__ cl,
                                                   const int *__restrict__ cs,
                                                   const real *__restrict__ x,
                                                   const unsigned short *__restrict__ perm,
                                                   const unsigned short *__restrict__ perm_off,
                                                   int rows, int seg_size, int M, int N, int nb8)
{
    // XCD-aware remap: hardware block b -> logical block (b % 8) * nb8 + b / 8
    const int lb = nb8 > 0 ? (int)(blockIdx.x & 7) * nb8 + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const long long i = (long long)lb * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    const int c = (int)(i / C);
    const int lane = (int)(i % C);
    const int row = (int)__builtin_nontemporal_load(perm + i) + (int)perm_off[c] * USHORT_MAX;
    const int cs0 = cs[c];
    const unsigned int length = cl[c];
    const int width = (int)(length & SCL_BIT);
    const int c_off = (int)(length >> SCL_BORDER) * seg_size;
    const real *v = val + cs0 + lane;
    const unsigned short *cp = col + cs0 / BSZ + lane;
    const int nmax = N - 1;
    real acc = 0;
#pragma unroll 4
    for (int h = 0; h <= width; h++) {
        const int cc = (int)(NT ? __builtin_nontemporal_load(cp) : *cp) + c_off;
#pragma unroll
        for (int b = 0; b < BSZ; b++) {
            const int xi = cc + b < nmax ? cc + b : nmax;
            acc += (NT ? __builtin_nontemporal_load(v) : *v) * x[xi];
            v += C;
        }
        cp += C;
    }
    if (row < M) {
        if (ATOMIC) unsafeAtomicAdd(y + row, acc);
        else y[row] = acc;
    }
}

// Second form of the same traversal for one chunk per wavefront (C = 64), pipelined by hand.  The first
// form (k_spmv_amb: `#pragma unroll 4`) makes, per batch of four blocks, a round trip for the column ids
// and a dependent one for x, and its remainder loop two per block: a row of nine blocks (27-point
// stencil, block size 3) costs 8 dependent trips per wavefront and the kernel is bound by their latency,
// not by HBM (0.192 ms where the byte stream alone needs 0.15).  Here the column ids of batch k + 1 are
// requested together with the values of batch k, so a batch costs ONE exposed trip (the x gathers), the
// width of a chunk is wave-uniform (one `cl` word per chunk), so the tail of a row is a scalar branch
// per block instead of a loop: 4-5 trips for the same row.
template <int BSZ, bool ATOMIC, int UB>
__global__ __launch_bounds__(1024) void k_spmv_amb_pipe(real *__restrict__ y, const real *__restrict__ val,
                                                        const unsigned short *__restrict__ col,
                                                        const unsigned int *__restrict__ cl,
                                                        const int *__restrict__ cs,
                                                        const real *__restrict__ x,
                                                        const unsigned short *__restrict__ perm,
                                                        const unsigned short *__restrict__ perm_off,
                                                        int rows, int seg_size, int M, int N, int nb8, int abl)
{
    // abl (NSPARSE_SPMV_ABL, diagnostics; results are wrong by construction): 1 = x read at the lane's own
    // index instead of the gathered column, 2 = plain store instead of the atomic add
    constexpr int C = 64;
    const int lb = nb8 > 0 ? (int)(blockIdx.x & 7) * nb8 + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const long long i = (long long)lb * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    const int c = (int)(i >> 6);
    const int lane = (int)(i & 63);
    // one chunk per wavefront: the chunk's words are wave-uniform (scalar loads)
    const int cu = __builtin_amdgcn_readfirstlane(c);
    const int cs0 = cs[cu];
    const unsigned int length = cl[cu];
    const int nblk = (int)(length & SCL_BIT) + 1;
    const int c_off = (int)(length >> SCL_BORDER) * seg_size;
    const int row = (abl & 16) ? (int)(i & 0xfffff) : (int)__builtin_nontemporal_load(perm + i) + (int)perm_off[cu] * USHORT_MAX;
    const real *v = val + cs0 + lane;
    const unsigned short *cp = col + cs0 / BSZ + lane;
    const int nmax = N - 1;
    real acc = 0;
    int ccur[UB], cnxt[UB];
#pragma unroll
    for (int u = 0; u < UB; u++) ccur[u] = (u < nblk && !(abl & 8)) ? (int)__builtin_nontemporal_load(cp + u * C) : 0;
    for (int h0 = 0; h0 < nblk; h0 += UB) {
        real vv[UB][BSZ];
#pragma unroll
        for (int u = 0; u < UB; u++)
            if (h0 + u < nblk) {
#pragma unroll
                for (int b = 0; b < BSZ; b++) vv[u][b] = __builtin_nontemporal_load(v + (u * BSZ + b) * C);
            }
#pragma unroll
        for (int u = 0; u < UB; u++)
            cnxt[u] = (h0 + UB + u < nblk && !(abl & 8)) ? (int)__builtin_nontemporal_load(cp + (UB + u) * C) : 0;
        real xx[UB][BSZ];
#pragma unroll
        for (int u = 0; u < UB; u++)
            if (h0 + u < nblk) {
                const int cc = ccur[u] + c_off;
#pragma unroll
                for (int b = 0; b < BSZ; b++) xx[u][b] = (abl & 1) ? x[lane + b] : x[cc + b < nmax ? cc + b : nmax];
            }
#pragma unroll
        for (int u = 0; u < UB; u++)
            if (h0 + u < nblk) {
#pragma unroll
                for (int b = 0; b < BSZ; b++) acc += vv[u][b] * xx[u][b];
            }
#pragma unroll
        for (int u = 0; u < UB; u++) ccur[u] = cnxt[u];
        v += UB * BSZ * C;
        cp += UB * C;
    }
    if (row < M) {
        if (ATOMIC && !(abl & 2)) unsafeAtomicAdd(y + row, acc);
        else y[row] = acc;
    }
}


// Third form (default): chunks of at most NB blocks keep the WHOLE row in
// flight -- all column ids first, then every value and every x of the row in one go: three dependent
// trips per chunk (chunk words, column ids, values + x) whatever the width.  Wider chunks fall back to
// the pipelined loop.  nlpkkt class 0.1727 -> 0.1697 ms, cant class (cache-resident) 12.7 -> 11.3 us.
template <int BSZ, bool ATOMIC, int NB, int UB>
__global__ __launch_bounds__(1024) void k_spmv_amb_row(real *__restrict__ y, const real *__restrict__ val,
                                                       const unsigned short *__restrict__ col,
                                                       const unsigned int *__restrict__ cl,
                                                       const int *__restrict__ cs,
                                                       const real *__restrict__ x,
                                                       const unsigned short *__restrict__ perm,
                                                       const unsigned short *__restrict__ perm_off,
                                                       int rows, int seg_size, int M, int N, int nb8)
{
    constexpr int C = 64;
    const int lb = nb8 > 0 ? (int)(blockIdx.x & 7) * nb8 + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const long long i = (long long)lb * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    const int c = (int)(i >> 6);
    const int lane = (int)(i & 63);
    const int cu = __builtin_amdgcn_readfirstlane(c);
    const int cs0 = cs[cu];
    const unsigned int length = cl[cu];
    const int nblk = (int)(length & SCL_BIT) + 1;
    const int c_off = (int)(length >> SCL_BORDER) * seg_size;
    const int row = (int)__builtin_nontemporal_load(perm + i) + (int)perm_off[cu] * USHORT_MAX;
    const real *v = val + cs0 + lane;
    const unsigned short *cp = col + cs0 / BSZ + lane;
    const int nmax = N - 1;
    real acc = 0;
    if (nblk <= NB) {
        int cc[NB];
        real vv[NB][BSZ], xx[NB][BSZ];
#pragma unroll
        for (int u = 0; u < NB; u++) cc[u] = u < nblk ? (int)__builtin_nontemporal_load(cp + u * C) : 0;
#pragma unroll
        for (int u = 0; u < NB; u++)
            if (u < nblk) {
#pragma unroll
                for (int b = 0; b < BSZ; b++) vv[u][b] = __builtin_nontemporal_load(v + (u * BSZ + b) * C);
            }
#pragma unroll
        for (int u = 0; u < NB; u++)
            if (u < nblk) {
                const int c0 = cc[u] + c_off;
#pragma unroll
                for (int b = 0; b < BSZ; b++) xx[u][b] = x[c0 + b < nmax ? c0 + b : nmax];
            }
#pragma unroll
        for (int u = 0; u < NB; u++)
            if (u < nblk) {
#pragma unroll
                for (int b = 0; b < BSZ; b++) acc += vv[u][b] * xx[u][b];
            }
    } else {
        int ccur[UB], cnxt[UB];
#pragma unroll
        for (int u = 0; u < UB; u++) ccur[u] = u < nblk ? (int)__builtin_nontemporal_load(cp + u * C) : 0;
        for (int h0 = 0; h0 < nblk; h0 += UB) {
            real vv[UB][BSZ];
#pragma unroll
            for (int u = 0; u < UB; u++)
                if (h0 + u < nblk) {
#pragma unroll
                    for (int b = 0; b < BSZ; b++) vv[u][b] = __builtin_nontemporal_load(v + (u * BSZ + b) * C);
                }
#pragma unroll
            for (int u = 0; u < UB; u++)
                cnxt[u] = h0 + UB + u < nblk ? (int)__builtin_nontemporal_load(cp + (UB + u) * C) : 0;
            real xx[UB][BSZ];
#pragma unroll
            for (int u = 0; u < UB; u++)
                if (h0 + u < nblk) {
                    const int c0 = ccur[u] + c_off;
#pragma unroll
                    for (int b = 0; b < BSZ; b++) xx[u][b] = x[c0 + b < nmax ? c0 + b : nmax];
                }
#pragma unroll
            for (int u = 0; u < UB; u++)
                if (h0 + u < nblk) {
#pragma unroll
                    for (int b = 0; b < BSZ; b++) acc += vv[u][b] * xx[u][b];
                }
#pragma unroll
            for (int u = 0; u < UB; u++) ccur[u] = cnxt[u];
            v += UB * BSZ * C;
            cp += UB * C;
        }
    }
    if (row < M) {
        if (ATOMIC) unsafeAtomicAdd(y + row, acc);
        else y[row] = acc;
    }
}

// Fourth form (round 4), for matrices that live in the caches: ONE CHUNK PER WORKGROUP of W wavefronts, wavefront w
// taking the blocks w, w + W, ... of every row of the chunk, partial sums folded through LDS in a fixed order.
// A cache-resident SpMV is not a bandwidth problem but a chain of dependent round trips with one wavefront per SIMD
// (cant class: 976 chunks for 1024 SIMDs): a row of 22 blocks of 3 is wider than the 10 blocks the whole-row form
// keeps in flight, so it ran the pipelined loop -- 6 batches, a dependent x gather each, 15.4 us where rocSPARSE's
// csrmv takes 13.1.  With W = 4 every wavefront has 6 blocks: chunk words -> column ids -> values + x, three trips
// whatever the width, and four times the wavefronts to hide them behind.  The traversal, the layout and the
// single-segment determinism (fixed summation order: blocks of wavefront 0, then the partials of 1 .. W-1) stay.
template <int BSZ, bool ATOMIC, int NB>
__global__ __launch_bounds__(512) void k_spmv_amb_split(real *__restrict__ y, const real *__restrict__ val,
                                                        const unsigned short *__restrict__ col,
                                                        const unsigned int *__restrict__ cl,
                                                        const int *__restrict__ cs,
                                                        const real *__restrict__ x,
                                                        const unsigned short *__restrict__ perm,
                                                        const unsigned short *__restrict__ perm_off,
                                                        int nchunks, int seg_size, int M, int N, int nb8)
{
    constexpr int C = 64;
    __shared__ real part[7][C];
    const int lb = nb8 > 0 ? (int)(blockIdx.x & 7) * nb8 + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (lb >= nchunks) return;  // (the whole workgroup: no barrier is skipped by part of it)
    const int W = (int)(blockDim.x >> 6), w = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    const int cs0 = cs[lb];
    const unsigned int length = cl[lb];
    const int nblk = (int)(length & SCL_BIT) + 1;
    const int c_off = (int)(length >> SCL_BORDER) * seg_size;
    const real *v = val + cs0 + lane;
    const unsigned short *cp = col + cs0 / BSZ + lane;
    const int nmax = N - 1;
    real acc = 0;
    for (int h0 = w; h0 < nblk; h0 += W * NB) {
        int cc[NB];
        real vv[NB][BSZ], xx[NB][BSZ];
#pragma unroll
        for (int u = 0; u < NB; u++) {
            const int h = h0 + u * W;
            cc[u] = h < nblk ? (int)__builtin_nontemporal_load(cp + h * C) : 0;
        }
#pragma unroll
        for (int u = 0; u < NB; u++) {
            const int h = h0 + u * W;
            if (h < nblk) {
#pragma unroll
                for (int b = 0; b < BSZ; b++) vv[u][b] = __builtin_nontemporal_load(v + (h * BSZ + b) * C);
            }
        }
#pragma unroll
        for (int u = 0; u < NB; u++) {
            if (h0 + u * W < nblk) {
                const int c0 = cc[u] + c_off;
#pragma unroll
                for (int b = 0; b < BSZ; b++) xx[u][b] = x[c0 + b < nmax ? c0 + b : nmax];
            }
        }
#pragma unroll
        for (int u = 0; u < NB; u++) {
            if (h0 + u * W < nblk) {
#pragma unroll
                for (int b = 0; b < BSZ; b++) acc += vv[u][b] * xx[u][b];
            }
        }
    }
    if (w > 0) part[w - 1][lane] = acc;
    __syncthreads();
    if (w == 0) {
        for (int q = 0; q < W - 1; q++) acc += part[q][lane];
        const int row = (int)__builtin_nontemporal_load(perm + (long long)lb * C + lane) + (int)perm_off[lb] * USHORT_MAX;
        if (row < M) {
            if (ATOMIC) unsafeAtomicAdd(y + row, acc);
            else y[row] = acc;
        }
    }
}

#ifdef NSPARSE_EXPERIMENTS
constexpr bool kSpmvExperiments = true;
#else
constexpr bool kSpmvExperiments = false;
#endif

template <int BSZ>
static void launch_bs(real *d_y, const sfAMB *mat, const real *d_x, int tb, hipStream_t st)
{
    const int rows = mat->c_size * mat->chunk;
    const int nb = ceil_div(rows, tb);
    const int nb8 = ceil_div(nb, 8);
    const bool atomic = mat->seg_num > 1;
    // XCD-aware block remap (each XCD walks one contiguous eighth of the chunks, so its private L2 sees one
    // window of x): pays only while the matrix lives in the 256 MiB Infinity Cache (cant class: 3 %).  A
    // matrix that streams from HBM reads eight far-apart regions at once under the remap and loses DRAM
    // locality: nlpkkt class 0.196 -> 0.178 ms without it (the bare value stream 0.164 -> 0.150).
    // NSPARSE_SPMV_NOREMAP=1 / NSPARSE_SPMV_REMAP=1 force either.
    static const int env_noremap = exp_env("NSPARSE_SPMV_NOREMAP", 0), env_remap = exp_env("NSPARSE_SPMV_REMAP", 0);
    const long long stream_bytes = (long long)mat->nnz * (long long)sizeof(real) + (long long)mat->nnz / mat->block_size * 2;
    const int no_remap = env_noremap || (!env_remap && stream_bytes > (128LL << 20));
    // (the forms below the whole-row kernel -- NSPARSE_SPMV_PIPE=0 / 1 / 2, NSPARSE_SPMV_PLAIN -- are round 1 / 2's kernels,
    //  kept for the before / after counters of tools/pmc_spmv.sh: instantiated in -DNSPARSE_EXPERIMENTS builds only
    //  (round 5: 200 of the 320 SpMV instantiations were reachable through those switches alone))
#ifdef NSPARSE_EXPERIMENTS
    static const int plain = exp_env("NSPARSE_SPMV_PLAIN", 0);
#else
    constexpr int plain = 0;
#endif
    const dim3 grid(no_remap ? nb : nb8 * 8), block(tb);
    const int nb8_arg = no_remap ? 0 : nb8;
#define NSP_GO(CC, AT)                                                                          \
    if constexpr (kSpmvExperiments) {                                                           \
        if (plain)                                                                              \
            hipLaunchKernelGGL((k_spmv_amb<BSZ, CC, AT, false>), grid, block, 0, st, d_y, mat->d_sellcs_val, \
                               mat->d_sellcs_col, mat->d_cl, mat->d_cs, d_x, mat->d_s_write_permutation, \
                               mat->d_s_write_permutation_offset, rows, (int)mat->seg_size, mat->M, mat->N, nb8_arg); \
    }                                                                                           \
    if (!plain)                                                                                 \
        hipLaunchKernelGGL((k_spmv_amb<BSZ, CC, AT, true>), grid, block, 0, st, d_y, mat->d_sellcs_val, \
                           mat->d_sellcs_col, mat->d_cl, mat->d_cs, d_x, mat->d_s_write_permutation, \
                           mat->d_s_write_permutation_offset, rows, (int)mat->seg_size, mat->M, mat->N, nb8_arg)
    // blocks of one batch of the pipelined form: as many as keep values + x within ~48 registers
    constexpr int UB = BSZ >= 12 ? 1 : (BSZ >= 6 ? 2 : (BSZ >= 3 ? 4 : 8));
#ifdef NSPARSE_EXPERIMENTS
    static const int pipe = exp_env("NSPARSE_SPMV_PIPE", 4);  // 0: first form, 1 / 2: pipelined, 4: whole row in flight
#else
    constexpr int pipe = 4;
#endif
#define NSP_PIPE(AT, UBX)                                                                       \
    hipLaunchKernelGGL((k_spmv_amb_pipe<BSZ, AT, UBX>), grid, block, 0, st, d_y, mat->d_sellcs_val, \
                       mat->d_sellcs_col, mat->d_cl, mat->d_cs, d_x, mat->d_s_write_permutation, \
                       mat->d_s_write_permutation_offset, rows, (int)mat->seg_size, mat->M, mat->N, nb8_arg, abl)
    static const int abl = exp_env("NSPARSE_SPMV_ABL", 0);
    // blocks of the whole-row form: values + x of NB blocks are live at once = NB * BSZ * 2 operands, two registers
    // each in double.  __launch_bounds__(1024) caps a lane at 128 VGPRs; past ~100 operand registers the compiler
    // spills (round 5: NB = 10 for BSZ 4 / 5 cost 148 / 316 B of scratch per lane, NB = 3 for BSZ 11 36 B -- an
    // HBM-bound kernel writing and re-reading its own operands).  BSZ <= 3 keep the measured settings (120 operand
    // registers at BSZ 3, 124 VGPRs, no scratch); above that NB comes from a budget of 96.  The float build uses the
    // same NB (half the registers).  tests/test_kernel_resources.py fails the build on any scratch.
    constexpr int NB = BSZ == 1 ? 24 : (BSZ == 2 ? 14 : (BSZ == 3 ? 10 : (96 / (BSZ * 4) > 0 ? 96 / (BSZ * 4) : 1)));
    // cache-resident matrix of few chunks whose rows are wider than the whole-row form keeps in flight: split the
    // rows over W wavefronts (k_spmv_amb_split).  NSPARSE_SPMV_SPLIT=1: W chosen from the average row width, =2 / 4 / 8:
    // that many; default 0 (off) until the kernel has been timed against the whole-row form on the device.  Experiments build only.
    static const int split_env = exp_env("NSPARSE_SPMV_SPLIT", 0);
    constexpr int NBS = BSZ >= 12 ? 1 : (BSZ >= 6 ? 2 : (BSZ >= 3 ? 6 : (BSZ == 2 ? 8 : 12)));
    if constexpr (kSpmvExperiments) {
    if (mat->chunk == 64 && pipe == 4 && !plain && split_env != 0 && !no_remap && mat->c_size <= 8192) {
        const double avgb = (double)mat->nnz / ((double)BSZ * (double)rows);  // average blocks per row (= chunk width)
        int W = 1;
        if (split_env == 2 || split_env == 4 || split_env == 8) W = split_env;
        else if (avgb > NB) W = avgb <= 2.0 * NBS ? 2 : (avgb <= 4.0 * NBS ? 4 : 8);
        if (W > 1) {
            const int nc = mat->c_size, ncb8 = ceil_div(nc, 8);
            if (atomic)
                hipLaunchKernelGGL((k_spmv_amb_split<BSZ, true, NBS>), dim3(ncb8 * 8), dim3(64 * W), 0, st, d_y, mat->d_sellcs_val,
                                   mat->d_sellcs_col, mat->d_cl, mat->d_cs, d_x, mat->d_s_write_permutation,
                                   mat->d_s_write_permutation_offset, nc, (int)mat->seg_size, mat->M, mat->N, ncb8);
            else
                hipLaunchKernelGGL((k_spmv_amb_split<BSZ, false, NBS>), dim3(ncb8 * 8), dim3(64 * W), 0, st, d_y, mat->d_sellcs_val,
                                   mat->d_sellcs_col, mat->d_cl, mat->d_cs, d_x, mat->d_s_write_permutation,
                                   mat->d_s_write_permutation_offset, nc, (int)mat->seg_size, mat->M, mat->N, ncb8);
            return;
        }
    }
    }  // (split-row form: experiments build only, until it has been timed against the whole-row form on the device)
    if (mat->chunk == 64 && pipe == 4 && !plain) {
        if (atomic)
            hipLaunchKernelGGL((k_spmv_amb_row<BSZ, true, NB, UB>), grid, block, 0, st, d_y, mat->d_sellcs_val,
                               mat->d_sellcs_col, mat->d_cl, mat->d_cs, d_x, mat->d_s_write_permutation,
                               mat->d_s_write_permutation_offset, rows, (int)mat->seg_size, mat->M, mat->N, nb8_arg);
        else
            hipLaunchKernelGGL((k_spmv_amb_row<BSZ, false, NB, UB>), grid, block, 0, st, d_y, mat->d_sellcs_val,
                               mat->d_sellcs_col, mat->d_cl, mat->d_cs, d_x, mat->d_s_write_permutation,
                               mat->d_s_write_permutation_offset, rows, (int)mat->seg_size, mat->M, mat->N, nb8_arg);
    } else if (mat->chunk == 64) {
        if constexpr (kSpmvExperiments) {
            if (pipe == 2 && !plain) {
                if (atomic) { NSP_PIPE(true, 2 * UB); } else { NSP_PIPE(false, 2 * UB); }
            } else if (pipe && !plain) {
                if (atomic) { NSP_PIPE(true, UB); } else { NSP_PIPE(false, UB); }
            } else {
                if (atomic) { NSP_GO(64, true); } else { NSP_GO(64, false); }
            }
        } else {
            // product builds instantiate the whole-row form only; reaching this means the invariant pipe == 4 && !plain broke
            set_error(-21, "AMB SpMV: kernel form not instantiated in this build", __FILE__, __LINE__);
        }
    } else {
        if (atomic) { NSP_GO(32, true); } else { NSP_GO(32, false); }
    }
#undef NSP_GO
#undef NSP_PIPE
}

static void launch(real *d_y, const sfAMB *mat, const real *d_x, const sfPlan *plan, hipStream_t st)
{
    // y = 0 (kernel_spmv_init_ans): needed when the kernel adds into y (several segments) or when
    // all-empty chunks were dropped (their rows are never visited).  With one segment and every
    // chunk present each row is stored exactly once, and the memset -- a third of the time of a
    // cache-resident SpMV -- is skipped.
    const bool every_row_stored = mat->seg_num == 1 && (long long)mat->c_size * mat->chunk >= (long long)mat->pad_M;
    static const int no_memset = exp_env("NSPARSE_SPMV_ABL", 0) & 4;  // diagnostics
    if (!every_row_stored && mat->M > 0 && !no_memset) NSP_CHECK(hipMemsetAsync(d_y, 0, sizeof(real) * (size_t)mat->M, st));
    if (mat->c_size <= 0) return;
    int tb = (int)plan->thread_block;
    if (tb < 64 || tb > 1024 || (tb & 63)) tb = 256;
    switch (mat->block_size) {
#define NSP_CASE(B) case B: launch_bs<B>(d_y, mat, d_x, tb, st); break;
        NSP_CASE(1) NSP_CASE(2) NSP_CASE(3) NSP_CASE(4) NSP_CASE(5) NSP_CASE(6) NSP_CASE(7)
        NSP_CASE(8) NSP_CASE(9) NSP_CASE(10) NSP_CASE(11) NSP_CASE(12) NSP_CASE(13) NSP_CASE(14)
        NSP_CASE(15) NSP_CASE(16) NSP_CASE(17) NSP_CASE(18) NSP_CASE(19) NSP_CASE(20)
#undef NSP_CASE
        default: set_error(-20, "AMB: block_size out of range", __FILE__, __LINE__);
    }
    NSP_LAUNCH_CHECK();
}

}  // namespace spmv
}  // namespace nsp

extern "C" {

void nsparse_spmv_amb_async(real *d_y, sfAMB *mat, real *d_x, sfPlan *plan, void *stream)
{
    nsp::spmv::launch(d_y, mat, d_x, plan, (hipStream_t)stream);
}

// Synchronous, on the null stream, so that it orders after whatever the caller queued there
// (the reference launches on the default stream and ends with cudaThreadSynchronize).
void sf_spmv_amb(real *d_y, sfAMB *mat, real *d_x, sfPlan *plan)
{
    nsp::ApiLock lk;
    nsp::TraceRange range("nsparse:spmv_amb");
    nsp::clear_error();
    nsp::Context &cx = nsp::ctx();
    if (cx.profiling) NSP_CHECK(hipEventRecord(cx.ev_t[4], 0));
    nsp::spmv::launch(d_y, mat, d_x, plan, 0);
    if (cx.profiling) NSP_CHECK(hipEventRecord(cx.ev_t[5], 0));
    NSP_CHECK(hipStreamSynchronize(0));
    if (cx.profiling) NSP_CHECK(hipEventElapsedTime(&nsp::spmv::g_last_ms, cx.ev_t[4], cx.ev_t[5]));
}

float nsparse_last_spmv_ms(void) { return nsp::spmv::g_last_ms; }

}  // extern "C"
