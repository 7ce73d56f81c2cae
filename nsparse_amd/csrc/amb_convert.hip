// amb_convert.hip -- CSR -> AMB (Adaptive Multi-level Blocking) on the device, plan search.
//
// Replaces (reference file:line):
//   sf_csr2amb                         cuda-c/src/conversion/convert_amb.cu:835-929
//   convert_amb_at and its ~25 kernels cuda-c/src/conversion/convert_amb.cu:22-833
//   evaluate_spmv (thread_block tuner) cuda-c/src/conversion/convert_amb.cu:556-600
//
// The FORMAT is the reference's, array for array: column segmentation so that a
// column id fits 16 bits, SELL-C-sigma with sigma = 32768, empty chunks dropped, one 16-bit
// base column per run of `block_size` consecutive columns, values zero padded, 16-bit output
// permutation with a per-chunk high part.  tests/ compare every array with the CPU oracle
// bit for bit.  The PIPELINE that produces it is different:
//
//  * no intermediate copies of the matrix.  The reference materialises a segmented CSR, an
//    int-column ELL, a ushort-column ELL and packed cl/cs arrays before blocking (4 extra
//    copies of nnz).  Here a (segment, row) "virtual row" is a contiguous run of the CSR row
//    (columns ascend inside a row, which the reference's blocking needs anyway; unsorted
//    input is detected and sorted once), so stages read col/val in place and only O(rows)
//    bookkeeping arrays exist.
//  * one rocprim::segmented_radix_sort_keys_desc over all (segment, sigma-window) pairs
//    instead of seg_num * ceil(M/32768) host-looped thrust::stable_sort_by_key calls
//    (convert_amb.cu:671-696; 6 K sorts for nlpkkt120).  Stability comes from a unique
//    composite key (length << 15 | reversed position), so the permutation is recovered from
//    the key itself and no payload array is sorted.
//  * chunk = 64 rows by default (one CDNA wavefront per chunk); chunk = 32 reproduces the
//    reference layout (mat->chunk = WARP = 32, convert_amb.cu:859).  Both are tested.
//  * the plan search uses the reference's own static footprint model (the `#undef AT` branch,
//    convert_amb.cu:783-797) by default: 20 cheap counting kernels per segment size instead
//    of 100 format builds and 500 timed SpMVs.  NSPARSE_AMB_TUNE=timed switches to the
//    reference's exhaustive timed search.
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>

#include <climits>
#include <cstring>

#include "internal.h"

namespace nsp {
namespace amb {

static int g_chunk = 64;

// ---- stage 1: virtual rows ---------------------------------------------------------
// One thread per CSR row.  Virtual row v = g * pad_M + i covers the entries of row i whose
// column / seg_size == g (set_segmented_nnz_num, convert_amb.cu:138-165).  With ascending
// columns that is the run [seg_start[v], seg_start[v] + nnz_num[v]).
__global__ __launch_bounds__(256) void k_virtual_rows(const int *__restrict__ rpt,
                                                      const int *__restrict__ col, int M, int pad_M,
                                                      int S, int *__restrict__ nnz_num,
                                                      int *__restrict__ seg_start, int *flags)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const int b = rpt[i], e = rpt[i + 1];
    int prev = -1, curg = -1, run = b;
    bool bad = false;
    for (int p = b; p < e; p++) {
        const int c = col[p];
        bad |= (c <= prev);
        prev = c;
        const int g = c / S;
        if (g != curg) {
            if (curg >= 0) {
                const long long v = (long long)curg * pad_M + i;
                nnz_num[v] = p - run;
                seg_start[v] = run;
            }
            curg = g;
            run = p;
        }
    }
    if (curg >= 0) {
        const long long v = (long long)curg * pad_M + i;
        nnz_num[v] = e - run;
        seg_start[v] = run;
    }
    if (bad) atomicOr(flags, 1);
}

// ---- stage 2: sigma sort -----------------------------------------------------------
// key = length << 15 | (32767 - position inside the window): descending order of the key is
// descending length with ties in ascending original position == the stable descending sort of
// convert_amb.cu:688-691.
__global__ __launch_bounds__(256) void k_make_keys(const int *__restrict__ nnz_num, long long R,
                                                   int M, int pad_M, int sig,
                                                   unsigned int *__restrict__ keys)
{
    const long long v = (long long)blockIdx.x * 256 + threadIdx.x;
    if (v >= R) return;
    const int i = (int)(v % pad_M);
    keys[v] = i < M ? (((unsigned)nnz_num[v] << 15) | (unsigned)(32767 - (i % sig))) : 0u;
}

__global__ __launch_bounds__(256) void k_sort_segments(int G, int nwin, int M, int pad_M, int sig,
                                                       int *__restrict__ beg, int *__restrict__ end)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= G * nwin) return;
    const int g = s / nwin, w = s % nwin;
    const long long base = (long long)g * pad_M;
    const long long hi = (long long)(w + 1) * sig;
    beg[s] = (int)(base + (long long)w * sig);
    end[s] = (int)(base + (hi < M ? hi : M));
}

__global__ __launch_bounds__(256) void k_decode_keys(const unsigned int *__restrict__ sorted,
                                                     const int *__restrict__ nnz_num, long long R,
                                                     int M, int pad_M, int sig, int do_sort,
                                                     int *__restrict__ len, int *__restrict__ perm)
{
    const long long v = (long long)blockIdx.x * 256 + threadIdx.x;
    if (v >= R) return;
    const int i = (int)(v % pad_M);
    if (i < M && do_sort) {
        const unsigned int k = sorted[v];
        len[v] = (int)(k >> 15);
        perm[v] = (int)(v - i + (long long)(i / sig) * sig + (32767 - (int)(k & 32767u)));
    } else {
        len[v] = i < M ? nnz_num[v] : 0;
        perm[v] = (int)v;
    }
}

// ---- stage 3: chunks ---------------------------------------------------------------
// width of every chunk = longest of its C virtual rows (set_cl, convert_amb.cu:46-64)
template <int C>
__global__ __launch_bounds__(256) void k_chunk_width(const int *__restrict__ len, long long R,
                                                     int *__restrict__ width, int *__restrict__ flag)
{
    const long long v = (long long)blockIdx.x * 256 + threadIdx.x;
    int w = v < R ? len[v] : 0;
#pragma unroll
    for (int o = C / 2; o >= 1; o >>= 1) {
        const int other = __shfl_xor(w, o);
        w = other > w ? other : w;
    }
    if (v < R && (v % C) == 0) {
        width[v / C] = w;
        flag[v / C] = w != 0;
    }
}

// non-empty chunks, packed (get_c_size .. set_packed_cl_cs, convert_amb.cu:301-386), and the
// output permutation (update / compress_write_permutation, compress_s_write_permutation
// :253-297): real row = virtual row - segment base OF THE POSITION.
__global__ __launch_bounds__(256) void k_pack(const int *__restrict__ len, const int *__restrict__ perm,
                                              const int *__restrict__ width, const int *__restrict__ gcs,
                                              long long R, int pad_M, int C,
                                              int *__restrict__ p_vrow, int *__restrict__ p_len,
                                              int *__restrict__ p_width, int *__restrict__ p_seg,
                                              int *__restrict__ wp, unsigned short *__restrict__ swp,
                                              unsigned short *__restrict__ swpo)
{
    const long long v = (long long)blockIdx.x * 256 + threadIdx.x;
    if (v >= R) return;
    const long long c = v / C;
    const int W = width[c];
    if (W == 0) return;
    const int t = (int)(v % C);
    const long long pc = gcs[c];
    const int seg = (int)(v / pad_M);
    const int r = perm[v] - seg * pad_M;
    p_vrow[pc * C + t] = perm[v];
    p_len[pc * C + t] = len[v];
    wp[pc * C + t] = r;
    swp[pc * C + t] = (unsigned short)(r % USHORT_MAX);
    if (t == 0) {
        swpo[pc] = (unsigned short)(r / USHORT_MAX);
        p_width[pc] = W;
        p_seg[pc] = seg;
    }
}

// ---- stage 4: blocking -------------------------------------------------------------
// The k-th 16-bit column of a lane: its own entry while k < own length, else the column the
// chunk's FIRST row holds at position k (padding rule of set_sellcs_col_val,
// convert_amb.cu:130-134, followed by set_ushort_col :313-346).
struct LaneView {
    const int *col;
    int own_start, own_len, first_start, seg_base;
    __device__ __forceinline__ int s(int k) const
    {
        return col[(k < own_len ? own_start : first_start) + k] - seg_base;
    }
};

// blocks per lane (set_blocked_cl, convert_amb.cu:388-413), chunk maximum over its C lanes
template <int C>
__global__ __launch_bounds__(256) void k_block_count(const int *__restrict__ col,
                                                     const int *__restrict__ seg_start,
                                                     const int *__restrict__ p_vrow,
                                                     const int *__restrict__ p_len,
                                                     const int *__restrict__ p_width,
                                                     const int *__restrict__ p_seg, int c_size, int S,
                                                     int bs, int *__restrict__ nblk,
                                                     unsigned long long *total_blocks)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < (long long)c_size * C;
    int nb = 0;
    if (live) {
        const int pc = (int)(i / C);
        const int W = p_width[pc];
        LaneView L{col, seg_start[p_vrow[i]], p_len[i], seg_start[p_vrow[(long long)pc * C]], p_seg[pc] * S};
        int base = L.s(0);
        nb = 1;
        for (int k = 1; k < W; k++) {
            const int sk = L.s(k);
            if (sk - base >= bs) { base = sk; nb++; }
        }
    }
#pragma unroll
    for (int o = C / 2; o >= 1; o >>= 1) {
        const int other = __shfl_xor(nb, o);
        nb = other > nb ? other : nb;
    }
    if (live && (i % C) == 0) {
        if (nblk) nblk[i / C] = nb * C * bs;  // value slots of the chunk (init_blocked_cs :431-445)
        if (total_blocks) atomicAdd(total_blocks, (unsigned long long)nb);
    }
}

// cl = (blocks - 1) | segment << 16 and the fill (set_blocked_cl :425-428,
// set_blocked_col_val :473-525)
template <int C>
__global__ __launch_bounds__(256) void k_block_fill(const int *__restrict__ col,
                                                    const real *__restrict__ val,
                                                    const int *__restrict__ seg_start,
                                                    const int *__restrict__ p_vrow,
                                                    const int *__restrict__ p_len,
                                                    const int *__restrict__ p_width,
                                                    const int *__restrict__ p_seg,
                                                    const int *__restrict__ cs,
                                                    const int *__restrict__ slots, int c_size, int S,
                                                    int bs, unsigned int *__restrict__ cl,
                                                    unsigned short *__restrict__ bcol,
                                                    real *__restrict__ bval)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)c_size * C) return;
    const int pc = (int)(i / C), t = (int)(i % C);
    const int W = p_width[pc];
    const int nblk = slots[pc] / (C * bs);
    if (t == 0) cl[pc] = (unsigned)(nblk - 1) | ((unsigned)p_seg[pc] << SCL_BORDER);
    const int own_start = seg_start[p_vrow[i]];
    const int own_len = p_len[i];
    LaneView L{col, own_start, own_len, seg_start[p_vrow[(long long)pc * C]], p_seg[pc] * S};
    unsigned short *bc = bcol + cs[pc] / bs + t;
    real *bv = bval + cs[pc] + t;
    int it = 0;
    for (int k = 0; k < nblk; k++) {
        if (it < W) {
            const int base = L.s(it);
            bc[(long long)k * C] = (unsigned short)base;
            bv[(long long)(k * bs) * C] = it < own_len ? val[own_start + it] : (real)0;
            it++;
            for (int h = 1; h < bs; h++) {
                real out = 0;
                if (it < W && L.s(it) - base == h) {
                    out = it < own_len ? val[own_start + it] : (real)0;
                    it++;
                }
                bv[(long long)(k * bs + h) * C] = out;
            }
        } else {
            bc[(long long)k * C] = (unsigned short)((L.s(W - 1) / bs) * bs);
            for (int h = 0; h < bs; h++) bv[(long long)(k * bs + h) * C] = 0;
        }
    }
}

// ---- host side ---------------------------------------------------------------------

static void exscan_int(const int *in, int *out, size_t n, hipStream_t st)
{
    size_t tb = 0;
    NSP_CHECK(rocprim::exclusive_scan(nullptr, tb, in, out, 0, n, rocprim::plus<int>(), st));
    void *tmp = dev_alloc(tb ? tb : 1);
    NSP_CHECK(rocprim::exclusive_scan(tmp, tb, in, out, 0, n, rocprim::plus<int>(), st));
    NSP_CHECK(hipStreamSynchronize(st));
    dev_free(tmp);
}

// everything of the conversion that does not depend on block_size
struct Ell {
    int M = 0, N = 0, pad_M = 0, C = 0, S = 0, G = 0, c_size = 0;
    const int *col = nullptr;   // column ids (sorted copy when the input was unsorted)
    const real *val = nullptr;
    int *own_col = nullptr;     // non-null when a sorted copy was made
    real *own_val = nullptr;
    int *seg_start = nullptr;   // [R]
    int *p_vrow = nullptr, *p_len = nullptr, *p_width = nullptr, *p_seg = nullptr;
    int *wp = nullptr;
    unsigned short *swp = nullptr, *swpo = nullptr;
    void release(bool keep_perm)
    {
        dev_free(seg_start);
        dev_free(p_vrow);
        dev_free(p_len);
        dev_free(p_width);
        dev_free(p_seg);
        dev_free(own_col);
        dev_free(own_val);
        if (!keep_perm) {
            dev_free(wp);
            dev_free(swp);
            dev_free(swpo);
        }
    }
};

template <int C>
static void launch_chunk_width(const int *len, long long R, int *width, int *flag, hipStream_t st)
{
    hipLaunchKernelGGL(k_chunk_width<C>, dim3(ceil_div(R, 256)), dim3(256), 0, st, len, R, width, flag);
}

static void build_ell(const sfCSR *csr, int S, int C, int sigma, Ell &E, hipStream_t st)
{
    Context &cx = ctx();
    const int M = csr->M, N = csr->N;
    const int pad_M = C * ceil_div(M, C);
    int G = ceil_div(N, S);
    if (G < 1) G = 1;
    const long long R = (long long)pad_M * G;
    if (R >= INT_MAX) set_error(-10, "AMB: rows * segments exceeds 2^31", __FILE__, __LINE__);
    E.M = M; E.N = N; E.pad_M = pad_M; E.C = C; E.S = S; E.G = G;
    E.col = csr->d_col;
    E.val = csr->d_val;

    int *nnz_num = (int *)dev_alloc(sizeof(int) * (size_t)R);
    E.seg_start = (int *)dev_alloc(sizeof(int) * (size_t)R);
    int *d_flags = cx.d_scratch + 200;
    NSP_CHECK(hipMemsetAsync(nnz_num, 0, sizeof(int) * (size_t)R, st));
    NSP_CHECK(hipMemsetAsync(E.seg_start, 0, sizeof(int) * (size_t)R, st));
    NSP_CHECK(hipMemsetAsync(d_flags, 0, sizeof(int), st));
    hipLaunchKernelGGL(k_virtual_rows, dim3(ceil_div(M, 256)), dim3(256), 0, st, csr->d_rpt, E.col, M,
                       pad_M, S, nnz_num, E.seg_start, d_flags);
    NSP_LAUNCH_CHECK();
    NSP_CHECK(hipMemcpyAsync(cx.h_pinned + 200, d_flags, sizeof(int), hipMemcpyDeviceToHost, st));
    NSP_CHECK(hipStreamSynchronize(st));
    if (cx.h_pinned[200] & 1) {
        // columns do not ascend inside some row: sort every row once (the reference would
        // silently produce a wrong matrix) and redo stage 1.
        E.own_col = (int *)dev_alloc(sizeof(int) * (size_t)csr->nnz);
        E.own_val = (real *)dev_alloc(sizeof(real) * (size_t)csr->nnz);
        size_t tb = 0;
        NSP_CHECK(rocprim::segmented_radix_sort_pairs(nullptr, tb, csr->d_col, E.own_col, csr->d_val,
                                                      E.own_val, (unsigned)csr->nnz, (unsigned)M,
                                                      csr->d_rpt, csr->d_rpt + 1, 0, 32, st));
        void *tmp = dev_alloc(tb ? tb : 1);
        NSP_CHECK(rocprim::segmented_radix_sort_pairs(tmp, tb, csr->d_col, E.own_col, csr->d_val,
                                                      E.own_val, (unsigned)csr->nnz, (unsigned)M,
                                                      csr->d_rpt, csr->d_rpt + 1, 0, 32, st));
        NSP_CHECK(hipStreamSynchronize(st));
        dev_free(tmp);
        E.col = E.own_col;
        E.val = E.own_val;
        NSP_CHECK(hipMemsetAsync(nnz_num, 0, sizeof(int) * (size_t)R, st));
        NSP_CHECK(hipMemsetAsync(E.seg_start, 0, sizeof(int) * (size_t)R, st));
        hipLaunchKernelGGL(k_virtual_rows, dim3(ceil_div(M, 256)), dim3(256), 0, st, csr->d_rpt, E.col,
                           M, pad_M, S, nnz_num, E.seg_start, d_flags);
        NSP_LAUNCH_CHECK();
    }

    // sigma sort (convert_amb.cu:667-696): SIGMA = min(SIGMA, M); windows over the REAL rows
    int sig = sigma;
    if (M < sig) sig = M;
    int *len = (int *)dev_alloc(sizeof(int) * (size_t)R);
    int *perm = (int *)dev_alloc(sizeof(int) * (size_t)R);
    const int grid_r = ceil_div(R, 256);
    unsigned int *keys = nullptr, *sorted = nullptr;
    if (sig > 1) {
        keys = (unsigned int *)dev_alloc(sizeof(unsigned int) * (size_t)R);
        sorted = (unsigned int *)dev_alloc(sizeof(unsigned int) * (size_t)R);
        const int nwin = ceil_div(M, sig);
        const int nseg = G * nwin;
        int *beg = (int *)dev_alloc(sizeof(int) * 2 * (size_t)nseg);
        int *end = beg + nseg;
        hipLaunchKernelGGL(k_make_keys, dim3(grid_r), dim3(256), 0, st, nnz_num, R, M, pad_M, sig, keys);
        hipLaunchKernelGGL(k_sort_segments, dim3(ceil_div(nseg, 256)), dim3(256), 0, st, G, nwin, M,
                           pad_M, sig, beg, end);
        NSP_LAUNCH_CHECK();
        int bits = 15;  // 15 position bits + length bits
        for (int m = csr->nnz_max > 0 ? csr->nnz_max : 1; m; m >>= 1) bits++;
        if (bits > 32 || csr->nnz_max <= 0) bits = 32;
        size_t tb = 0;
        NSP_CHECK(rocprim::segmented_radix_sort_keys_desc(nullptr, tb, keys, sorted, (unsigned)R,
                                                          (unsigned)nseg, beg, end, 0, bits, st));
        void *tmp = dev_alloc(tb ? tb : 1);
        NSP_CHECK(rocprim::segmented_radix_sort_keys_desc(tmp, tb, keys, sorted, (unsigned)R,
                                                          (unsigned)nseg, beg, end, 0, bits, st));
        NSP_CHECK(hipStreamSynchronize(st));
        dev_free(tmp);
        dev_free(beg);
    }
    hipLaunchKernelGGL(k_decode_keys, dim3(grid_r), dim3(256), 0, st, sorted, nnz_num, R, M, pad_M,
                       sig > 1 ? sig : 1, sig > 1 ? 1 : 0, len, perm);
    NSP_LAUNCH_CHECK();

    // chunk widths, non-empty flags, packed index
    const long long nchunk = R / C;
    int *width = (int *)dev_alloc(sizeof(int) * (size_t)(nchunk + 1));
    int *flag = (int *)dev_alloc(sizeof(int) * (size_t)(nchunk + 1));
    int *gcs = (int *)dev_alloc(sizeof(int) * (size_t)(nchunk + 1));
    NSP_CHECK(hipMemsetAsync(flag + nchunk, 0, sizeof(int), st));
    if (C == 64) launch_chunk_width<64>(len, R, width, flag, st);
    else launch_chunk_width<32>(len, R, width, flag, st);
    NSP_LAUNCH_CHECK();
    exscan_int(flag, gcs, (size_t)nchunk + 1, st);
    NSP_CHECK(hipMemcpy(&E.c_size, gcs + nchunk, sizeof(int), hipMemcpyDeviceToHost));

    const size_t cs_n = (size_t)(E.c_size > 0 ? E.c_size : 1);
    E.p_vrow = (int *)dev_alloc(sizeof(int) * cs_n * C);
    E.p_len = (int *)dev_alloc(sizeof(int) * cs_n * C);
    E.p_width = (int *)dev_alloc(sizeof(int) * cs_n);
    E.p_seg = (int *)dev_alloc(sizeof(int) * cs_n);
    E.wp = (int *)dev_alloc(sizeof(int) * cs_n * C);
    E.swp = (unsigned short *)dev_alloc(sizeof(unsigned short) * cs_n * C);
    E.swpo = (unsigned short *)dev_alloc(sizeof(unsigned short) * cs_n);
    hipLaunchKernelGGL(k_pack, dim3(grid_r), dim3(256), 0, st, len, perm, width, gcs, R, pad_M, C,
                       E.p_vrow, E.p_len, E.p_width, E.p_seg, E.wp, E.swp, E.swpo);
    NSP_LAUNCH_CHECK();
    NSP_CHECK(hipStreamSynchronize(st));
    dev_free(gcs);
    dev_free(flag);
    dev_free(width);
    dev_free(sorted);
    dev_free(keys);
    dev_free(perm);
    dev_free(len);
    dev_free(nnz_num);
}

// total blocks (sum over chunks of blocks per lane) for one block_size, without building
static unsigned long long count_blocks(const Ell &E, int bs, int *slots, hipStream_t st)
{
    Context &cx = ctx();
    unsigned long long *d_total = reinterpret_cast<unsigned long long *>(cx.d_scratch + 208);
    NSP_CHECK(hipMemsetAsync(d_total, 0, sizeof(unsigned long long), st));
    const int grid = ceil_div((long long)E.c_size * E.C, 256);
    if (grid > 0) {
        if (E.C == 64)
            hipLaunchKernelGGL(k_block_count<64>, dim3(grid), dim3(256), 0, st, E.col, E.seg_start, E.p_vrow,
                               E.p_len, E.p_width, E.p_seg, E.c_size, E.S, bs, slots, d_total);
        else
            hipLaunchKernelGGL(k_block_count<32>, dim3(grid), dim3(256), 0, st, E.col, E.seg_start, E.p_vrow,
                               E.p_len, E.p_width, E.p_seg, E.c_size, E.S, bs, slots, d_total);
        NSP_LAUNCH_CHECK();
    }
    unsigned long long h = 0;
    NSP_CHECK(hipMemcpyAsync(&h, d_total, sizeof(h), hipMemcpyDeviceToHost, st));
    NSP_CHECK(hipStreamSynchronize(st));
    return h;
}

// footprint model of convert_amb.cu:785-791 from the block count (64-bit arithmetic)
static long long footprint(const Ell &E, unsigned long long blocks, int bs)
{
    const long long w = (long long)sizeof(real);
    const long long nnzp = (long long)blocks * E.C * bs;
    long long f = 0;
    f += (nnzp / bs) * 2;
    f += nnzp * w;
    f += (long long)E.c_size * 4 * 2;
    f += (long long)E.c_size * E.C * 2 + (long long)E.c_size * 2;
    f += (long long)E.c_size * E.C * w * 2;
    f += (long long)E.M * w * 2;
    return f;
}

static void build_blocked(const Ell &E, int bs, sfAMB *mat, hipStream_t st)
{
    const size_t cs_n = (size_t)(E.c_size > 0 ? E.c_size : 1);
    int *slots = (int *)dev_alloc(sizeof(int) * (cs_n + 1));
    NSP_CHECK(hipMemsetAsync(slots, 0, sizeof(int) * (cs_n + 1), st));
    count_blocks(E, bs, slots, st);
    int *cs = (int *)dev_alloc(sizeof(int) * (cs_n + 1));
    exscan_int(slots, cs, (size_t)E.c_size + 1, st);
    int total = 0;
    NSP_CHECK(hipMemcpy(&total, cs + E.c_size, sizeof(int), hipMemcpyDeviceToHost));
    mat->nnz = total;
    mat->block_size = bs;
    mat->c_size = E.c_size;
    mat->d_cs = cs;
    mat->d_cl = (unsigned int *)dev_alloc(sizeof(unsigned int) * cs_n);
    mat->d_sellcs_col = (unsigned short *)dev_alloc(sizeof(unsigned short) * (size_t)(total / bs > 0 ? total / bs : 1));
    mat->d_sellcs_val = (real *)dev_alloc(sizeof(real) * (size_t)(total > 0 ? total : 1));
    const int grid = ceil_div((long long)E.c_size * E.C, 256);
    if (grid > 0) {
        if (E.C == 64)
            hipLaunchKernelGGL(k_block_fill<64>, dim3(grid), dim3(256), 0, st, E.col, E.val, E.seg_start,
                               E.p_vrow, E.p_len, E.p_width, E.p_seg, cs, slots, E.c_size, E.S, bs,
                               mat->d_cl, mat->d_sellcs_col, mat->d_sellcs_val);
        else
            hipLaunchKernelGGL(k_block_fill<32>, dim3(grid), dim3(256), 0, st, E.col, E.val, E.seg_start,
                               E.p_vrow, E.p_len, E.p_width, E.p_seg, cs, slots, E.c_size, E.S, bs,
                               mat->d_cl, mat->d_sellcs_col, mat->d_sellcs_val);
        NSP_LAUNCH_CHECK();
    }
    NSP_CHECK(hipStreamSynchronize(st));
    dev_free(slots);
    mat->d_write_permutation = E.wp;
    mat->d_s_write_permutation = E.swp;
    mat->d_s_write_permutation_offset = E.swpo;
}

// thread_block tuner: the reference times 64..1024 with TEST_NUM = 2 runs and keeps the
// fastest (evaluate_spmv, convert_amb.cu:556-600).
static float tune_thread_block(sfAMB *mat, real *d_x, real *d_y, sfPlan *plan, int reps)
{
    Context &cx = ctx();
    float best = 1e30f;
    size_t best_tb = 256;
    for (size_t tb = 64; tb <= 1024; tb *= 2) {
        sfPlan p = *plan;
        p.thread_block = tb;
        p.thread_grid = (size_t)ceil_div((long long)mat->chunk * mat->c_size, (long long)tb);
        // one warm-up, then `reps` launches back to back between two events: the steady-state
        // rate is what the caller's loop sees (the reference times the single second run,
        // TEST_NUM = 2; one isolated sample flips between candidates from run to run)
        float ms_best = 1e30f;
        nsparse_spmv_amb_async(d_y, mat, d_x, &p, nullptr);
        NSP_CHECK(hipEventRecord(cx.ev_t[6], 0));
        for (int i = 0; i < reps; i++) nsparse_spmv_amb_async(d_y, mat, d_x, &p, nullptr);
        NSP_CHECK(hipEventRecord(cx.ev_t[7], 0));
        NSP_CHECK(hipEventSynchronize(cx.ev_t[7]));
        NSP_CHECK(hipEventElapsedTime(&ms_best, cx.ev_t[6], cx.ev_t[7]));
        ms_best /= (float)(reps > 0 ? reps : 1);
        if (ms_best < best) { best = ms_best; best_tb = tb; }
    }
    plan->thread_block = best_tb;
    plan->thread_grid = (size_t)ceil_div((long long)mat->chunk * mat->c_size, (long long)best_tb);
    return best;
}

static void convert(sfAMB *mat, sfCSR *csr_in, real *d_x, sfPlan *plan)
{
    ApiLock api_lock;
    CallScope call_scope;
    TraceRange range("nsparse:csr2amb");
    clear_error();
    Context &cx = ctx();
    // nnz_max sizes the sort keys: a value outside (0, N] (a caller-built sfCSR) means unknown
    sfCSR csr_chk = *csr_in;
    if (!(csr_chk.nnz_max > 0 && csr_chk.nnz_max <= csr_chk.N)) csr_chk.nnz_max = 0;
    sfCSR *csr = &csr_chk;
    if (csr->M <= 0) {
        // no rows (the empty block of a row-sharded run): an AMB with no chunks; sf_spmv_amb on it
        // writes nothing.  Every d_* array still exists, so release_amb works as usual.
        memset(mat, 0, sizeof(*mat));
        mat->N = csr->N;
        mat->chunk = g_chunk;
        mat->block_size = 1;
        mat->SIGMA = SHORT_MAX;
        mat->seg_size = USHORT_MAX;
        mat->seg_num = (size_t)(csr->N > 0 ? ceil_div(csr->N, USHORT_MAX) : 1);
        mat->group_num_col = (int)mat->seg_num;
        mat->matrix_name = csr->matrix_name;
        mat->d_cs = (int *)dev_alloc(sizeof(int));
        mat->d_cl = (unsigned int *)dev_alloc(sizeof(unsigned int));
        mat->d_sellcs_col = (unsigned short *)dev_alloc(sizeof(unsigned short));
        mat->d_sellcs_val = (real *)dev_alloc(sizeof(real));
        mat->d_write_permutation = (int *)dev_alloc(sizeof(int));
        mat->d_s_write_permutation = (unsigned short *)dev_alloc(sizeof(unsigned short));
        mat->d_s_write_permutation_offset = (unsigned short *)dev_alloc(sizeof(unsigned short));
        plan->isPlan = TRUE;
        plan->seg_size = mat->seg_size;
        plan->seg_num = mat->seg_num;
        plan->block_size = 1;
        plan->SIGMA = mat->SIGMA;
        plan->thread_block = 256;
        plan->thread_grid = 0;
        return;
    }
    NSP_CHECK(hipDeviceSynchronize());  // inputs may have been produced on any stream
    hipStream_t st = cx.stream[0];
    const int C = g_chunk;
    memset(mat, 0, sizeof(*mat));
    mat->M = csr->M;
    mat->N = csr->N;
    mat->chunk = C;
    mat->pad_M = C * ceil_div(csr->M, C);
    mat->nnz_max = csr->nnz_max;
    mat->matrix_name = csr->matrix_name;
    mat->SIGMA = SHORT_MAX;

    real *d_y = (real *)dev_alloc(sizeof(real) * (size_t)(csr->M + WARP));
    const char *tune = getenv("NSPARSE_AMB_TUNE");
    const bool timed = tune && strcmp(tune, "timed") == 0;

    long long seg_size;
    int bs;
    if (plan->isPlan == TRUE) {
        seg_size = (long long)plan->seg_size;
        bs = plan->block_size;
        if (seg_size > USHORT_MAX) seg_size = USHORT_MAX;
        if (seg_size < 1) seg_size = 1;
        if (bs < 1 || bs > MAX_BLOCK_SIZE) bs = 1;
    } else {
        // candidate segment sizes of sf_csr2amb (convert_amb.cu:878-892)
        long long cand[5];
        const int ncand = csr->N < 128 * 1024 ? 5 : 1;
        cand[0] = 64 * 1024;
        for (int i = 1; i < ncand; i++) cand[i] = csr->N < 100 ? i : (long long)i * 1024;
        long long best = LLONG_MAX;
        float best_ms = 1e30f;
        seg_size = cand[0];
        bs = 1;
        for (int s = 0; s < ncand; s++) {
            Ell E;
            build_ell(csr, (int)cand[s], C, mat->SIGMA, E, st);
            for (int b = 1; b <= MAX_BLOCK_SIZE; b++) {
                if (!timed) {
                    const long long f = footprint(E, count_blocks(E, b, nullptr, st), b);
                    if (best > f) { best = f; seg_size = cand[s]; bs = b; }
                } else {
                    sfAMB trial;
                    memset(&trial, 0, sizeof(trial));
                    trial.M = csr->M; trial.N = csr->N; trial.chunk = C; trial.pad_M = mat->pad_M;
                    trial.seg_size = (size_t)cand[s];
                    trial.seg_num = (size_t)E.G;
                    trial.group_num_col = E.G;
                    build_blocked(E, b, &trial, st);
                    sfPlan p = *plan;
                    p.seg_size = (size_t)cand[s];
                    p.block_size = b;
                    const float ms = tune_thread_block(&trial, d_x, d_y, &p, TEST_NUM - 1);
                    if (ms < best_ms) { best_ms = ms; seg_size = cand[s]; bs = b; }
                    dev_free(trial.d_cs);
                    dev_free(trial.d_cl);
                    dev_free(trial.d_sellcs_col);
                    dev_free(trial.d_sellcs_val);
                }
            }
            E.release(false);
        }
    }

    Ell E;
    build_ell(csr, (int)seg_size, C, mat->SIGMA, E, st);
    mat->seg_size = (size_t)seg_size;
    mat->seg_num = (size_t)E.G;
    mat->group_num_col = E.G;
    build_blocked(E, bs, mat, st);
    E.release(true);

    plan->isPlan = TRUE;
    plan->seg_size = (size_t)seg_size;
    plan->seg_num = (size_t)E.G;
    plan->block_size = bs;
    plan->SIGMA = mat->SIGMA;
    plan->thread_block = 256;
    plan->thread_grid = (size_t)ceil_div((long long)mat->chunk * mat->c_size, 256);
    if (d_x && mat->c_size > 0) tune_thread_block(mat, d_x, d_y, plan, 10);
    dev_free(d_y);
    NSP_CHECK(hipDeviceSynchronize());
}

}  // namespace amb
}  // namespace nsp

extern "C" {

int nsparse_set_amb_chunk(int chunk)
{
    if (chunk == 32 || chunk == 64) nsp::amb::g_chunk = chunk;
    return nsp::amb::g_chunk;
}

long long nsparse_amb_footprint_bytes(const sfAMB *mat)
{
    const long long w = (long long)sizeof(real);
    long long f = 0;
    f += ((long long)mat->nnz / mat->block_size) * 2;
    f += (long long)mat->nnz * w;
    f += (long long)mat->c_size * 4 * 2;
    f += (long long)mat->c_size * mat->chunk * 2 + (long long)mat->c_size * 2;
    f += (long long)mat->c_size * mat->chunk * w * 2;
    f += (long long)mat->M * w * 2;
    return f;
}

void sf_csr2amb(sfAMB *mat, sfCSR *csr_mat, real *d_x, sfPlan *plan) { nsp::amb::convert(mat, csr_mat, d_x, plan); }

}  // extern "C"
