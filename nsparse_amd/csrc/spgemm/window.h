// spgemm/window.h -- dense-window and bit-window rows (bins 6-10).
// Part of the spgemm_hash.hip translation unit (kernels are launched from its host code).
#pragma once
#include "common.h"

namespace nsp {
namespace spgemm {

#ifndef NSP_EMU
extern __shared__ __attribute__((aligned(16))) unsigned char nsp_dyn_lds[];
#else
#define nsp_dyn_lds (::emu::t_dyn_lds)  // tests/emu: the dynamic LDS of the workgroup the CPU emulation is running
#endif

// ===================================================================================
//  dense-window rows (bins 6..8)
// ===================================================================================
// The columns a C row can contain lie in [lo, lo + span) (union of the column windows of the
// B rows it touches, computed in k_row_products).  When span fits LDS the row needs no hash
// table: symbolic = one byte flag per column, set with a plain LDS store (idempotent, no
// atomic, no return value to wait for), count = popcount of the flags; numeric = one real per
// column accumulated with a no-return LDS atomic add, emitted in ascending order by scanning
// the flags with ballot/popcount -- no compaction pass and no sort.  The reference has no such
// path (48 KB of shared memory per block on its target); on CDNA4's 160 KiB it covers every row
// of a banded / FEM matrix.  Wide-window rows (graphs) stay on the hash bins.

template <int BS, int SPAN_MAX>
__global__ __launch_bounds__(BS) void k_sym_dense(const int *__restrict__ arpt, const int *__restrict__ acol,
                                                  const int *__restrict__ brpt, const int *__restrict__ bcol,
                                                  const int *__restrict__ row_perm,
                                                  const int *__restrict__ row_prod,
                                                  const int *__restrict__ row_maxb,
                                                  const int *__restrict__ row_lo,
                                                  const int *__restrict__ row_span,
                                                  int *__restrict__ row_nz, int bin_off, int bin_size,
                                                  int bnnz, const int *__restrict__ bm_off,
                                                  unsigned int *__restrict__ bm,
                                                  int *__restrict__ row_span_num,
                                                  const unsigned char *__restrict__ btwin,
                                                  const int4 *__restrict__ desc)
{
    // desc != nullptr: k_setup_tail left a record per listed row, in list order (fused.h): the row words
    // come back in one round trip instead of list -> row words.
    // flags: dynamic LDS sized by the widest window actually in the bin, (span/4 + 8) words -- a
    // bin spans a 4x range of windows and a static array for its upper end would cost occupancy
    unsigned int *flag4 = reinterpret_cast<unsigned int *>(nsp_dyn_lds);
    __shared__ int2 s_ext[BS];
    __shared__ int s_nz;
    const int slot = xcd_row_slot(bin_size);
    if (slot < 0) return;
    int rid, lo, span, a_beg, a_end, prod, maxb, bmo = 0, bw = 0;
    if (desc) {
        const int4 d0 = desc[3 * (bin_off + slot)], d1 = desc[3 * (bin_off + slot) + 1],
                   d2 = desc[3 * (bin_off + slot) + 2];
        rid = d0.x, lo = d0.y, span = d0.z, maxb = d0.w;
        a_beg = d1.x, a_end = d1.y, prod = d1.z, bmo = d1.w;
        bw = d2.x;
    } else {
        rid = row_perm[bin_off + slot];
        lo = row_lo[rid];
        span = row_span[rid];
        a_beg = arpt[rid], a_end = arpt[rid + 1];
        prod = row_prod[rid];
        maxb = row_maxb[rid];
        if (bm != nullptr) {
            bmo = bm_off[rid];
            bw = bm_off[rid + 1] - bmo;
        }
    }
    const int words = (span + 3) >> 2;
    {
        uint4 *f4 = reinterpret_cast<uint4 *>(flag4);
        const uint4 z = make_uint4(0, 0, 0, 0);
        for (int i = threadIdx.x; i < (words + 3) / 4 + 2; i += BS) f4[i] = z;  // + bitmap tail
    }
    if (threadIdx.x == 0) s_nz = 0;
    __syncthreads();
    unsigned char *flag = reinterpret_cast<unsigned char *>(flag4);
    // lanes past the end of a B row store into scratch bytes behind the flags instead of being
    // masked off: a select costs less than an exec-mask save / restore per element
    const int dummy = 4 * words + 32 + (threadIdx.x & 31);
    const int g = group_width(prod, a_end - a_beg, BS, maxb, VWS);
    walk_products<BS, false, VWS>(acol, (const real *)nullptr, brpt, bcol, (const real *)nullptr, bnnz, a_beg,
                             a_end, g, s_ext, (real *)nullptr,
                             [&](const IVecS &k, const RVecS &, int n, real) {
#pragma unroll
                                 for (int i = 0; i < VWS; i++) flag[i < n ? k.v[i] - lo : dummy] = 1;
                             },
                             (DeferList<false, 32> *)nullptr, 0x7fffffff, btwin);
    __syncthreads();
    int cnt = 0;
    for (int i = threadIdx.x; i < words; i += BS) cnt += __popc(flag4[i] & 0x01010101u);
    cnt = wave_sum(cnt);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&s_nz, cnt);
    // Hand the structure to the numeric phase: 1 bit per column of the window, 32 flag bytes
    // -> one word.  The numeric dense kernel then needs no flags of its own (one LDS atomic
    // per product instead of an atomic and a store) and no sort.
    if (bm != nullptr) {
        unsigned int *dst = bm + bmo;
        for (int wi = threadIdx.x; wi < bw; wi += BS) {
            unsigned int bits = 0;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const unsigned int x = flag4[wi * 8 + q] & 0x01010101u;
                bits |= ((x * 0x01020408u) >> 24) << (4 * q);
            }
            dst[wi] = bits;
        }
        if (threadIdx.x == 0) row_span_num[rid] = bw > 0 ? span : 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) row_nz[rid] = s_nz;
}

// Symbolic for rows with many products and a wide window: one BIT per column of the window in
// LDS (128 KiB cover 2^20 columns), set with a no-return LDS atomic OR.  Replaces the 32768-key
// hash table (1 workgroup per CU, CAS with return per product) and the try-in-LDS / global
// table detour for every row of a matrix with up to a million columns.
template <int BS, int WORDS_MAX>
__global__ __launch_bounds__(BS) void k_sym_bits(const int *__restrict__ arpt, const int *__restrict__ acol,
                                                 const int *__restrict__ brpt, const int *__restrict__ bcol,
                                                 const int *__restrict__ row_perm,
                                                 const int *__restrict__ row_prod,
                                                  const int *__restrict__ row_maxb,
                                                 const int *__restrict__ row_lo,
                                                 const int *__restrict__ row_span,
                                                 int *__restrict__ row_nz, int bin_off, int bin_size,
                                                 int bnnz, BinState *bs = nullptr, int *__restrict__ tcol = nullptr,
                                                 long long *__restrict__ list_off = nullptr, long long list_work = 0,
                                                 int dens = 0, int tiled_w = 0)
{
    // tcol != nullptr: the row's columns are written out as a sorted list for the numeric phase (common.h:
    // bits_to_list), list_off[rid] = where
    __shared__ __attribute__((aligned(16))) unsigned int bits[WORDS_MAX];
    __shared__ int2 s_ext[BS];
    __shared__ int s_wsum[BS / 64];
    __shared__ long long s_off;
    const int slot = xcd_row_slot(bin_size);
    if (slot < 0) return;
    const int rid = row_perm[bin_off + slot];
    const int a_beg = arpt[rid], a_end = arpt[rid + 1];
    const int g = group_width(row_prod[rid], a_end - a_beg, BS, row_maxb[rid]);
    // a list for the numeric listed kernel?  One piece: decided with the exact count, once it is known; several
    // pieces: up front, with the bound (common.h: list_wanted)
    const int np = row_prod[rid], sp = row_span[rid];
    const int lcap = np < sp ? np : sp;
    const bool one_piece = sp <= WORDS_MAX * 32;
    // list_work < 0: every row that the numeric ranked kernel will take (more non-zeros than the hash bins hold,
    // not dense enough for the dense tiles: the rule of k_num_tiled / k_num_ranked); > 0: the listed kernel's rows
    auto wanted = [&](int n, bool exact) {
        if (list_work >= 0) return list_wanted(n, np, list_work);
        if (n <= kListMinNnz) return false;
        const bool to_tiled = exact && dens > 0 && (long long)n * dens >= sp && sp <= 32 * tiled_w;
        return !to_tiled;
    };
    bool listing = tcol != nullptr && !one_piece && wanted(lcap, false);
    if (listing && threadIdx.x == 0) {
        s_off = (long long)atomicAdd(&bs->list_cursor, (unsigned long long)lcap);
        list_off[rid] = s_off;
    }
    int cnt = 0;
    // A window wider than the bitmap is covered in pieces: every piece walks all products again
    // and keeps the columns that fall into it (no cursors: the walk is a fraction of what a hash
    // table filled to the brim costs, and the row need not be sorted).
    const int row_hi = row_lo[rid] + row_span[rid];
    for (int lo = row_lo[rid]; lo < row_hi; lo += WORDS_MAX * 32) {
        const int cols = row_hi - lo < WORDS_MAX * 32 ? row_hi - lo : WORDS_MAX * 32;
        const int words = (cols + 31) >> 5;
        {
            uint4 *b4 = reinterpret_cast<uint4 *>(bits);
            const uint4 z = make_uint4(0, 0, 0, 0);
            for (int i = threadIdx.x; i < (words + 3) / 4; i += BS) b4[i] = z;
        }
        __syncthreads();
        walk_products<BS, false>(acol, (const real *)nullptr, brpt, bcol, (const real *)nullptr, bnnz, a_beg,
                                 a_end, g, s_ext, (real *)nullptr,
                                 [&](const IVec &k, const RVecT<1> &, int n, real) {
#pragma unroll
                                     for (int i = 0; i < VW; i++)
                                         if (i < n) {
                                             const unsigned int idx = (unsigned int)(k.v[i] - lo);
                                             if (idx < (unsigned int)cols) atomicOr(bits + (idx >> 5), 1u << (idx & 31));
                                         }
                                 });
        __syncthreads();  // (also orders s_off)
        if (tcol != nullptr && one_piece) {
            const int n1 = bits_to_list<BS, false>(bits, words, lo, (int *)nullptr, s_wsum);  // count first
            if (wanted(n1, true)) {
                if (threadIdx.x == 0) {
                    s_off = (long long)atomicAdd(&bs->list_cursor, (unsigned long long)n1);
                    list_off[rid] = s_off;
                }
                __syncthreads();
                bits_to_list<BS, false>(bits, words, lo, tcol + s_off, s_wsum);
            }
            cnt += n1;
        } else {
            cnt += bits_to_list<BS, false>(bits, words, lo, listing ? tcol + s_off + cnt : (int *)nullptr, s_wsum);
        }
    }
    if (threadIdx.x == 0) row_nz[rid] = cnt;
}

// lanes per B row for the one-entry-per-lane walk of the node-block kernel (block.h): the smallest power
// of two that covers the longest B row of the C row in CH chunks, at least 8 (a quarter of a 128-byte line
// of columns), at most 64
__device__ __forceinline__ int lean_group(int maxb, int CH)
{
    int g = 8;
    while (g < 64 && g * CH < maxb) g <<= 1;
    return g;
}

template <int BS, int SPAN_MAX, int MODE>
__global__ __launch_bounds__(BS) void k_num_dense(const int *__restrict__ arpt, const int *__restrict__ acol,
                                                  const real *__restrict__ aval,
                                                  const int *__restrict__ brpt, const int *__restrict__ bcol,
                                                  const real *__restrict__ bval,
                                                  const int *__restrict__ crpt, int *__restrict__ ccol,
                                                  real *__restrict__ cval,
                                                  const int *__restrict__ row_perm,
                                                  const int *__restrict__ row_prod,
                                                  const int *__restrict__ row_maxb,
                                                  const int *__restrict__ row_lo,
                                                  const int *__restrict__ row_span, int bin_off,
                                                  int bin_size, int bnnz,
                                                  const int *__restrict__ bm_off,
                                                  const unsigned int *__restrict__ bm)
{
    // MODE 1: full call -- the column structure of the row comes from the bitmap written by
    //         k_sym_dense; columns and values are emitted in ascending order.
    // MODE 2: numeric-only re-run -- C.col exists; values are gathered at its columns.
    constexpr int NW = BS / 64;
    acc_t *dense = reinterpret_cast<acc_t *>(nsp_dyn_lds);  // dynamic: (widest window of the bin + 4) values
    __shared__ int2 s_ext[BS];
    __shared__ real s_av[BS];
    __shared__ int s_wcnt[NW];
    const int slot = xcd_row_slot(bin_size);
    if (slot < 0) return;
    const int rid = row_perm[bin_off + slot];
    const int off = crpt[rid];
    const int lo = row_lo[rid];
    const int span = row_span[rid];
    // The VW entries a lane holds have consecutive columns inside a run, so one atomic
    // instruction sees columns of stride VW across the lanes: the value of column idx lives at
    // (idx & 3) * Q + (idx >> 2), which turns that stride into consecutive 8-byte slots.
    const int Q = (span + 3) >> 2;
    auto slot_of = [&](int idx) { return __mul24(idx & 3, Q) + (idx >> 2); };
    for (int i = threadIdx.x; i < 4 * Q; i += BS) dense[i] = 0;
    __syncthreads();
    const int a_beg = arpt[rid], a_end = arpt[rid + 1];
    const int g = group_width(row_prod[rid], a_end - a_beg, BS, row_maxb[rid]);
    walk_products<BS, true>(acol, aval, brpt, bcol, bval, bnnz, a_beg, a_end, g, s_ext, s_av,
                            [&](const IVec &k, const RVec &v, int n, real sc) {
#pragma unroll
                                for (int i = 0; i < VW; i++)
                                    if (i < n) unsafeAtomicAdd(dense + slot_of(k.v[i] - lo), (acc_t)(sc * v.v[i]));
                            });
    __syncthreads();
    if (MODE == 2) {
        const int n = crpt[rid + 1] - off;
        for (int p = threadIdx.x; p < n; p += BS) {
            const int idx = ccol[off + p] - lo;
            cval[off + p] = (real)dense[slot_of(idx)];
        }
        return;
    }
    // Ordered emission: wavefront w owns the column range [w*R, (w+1)*R), R a multiple of 64 and
    // at most 4096, so the range is at most 128 bitmap words: every lane fetches one or two of them (a single
    // coalesced load -- the loop below then runs on registers, where it used to wait for a global
    // load per 64 columns), the count is a wave sum of popcounts and the 64-column masks come from
    // readlane instead of ballots.
    constexpr int WPL = (SPAN_MAX / NW + 2047) / 2048;  // bitmap words per lane (1 or 2)
    static_assert(WPL <= 2, "at most two bitmap words per lane");
    const unsigned int *bits = bm + bm_off[rid];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int R = ((span + NW * 64 - 1) / (NW * 64)) * 64;
    const int rb = w * R, re = rb + R < span ? rb + R : span;
    const int nwords = (span + 31) >> 5, w0 = rb >> 5, rw = R >> 5;
    unsigned int word[WPL];
    int cnt = 0;
#pragma unroll
    for (int h = 0; h < WPL; h++) {
        const int wi = 64 * h + lane;
        word[h] = (wi < rw && w0 + wi < nwords) ? bits[w0 + wi] : 0u;  // bits past span are 0
        cnt += __popc(word[h]);
    }
    cnt = wave_sum(cnt);
    if (lane == 0) s_wcnt[w] = cnt;
    __syncthreads();
    int pos = off;
    for (int u = 0; u < w; u++) pos += s_wcnt[u];
#pragma unroll
    for (int h = 0; h < WPL; h++) {
        for (int j = 32 * h; j < 32 * (h + 1) && rb + 64 * j < re; j++) {
            const int l2 = 2 * (j - 32 * h);
            const unsigned long long m =
                (unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)word[h], l2) |
                ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)word[h], l2 + 1) << 32);
            if (m == 0) continue;  // wave-uniform
            if ((m >> lane) & 1ull) {
                const int idx = rb + 64 * j + lane;
                const int p = pos + __popcll(m & ((1ull << lane) - 1ull));
                ccol[p] = lo + idx;
                cval[p] = (real)dense[slot_of(idx)];
            }
            pos += __popcll(m);
        }
    }
}

}  // namespace spgemm
}  // namespace nsp
