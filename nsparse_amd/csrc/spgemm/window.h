// spgemm/window.h -- dense-window and bit-window rows (bins 6-10).
// Part of the spgemm_hash.hip translation unit (kernels are launched from its host code).
#pragma once
#include "common.h"

namespace nsp {
namespace spgemm {

extern __shared__ __attribute__((aligned(16))) unsigned char nsp_dyn_lds[];

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
                                                  const unsigned char *__restrict__ btwin)
{
    // flags: dynamic LDS sized by the widest window actually in the bin, (span/4 + 8) words -- a
    // bin spans a 4x range of windows and a static array for its upper end would cost occupancy
    unsigned int *flag4 = reinterpret_cast<unsigned int *>(nsp_dyn_lds);
    __shared__ int2 s_ext[BS];
    __shared__ int s_nz;
    const int slot = xcd_row_slot(bin_size);
    if (slot < 0) return;
    const int rid = row_perm[bin_off + slot];
    const int lo = row_lo[rid];
    const int span = row_span[rid];
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
    const int a_beg = arpt[rid], a_end = arpt[rid + 1];
    const int g = group_width(row_prod[rid], a_end - a_beg, BS, row_maxb[rid], VWS);
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
        const int bw = bm_off[rid + 1] - bm_off[rid];
        unsigned int *dst = bm + bm_off[rid];
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
                                                 int bnnz)
{
    __shared__ __attribute__((aligned(16))) unsigned int bits[WORDS_MAX];
    __shared__ int2 s_ext[BS];
    __shared__ int s_nz;
    const int slot = xcd_row_slot(bin_size);
    if (slot < 0) return;
    const int rid = row_perm[bin_off + slot];
    const int a_beg = arpt[rid], a_end = arpt[rid + 1];
    const int g = group_width(row_prod[rid], a_end - a_beg, BS, row_maxb[rid]);
    if (threadIdx.x == 0) s_nz = 0;
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
        __syncthreads();
        for (int i = threadIdx.x; i < words; i += BS) cnt += __popc(bits[i]);
        __syncthreads();
    }
    cnt = wave_sum(cnt);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&s_nz, cnt);
    __syncthreads();
    if (threadIdx.x == 0) row_nz[rid] = s_nz;
}

// ---- lean product walk for the numeric window kernels -----------------------------------------
// rocprofv3 / ISA of the first k_num_dense: ~120 instructions per step of 4 products per lane -- the
// generic walk's state machine, four exec-mask regions (one per vector element), six instructions
// of slot arithmetic per product and two dozen register moves of its double buffer -- against an
// LDS atomic rate (tools/lds_atomic) that would allow 4-15x more products per clock.  The kernel was
// VALU-issue-bound.  Here every lane takes ONE entry of a B row per chunk: a group of G lanes reads
// G consecutive entries (coalesced dword / qword loads), so one atomic instruction sees consecutive
// columns -- consecutive 8-byte slots of a plainly indexed window, no slot swizzle -- and a partial
// chunk is one exec mask, not one per element.  The A entries of the row are parked in LDS as
// 16-byte records {first, end, value}; group q walks a CONTIGUOUS range of them (the rows of B that
// belong to one mesh node have the same columns: interleaved, neighbouring groups would add to the
// same slots in the same instruction), up to CH chunks of an entry in flight, the next entry's
// loads issued before the current one is accumulated (two register sets, unrolled by hand: no moves).
struct __attribute__((aligned(16))) LeanEnt {
    int kb, ke;
    acc_t av;  // aval as acc_t: 16 bytes in both builds
};

template <int BS, int CH>
__device__ __forceinline__ void lean_accumulate(const int *__restrict__ acol, const real *__restrict__ aval,
                                                const int *__restrict__ brpt, const int *__restrict__ bcol,
                                                const real *__restrict__ bval, int bnnz, int a_beg, int a_end,
                                                int lo, acc_t *dense, LeanEnt *s_ent, int G, int abl = 0)
{
    const int lg = 31 - __clz(G);
    const int gid = (int)threadIdx.x >> lg, gl = (int)threadIdx.x & (G - 1);
    const int NG = BS >> lg;
    const unsigned last = (unsigned)(bnnz - 1);
    acc_t *win = dense - lo;
    struct Buf {
        int c[CH];
        real v[CH];
    };
    auto issue = [&](const LeanEnt &E, Buf &b) {
#pragma unroll
        for (int q = 0; q < CH; q++) {
            const unsigned k = (unsigned)(E.kb + gl + q * G);
            const unsigned kk = k < last ? k : last;  // past the row: a valid address, masked at the add
            if (abl & 2) {  // ablation: no loads of B
                b.c[q] = lo + (int)(kk & 1023);
                b.v[q] = (real)kk;
            } else {
                b.c[q] = bcol[kk];
                b.v[q] = bval[kk];
            }
        }
    };
    acc_t abl_sum = 0;
    auto consume = [&](const LeanEnt &E, const Buf &b) {
#pragma unroll
        for (int q = 0; q < CH; q++)
            if (E.kb + gl + q * G < E.ke) {
                if (abl & 1) abl_sum += (acc_t)((real)E.av * b.v[q]) + (acc_t)b.c[q];  // ablation: no LDS atomics
                else unsafeAtomicAdd(win + b.c[q], (acc_t)((real)E.av * b.v[q]));
            }
        // B rows longer than CH chunks (rare: G is chosen from the longest row): plain loop
        for (int k = E.kb + gl + CH * G; k < E.ke; k += G) unsafeAtomicAdd(win + bcol[k], (acc_t)((real)E.av * bval[k]));
    };
    for (int a0 = a_beg; a0 < a_end; a0 += BS) {
        LeanEnt me;
        me.kb = 0, me.ke = 0, me.av = 0;
        const int j = a0 + (int)threadIdx.x;
        if (j < a_end) {
            const int c = __builtin_nontemporal_load(acol + j);
            me.av = (acc_t)__builtin_nontemporal_load(aval + j);
            struct __attribute__((aligned(4))) I2 {
                int b, e;
            };
            const I2 r = *reinterpret_cast<const I2 *>(brpt + c);
            me.kb = r.b, me.ke = r.e;
        }
        s_ent[threadIdx.x] = me;
        __syncthreads();
        const int nb = a_end - a0 < BS ? a_end - a0 : BS;
        const int per = (nb + NG - 1) / NG;
        const int e0 = gid * per;
        const int e1 = e0 + per < nb ? e0 + per : nb;
        if (e0 < e1 && !(abl & 4)) {
            Buf b0, b1;
            LeanEnt E0 = s_ent[e0], E1;
            issue(E0, b0);
            int e = e0;
            while (true) {
                const bool more1 = e + 1 < e1;
                if (more1) {
                    E1 = s_ent[e + 1];
                    issue(E1, b1);
                }
                consume(E0, b0);
                if (!more1) break;
                const bool more0 = e + 2 < e1;
                if (more0) {
                    E0 = s_ent[e + 2];
                    issue(E0, b0);
                }
                consume(E1, b1);
                if (!more0) break;
                e += 2;
            }
        }
        __syncthreads();
    }
    if ((abl & 1) && abl_sum == (acc_t)-1.2345) dense[0] = abl_sum;
}

// lanes per B row for the lean walk: the smallest power of two that covers the longest B row of the C
// row in CH chunks, at least 8 (a quarter of a 128-byte line of columns), at most 64
__device__ __forceinline__ int lean_group(int maxb, int CH)
{
    int g = 8;
    while (g < 64 && g * CH < maxb) g <<= 1;
    return g;
}

template <int BS, int SPAN_MAX, int MODE, bool LEAN>
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
                                                  const unsigned int *__restrict__ bm, int abl)
{
    // MODE 1: full call -- the column structure of the row comes from the bitmap written by
    //         k_sym_dense; columns and values are emitted in ascending order.
    // MODE 2: numeric-only re-run -- C.col exists; values are gathered at its columns.
    constexpr int NW = BS / 64;
    acc_t *dense = reinterpret_cast<acc_t *>(nsp_dyn_lds);  // dynamic: (widest window of the bin + 4) values
    // the walk's scratch: the generic walk parks (B extent, A value) per thread, the lean one a 16-byte record
    __shared__ __attribute__((aligned(16))) unsigned char s_walk[BS * (LEAN ? sizeof(LeanEnt) : sizeof(int2) + sizeof(real))];
    __shared__ int s_wcnt[NW];
    const int slot = xcd_row_slot(bin_size);
    if (slot < 0) return;
    const int rid = row_perm[bin_off + slot];
    const int off = crpt[rid];
    const int lo = row_lo[rid];
    const int span = row_span[rid];
    // Generic walk: the VW entries a lane holds have consecutive columns inside a run, so one atomic
    // instruction sees columns of stride VW across the lanes: the value of column idx lives at
    // (idx & 3) * Q + (idx >> 2), which turns that stride into consecutive 8-byte slots.
    // Lean walk: consecutive lanes hold consecutive entries; the window is indexed plainly.
    const int Q = (span + 3) >> 2;
    auto slot_of = [&](int idx) { return LEAN ? idx : __mul24(idx & 3, Q) + (idx >> 2); };
    for (int i = threadIdx.x; i < 4 * Q; i += BS) dense[i] = 0;
    __syncthreads();
    const int a_beg = arpt[rid], a_end = arpt[rid + 1];
    if (LEAN) {
        constexpr int CH = 4;
        lean_accumulate<BS, CH>(acol, aval, brpt, bcol, bval, bnnz, a_beg, a_end, lo, dense,
                                reinterpret_cast<LeanEnt *>(s_walk), (abl >> 8) ? (abl >> 8) : lean_group(row_maxb[rid], CH), abl & 255);
    } else {
        int2 *s_ext = reinterpret_cast<int2 *>(s_walk);
        real *s_av = reinterpret_cast<real *>(s_walk + BS * sizeof(int2));
        const int g = group_width(row_prod[rid], a_end - a_beg, BS, row_maxb[rid]);
        walk_products<BS, true>(acol, aval, brpt, bcol, bval, bnnz, a_beg, a_end, g, s_ext, s_av,
                                [&](const IVec &k, const RVec &v, int n, real sc) {
#pragma unroll
                                    for (int i = 0; i < VW; i++)
                                        if (i < n) {
                                            const int idx = k.v[i] - lo;
                                            unsafeAtomicAdd(dense + __mul24(idx & 3, Q) + (idx >> 2), (acc_t)(sc * v.v[i]));
                                        }
                                });
    }
    __syncthreads();
    if (abl & 8) return;  // diagnostics: no emission
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
