// spgemm/symbolic.h -- symbolic hash kernels (bins 0-5).
// Part of the spgemm_hash.hip translation unit (kernels are launched from its host code).
#pragma once
#include "common.h"

namespace nsp {
namespace spgemm {

// ===================================================================================
//  symbolic phase
// ===================================================================================

// bin 0: LPR lanes per row, TROW keys per row (set_row_nz_bin_pwarp :266-327).
template <int BS, int LPR, int TROW>
__global__ __launch_bounds__(BS) void k_sym_small(const int *__restrict__ arpt,
                                                  const int *__restrict__ acol,
                                                  const int *__restrict__ brpt,
                                                  const int *__restrict__ bcol,
                                                  const int *__restrict__ row_perm,
                                                  int *__restrict__ row_nz, int bin_off, int bin_size)
{
    constexpr int RPB = BS / LPR;
    __shared__ int tab[RPB * TROW];
    {  // every wavefront clears only the slots of its own rows: no workgroup barrier
        constexpr int WSLOTS = 64 / LPR * TROW;
        const int w0 = (threadIdx.x >> 6) * WSLOTS;
        for (int i = threadIdx.x & 63; i < WSLOTS; i += 64) tab[w0 + i] = -1;
    }
    wave_lds_sync();
    const int lrow = threadIdx.x / LPR;
    const int sub = threadIdx.x % LPR;
    const int q = blockIdx.x * RPB + lrow;
    int cnt = 0;
    int rid = 0;
    if (q < bin_size) {
        rid = row_perm[bin_off + q];
        int *t = tab + lrow * TROW;
        const int e = arpt[rid + 1];
        // A lane takes EB of its A entries at a time and asks for everything they need level by
        // level -- the EB columns, then the EB row extents, then the first two entries of each
        // B row -- so a row costs ~3 memory round trips instead of 3 per entry (these rows are a
        // dependent-load chain with nothing else to hide it: power-law inputs put a million of
        // them into this bin).
        constexpr int EB = 4;
        for (int j0 = arpt[rid] + sub; j0 < e; j0 += LPR * EB) {
            int c[EB], kb[EB], ke[EB], k0[EB], k1[EB];
#pragma unroll
            for (int u = 0; u < EB; u++) {
                const int j = j0 + u * LPR;
                c[u] = j < e ? __builtin_nontemporal_load(acol + j) : -1;
            }
#pragma unroll
            for (int u = 0; u < EB; u++) {
                kb[u] = c[u] >= 0 ? brpt[c[u]] : 0;
                ke[u] = c[u] >= 0 ? brpt[c[u] + 1] : 0;
            }
#pragma unroll
            for (int u = 0; u < EB; u++) {
                k0[u] = kb[u] < ke[u] ? bcol[kb[u]] : -1;
                k1[u] = kb[u] + 1 < ke[u] ? bcol[kb[u] + 1] : -1;
            }
#pragma unroll
            for (int u = 0; u < EB; u++) {
                int fresh;
                if (k0[u] >= 0) {
                    ht_find_or_insert(t, TROW - 1, k0[u], &fresh);
                    cnt += fresh;
                }
                if (k1[u] >= 0) {
                    ht_find_or_insert(t, TROW - 1, k1[u], &fresh);
                    cnt += fresh;
                }
                for (int k = kb[u] + 2; k < ke[u]; k++) {
                    ht_find_or_insert(t, TROW - 1, bcol[k], &fresh);
                    cnt += fresh;
                }
            }
        }
    }
#pragma unroll
    for (int o = LPR / 2; o >= 1; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (q < bin_size && sub == 0) row_nz[rid] = cnt;
}

// bins 1..5: one workgroup per row (set_row_nz_bin_each_tb :399-472; LARGE = the try-in-LDS
// kernel with a fail list, set_row_nz_bin_each_tb_large :474-554).
#ifndef NSP_FLAT_SYM_MIN_T
#define NSP_FLAT_SYM_MIN_T 512
#endif
template <int BS, int TMAX, bool LARGE>
__global__ __launch_bounds__(BS) void k_sym_tb(const int *__restrict__ arpt,
                                               const int *__restrict__ acol,
                                               const int *__restrict__ brpt,
                                               const int *__restrict__ bcol,
                                               const int *__restrict__ row_perm,
                                               const int *__restrict__ row_prod,
                                                  const int *__restrict__ row_maxb,
                                               int *__restrict__ row_nz, int bin_off, int bin_size,
                                               int bnnz, BinState *bs, int *__restrict__ fail_list, int flat_on = 1,
                                               int *__restrict__ tcol = nullptr, long long *__restrict__ list_off = nullptr,
                                               const int *__restrict__ row_span = nullptr, int dens = 0, int tiled_w = 0)
{
    // tcol != nullptr (big-table bins only): a row that will be heavy in the numeric phase -- more non-zeros than
    // the LDS hash bins take, not dense enough for the dense tiles -- also leaves its columns as a sorted list for
    // the list-driven tiles of the ranked kernel (heavy_ranked.h): compacted in place, sorted by the bitonic
    // network of the numeric bins, one returning atomic for its place in the slab.
    __shared__ __attribute__((aligned(16))) int tab[TMAX];
    __shared__ int2 s_ext[LARGE ? 1 : BS];
    // (the 1024-slot, one-wavefront instance: 24 parked rows keep a row's LDS below 5120 B = 32 rows per CU)
    __shared__ DeferList<false, (LARGE ? 32 : (TMAX / 32 > 32 ? TMAX / 32 : (TMAX == 1024 && BS == 64 ? 24 : 32)))> s_defer;
    __shared__ FlatScratch<LARGE ? 64 : BS> s_flat;
    __shared__ int s_nz;
    const int slot = xcd_row_slot(bin_size);
    if (slot < 0) return;
    const int rid = row_perm[bin_off + slot];
    // every row word in ONE round trip (the loads that stood behind the barrier below started a trip later)
    const int np = row_prod[rid];
    const int a_beg = arpt[rid], a_end = arpt[rid + 1];
    const int mb_row = row_maxb[rid];
    int T = LARGE ? TMAX : pow2_ceil(np + (np >> 1));  // load factor <= 2/3 where the bin's table allows
    if (T < 64) T = 64;
    if (T > TMAX) T = TMAX;
    const int mask = T - 1;
    {
        int4 *t4 = reinterpret_cast<int4 *>(tab);
        const int4 m1 = make_int4(-1, -1, -1, -1);
        for (int i = threadIdx.x; i < T / 4; i += BS) t4[i] = m1;
    }
    if (threadIdx.x == 0) {
        s_nz = 0;
        s_defer.n = 0;
    }
    __syncthreads();

    const int g = group_width(np, a_end - a_beg, BS, mb_row);
    int cnt = 0;
    if (!LARGE) {
        walk_products_mixed<BS, false>(acol, (const real *)nullptr, brpt, bcol, (const real *)nullptr, bnnz,
                                       a_beg, a_end, np, mb_row, s_ext, (real *)nullptr, &s_defer,
                                       [&](const IVec &k, const RVecT<1> &, int n, real) {
                                           int h[VW];
                                           ht_insert_vec(tab, mask, k, n, h, cnt);
                                       }, (TMAX >= NSP_FLAT_SYM_MIN_T && flat_on) ? reinterpret_cast<FlatScratch<BS> *>(&s_flat) : (FlatScratch<BS> *)nullptr,
                                       flat_on == 2);
    } else {
        // try-in-LDS: plain walk with early exit once the table holds kSymLargeLimit keys
        const int lg = 31 - __clz(g);
        const int ngroups = BS >> lg;
        const int gid = (int)threadIdx.x >> lg, gl = (int)threadIdx.x & (g - 1);
        bool full = false;
        for (int j = a_beg + gid; j < a_end && !full; j += ngroups) {
            const int c = __builtin_nontemporal_load(acol + j);
            const int ke = brpt[c + 1];
            for (int k = brpt[c] + gl; k < ke; k += g) {
                if (lds_load(&s_nz) >= kSymLargeLimit) { full = true; break; }
                int fresh;
                ht_find_or_insert(tab, mask, bcol[k], &fresh);
                if (fresh) atomicAdd(&s_nz, 1);
            }
        }
    }
    if (!LARGE) {
        cnt = wave_sum(cnt);
        if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&s_nz, cnt);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nz = s_nz;
        if (LARGE && nz >= kSymLargeLimit) {
            fail_list[atomicAdd(&bs->fail_count, 1)] = rid;
        } else {
            row_nz[rid] = nz;
        }
    }
    if constexpr (!LARGE && TMAX >= 8192) {
        const int nz = s_nz;
        bool want = tcol != nullptr && nz > kListMinNnz;
        if (want) {
            const int sp = row_span[rid];
            want = !(dens > 0 && (long long)nz * dens >= sp && sp <= 32 * tiled_w);  // k_num_tiled's rows need no list
        }
        if (want) {
            constexpr int SPT = TMAX / BS, NWV = BS / 64;
            __shared__ int s_ws[NWV];
            __shared__ long long s_off;
            int keys[SPT], mine = 0;
#pragma unroll
            for (int j = 0; j < SPT; j++) {
                const int i = (int)threadIdx.x * SPT + j;
                keys[j] = i < T ? tab[i] : -1;
                mine += keys[j] != -1;
            }
            const int incl = wave_incl_scan(mine);
            if ((threadIdx.x & 63) == 63) s_ws[threadIdx.x >> 6] = incl;
            if (threadIdx.x == 0) {
                s_off = (long long)atomicAdd(&bs->list_cursor, (unsigned long long)nz);
                list_off[rid] = s_off;
            }
            __syncthreads();  // every slot has been read: the table may be overwritten
            int at = incl - mine;
#pragma unroll
            for (int u = 0; u < NWV; u++) at += u < (int)(threadIdx.x >> 6) ? s_ws[u] : 0;
#pragma unroll
            for (int j = 0; j < SPT; j++)
                if (keys[j] != -1) tab[at++] = keys[j];
            const int P = pow2_ceil(nz);  // <= T: the table was sized for the products
            for (int i = nz + threadIdx.x; i < P; i += BS) tab[i] = 0x7fffffff;
            __syncthreads();
            bitonic_sort_lds<BS>(tab, P);
            int *dst = tcol + s_off;
            for (int i = threadIdx.x; i < nz; i += BS) dst[i] = tab[i];
        }
    }
}

// overflow rows: persistent workgroups, private slice of a global slab
// (set_row_nz_bin_each_gl :556-622, bounded-memory variant HashSpGEMM_volta.hpp:341-412).
template <int BS>
__global__ __launch_bounds__(BS) void k_sym_global(const int *__restrict__ arpt,
                                                   const int *__restrict__ acol,
                                                   const int *__restrict__ brpt,
                                                   const int *__restrict__ bcol,
                                                   const int *__restrict__ fail_list, int count,
                                                   const int *__restrict__ row_prod,
                                                  const int *__restrict__ row_maxb,
                                                   int *__restrict__ row_nz, int ncols,
                                                   BinState *bs, int *__restrict__ slab,
                                                   long long slice)
{
    __shared__ int s_row;
    __shared__ int s_nz;
    int *tab = slab + (long long)blockIdx.x * slice;
    while (true) {
        __syncthreads();
        if (threadIdx.x == 0) {
            s_row = atomicAdd(&bs->queue_head, 1);
            s_nz = 0;
        }
        __syncthreads();
        const int q = s_row;
        if (q >= count) break;
        const int rid = fail_list[q];
        long long bound = row_prod[rid];
        if (bound > ncols) bound = ncols;  // a row of C has at most ncols distinct columns
        long long T = 64;
        while (T < 2 * bound) T <<= 1;
        if (T > slice) T = slice;
        const long long mask = T - 1;
        for (long long i = threadIdx.x; i < T; i += BS) tab[i] = -1;
        __syncthreads();
        const int a_beg = arpt[rid], a_end = arpt[rid + 1];
        int cnt = 0;
        for (int j = a_beg + (threadIdx.x >> 6); j < a_end; j += BS / 64) {
            const int c = acol[j];
            const int ke = brpt[c + 1];
            for (int k = brpt[c] + (threadIdx.x & 63); k < ke; k += 64) {
                int fresh;
                gt_find_or_insert(tab, mask, bcol[k], &fresh);
                cnt += fresh;
            }
        }
        cnt = wave_sum(cnt);
        if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&s_nz, cnt);
        __syncthreads();
        if (threadIdx.x == 0) row_nz[rid] = s_nz;
    }
}

}  // namespace spgemm
}  // namespace nsp
