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
template <int BS, int TMAX, bool LARGE, int COOP = 0>
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
                                           if (COOP) ht_insert_vec_coop(tab, mask, k, n, h, cnt, COOP);
                                           else ht_insert_vec(tab, mask, k, n, h, cnt);
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

// One-wavefront symbolic bin with PERSISTENT wavefronts and the next rows' dependent loads in flight behind the
// current row (round 3).  k_sym_tb<64, ...> is five dependent round trips per row (row number -> row words ->
// A entries -> B extents -> B entries) around a few hundred instructions of hashing, with 32 rows in flight per CU
// (one per wavefront: counters say 60-70 % of the wave cycles wait on memory).  Here a wavefront strides over the
// rows of its XCD's eighth of the bin with a three-stage pipeline in registers: row i + 2's four row words (ONE
// vector load, lanes 0..3 one word each: a scalar load would be waited for at the first LDS fence), row i + 1's A
// entry of the first batch and then its B extent, row i walked from registers.  Rows with one B row far longer than
// the others ("mixed") take the flat walk as in k_sym_tb; their first batch is not prefetched.
// MEASURED SLOWER than k_sym_tb<64, 1024> (stencil symbolic 0.75 -> 1.05 ms; so was the same pipeline in the numeric
// bin, numeric.h: k_num_wave): what helped these bins was more rows in flight per CU, not a shorter chain per
// wavefront.  Opt-in (NSPARSE_SYM_WAVE=n), parity test test_persistent_wavefront_bin.
template <int TMAX>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8))) void k_sym_wave(const int *__restrict__ arpt, const int *__restrict__ acol,
                                                 const int *__restrict__ brpt, const int *__restrict__ bcol,
                                                 const int *__restrict__ row_perm, const int *__restrict__ row_prod,
                                                 const int *__restrict__ row_maxb, int *__restrict__ row_nz, int bin_off,
                                                 int bin_size, int bnnz, int flat_on)
{
    constexpr int BS = 64;
    __shared__ __attribute__((aligned(16))) int tab[TMAX];
    __shared__ int2 s_ext[BS];
    __shared__ DeferList<false, (TMAX == 1024 ? 24 : 32)> s_defer;
    __shared__ FlatScratch<BS> s_flat;
    const int lane = threadIdx.x;
    const int nb8 = (bin_size + 7) >> 3;
    const int xcd = (int)(blockIdx.x & 7), stride = (int)(gridDim.x >> 3), j0 = (int)(blockIdx.x >> 3);
    auto rid_batch = [&](int i0) {  // lane l: the (i0 + l)-th row of this wavefront, -1 beyond the eighth / the bin
        const unsigned j = (unsigned)j0 + (unsigned)(i0 + lane) * (unsigned)stride;
        const unsigned slot = (unsigned)xcd * (unsigned)nb8 + j;
        return (j < (unsigned)nb8 && slot < (unsigned)bin_size) ? row_perm[(unsigned)bin_off + slot] : -1;
    };
    // lanes 0..3 fetch row_prod[rid], arpt[rid], arpt[rid + 1], row_maxb[rid]
    const int *lane_base = row_prod;
    lane_base = lane == 1 ? arpt : lane_base;
    lane_base = lane == 2 ? arpt + 1 : lane_base;
    lane_base = lane == 3 ? row_maxb : lane_base;
    auto row_words = [&](int rid) { return (rid >= 0 && lane < 4) ? lane_base[rid] : 0; };
    // lane groups of the walk for the row whose words are w; bit 8: mixed (walk_products_mixed's rule)
    auto width_of = [&](int w) {
        const int np = __builtin_amdgcn_readlane(w, 0);
        const int alen = __builtin_amdgcn_readlane(w, 2) - __builtin_amdgcn_readlane(w, 1);
        const int mb = __builtin_amdgcn_readlane(w, 3);
        const bool mixed = alen > 1 && (long long)mb * alen > 8LL * np;
        return group_width(np, alen, BS, mb) | (mixed ? 256 : 0);
    };
    // the lane's A entry of the first batch of walk_products (b0 = 0: m = gl), -1: none (or a mixed row)
    auto first_entry = [&](int rid, int w, int gm) {
        const int g = gm & 255;
        const int a_beg = __builtin_amdgcn_readlane(w, 1), a_end = __builtin_amdgcn_readlane(w, 2);
        const int lg = 31 - __clz(g), lng = 6 - lg, ng = 1 << lng, gid = lane >> lg, gl = lane & (g - 1);
        const int first = a_beg + gid;
        const int cnt = first < a_end ? (a_end - first + ng - 1) >> lng : 0;
        return (rid >= 0 && !(gm & 256) && gl < cnt) ? first + gl * ng : -1;
    };
    struct __attribute__((aligned(4))) I2 {
        int b, e;
    };
    int rids = rid_batch(0);
    if (__builtin_amdgcn_readlane(rids, 0) < 0) return;
    int rid_cur = __builtin_amdgcn_readlane(rids, 0);
    int w_cur = row_words(rid_cur);
    int gm_cur = width_of(w_cur);
    int2 pre_e = make_int2(0, 0);
    {
        const int j = first_entry(rid_cur, w_cur, gm_cur);
        if (j >= 0) {
            const int c = __builtin_nontemporal_load(acol + j);
            const I2 r = *reinterpret_cast<const I2 *>(brpt + c);
            pre_e = make_int2(r.b, r.e);
        }
    }
    int rid_nxt = __builtin_amdgcn_readlane(rids, 1);
    int w_nxt = row_words(rid_nxt);
    for (int i = 0;; i++) {
        // ---- requests for the rows behind this one ----------------------------------------------
        if (((i + 2) & 63) == 0) rids = rid_batch(i + 2);  // (one trip per 64 rows)
        const int rid2 = rid_nxt >= 0 ? __builtin_amdgcn_readlane(rids, (i + 2) & 63) : -1;
        const int w2 = row_words(rid2);
        const int gm_nxt = width_of(w_nxt);
        const int jn = first_entry(rid_nxt, w_nxt, gm_nxt);
        int cn = 0;
        if (jn >= 0) cn = __builtin_nontemporal_load(acol + jn);
        // ---- this row -------------------------------------------------------------------------------
        const int np = __builtin_amdgcn_readlane(w_cur, 0);
        const int a_beg = __builtin_amdgcn_readlane(w_cur, 1), a_end = __builtin_amdgcn_readlane(w_cur, 2);
        int T = pow2_ceil(np + (np >> 1));
        T = T < 64 ? 64 : (T > TMAX ? TMAX : T);
        const int mask = T - 1;
        {
            int4 *t4 = reinterpret_cast<int4 *>(tab);
            const int4 m1 = make_int4(-1, -1, -1, -1);
            for (int q = lane; q < T / 4; q += BS) t4[q] = m1;
        }
        if (lane == 0) s_defer.n = 0;
        wave_lds_sync();
        int cnt = 0;
        auto consume = [&](const IVec &k, const RVecT<1> &, int n, real) {
            int h[VW];
            ht_insert_vec(tab, mask, k, n, h, cnt);
        };
        if (gm_cur & 256) {
            walk_products_mixed<BS, false>(acol, (const real *)nullptr, brpt, bcol, (const real *)nullptr, bnnz, a_beg, a_end,
                                           np, __builtin_amdgcn_readlane(w_cur, 3), s_ext, (real *)nullptr, &s_defer, consume,
                                           flat_on ? &s_flat : (FlatScratch<BS> *)nullptr, flat_on == 2);
        } else {
            walk_products<BS, false, VW, decltype(consume) &, (TMAX == 1024 ? 24 : 32)>(
                acol, (const real *)nullptr, brpt, bcol, (const real *)nullptr, bnnz, a_beg, a_end, gm_cur & 255, s_ext,
                (real *)nullptr, consume, (DeferList<false, (TMAX == 1024 ? 24 : 32)> *)nullptr, 0x7fffffff, nullptr, &pre_e,
                (const real *)nullptr);
        }
        // ---- next row: B extent of the lane's entry (its column has had the whole walk to arrive) ----
        pre_e = make_int2(0, 0);
        if (jn >= 0) {
            const I2 r = *reinterpret_cast<const I2 *>(brpt + cn);
            pre_e = make_int2(r.b, r.e);
        }
        cnt = wave_sum(cnt);
        if (lane == 0) row_nz[rid_cur] = cnt;
        if (rid_nxt < 0) return;
        wave_lds_sync();  // the next row clears the table this one has just filled
        rid_cur = rid_nxt;
        w_cur = w_nxt;
        gm_cur = gm_nxt;
        w_nxt = w2;
        rid_nxt = rid2;
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
