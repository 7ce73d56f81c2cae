// spgemm/numeric.h -- numeric hash kernels (bins 0-4) and the global-table fallback.
// Part of the spgemm_hash.hip translation unit (kernels are launched from its host code).
#pragma once
#include "common.h"

namespace nsp {
namespace spgemm {

// ===================================================================================
//  numeric phase
// ===================================================================================

// bin 0: LPR lanes per row, TROW slots per row, rank sort (calculate_value_col_bin_pwarp
// :631-723).  The LPR lanes of a row live in one wavefront, so wave-level ordering of LDS
// operations is all the synchronisation needed between accumulate and read-out.
template <int BS, int LPR, int TROW>
__global__ __launch_bounds__(BS) void k_num_small(const int *__restrict__ arpt,
                                                  const int *__restrict__ acol,
                                                  const real *__restrict__ aval,
                                                  const int *__restrict__ brpt,
                                                  const int *__restrict__ bcol,
                                                  const real *__restrict__ bval,
                                                  const int *__restrict__ crpt,
                                                  int *__restrict__ ccol, real *__restrict__ cval,
                                                  const int *__restrict__ row_perm, int bin_off,
                                                  int bin_size, int write_col)
{
    constexpr int RPB = BS / LPR;
    __shared__ int keys[RPB * TROW];
    // values in acc_t (double in both builds) like every other accumulator: a row of this bin has at most
    // 16 non-zeros but may have thousands of products (a C with one column), and float sums of those
    // miss the reference's 1e-6 (fuzz seed 9047, float build: 8 of 63 entries); besides, ds_add_f32 is
    // the slow LDS atomic on gfx950
    __shared__ acc_t vals[RPB * TROW];
    // every wavefront clears and uses only the slots of its own 64 / LPR rows: no workgroup barrier,
    // so a wavefront does not wait for the slowest row of the other three
    {
        constexpr int WSLOTS = 64 / LPR * TROW;
        const int w0 = (threadIdx.x >> 6) * WSLOTS;
        for (int i = threadIdx.x & 63; i < WSLOTS; i += 64) {
            keys[w0 + i] = -1;
            vals[w0 + i] = 0;
        }
    }
    wave_lds_sync();
    const int lrow = threadIdx.x / LPR;
    const int sub = threadIdx.x % LPR;
    const int q = blockIdx.x * RPB + lrow;
    const bool active = q < bin_size;
    int rid = 0;
    int *kt = keys + lrow * TROW;
    acc_t *vt = vals + lrow * TROW;
    if (active) {
        rid = row_perm[bin_off + q];
        const int e = arpt[rid + 1];
        auto add = [&](int key, real x) {
            int fresh;
            const int h = ht_find_or_insert(kt, TROW - 1, key, &fresh);
            unsafeAtomicAdd(vt + h, (acc_t)x);
        };
        // EB of the lane's A entries at a time, their loads requested level by level (see
        // k_sym_small): ~3 memory round trips per row instead of 3 per entry
        constexpr int EB = 4;
        for (int j0 = arpt[rid] + sub; j0 < e; j0 += LPR * EB) {
            int c[EB], kb[EB], ke[EB], k0[EB], k1[EB];
            real av[EB], v0[EB], v1[EB];
#pragma unroll
            for (int u = 0; u < EB; u++) {
                const int j = j0 + u * LPR;
                c[u] = j < e ? __builtin_nontemporal_load(acol + j) : -1;
                av[u] = j < e ? __builtin_nontemporal_load(aval + j) : (real)0;
            }
#pragma unroll
            for (int u = 0; u < EB; u++) {
                kb[u] = c[u] >= 0 ? brpt[c[u]] : 0;
                ke[u] = c[u] >= 0 ? brpt[c[u] + 1] : 0;
            }
#pragma unroll
            for (int u = 0; u < EB; u++) {
                const bool h0 = kb[u] < ke[u], h1 = kb[u] + 1 < ke[u];
                k0[u] = h0 ? bcol[kb[u]] : -1;
                v0[u] = h0 ? bval[kb[u]] : (real)0;
                k1[u] = h1 ? bcol[kb[u] + 1] : -1;
                v1[u] = h1 ? bval[kb[u] + 1] : (real)0;
            }
#pragma unroll
            for (int u = 0; u < EB; u++) {
                if (k0[u] >= 0) add(k0[u], av[u] * v0[u]);
                if (k1[u] >= 0) add(k1[u], av[u] * v1[u]);
                for (int k = kb[u] + 2; k < ke[u]; k++) add(bcol[k], av[u] * bval[k]);
            }
        }
    }
    wave_lds_sync();
    if (active) {
        const int off = crpt[rid];
        for (int s = sub; s < TROW; s += LPR) {
            const int key = kt[s];
            if (key == -1) continue;
            int rank = 0;
            for (int u = 0; u < TROW; u++) {
                const int o = kt[u];
                rank += (o != -1 && o < key) ? 1 : 0;
            }
            if (write_col & 1) ccol[off + rank] = key;
            cval[off + rank] = (real)vt[s];
        }
    }
}

// (lane_xor, bitonic_stages_reg*, bitonic_sort_lds: common.h -- the symbolic hash kernels sort their lists with them too)

// bins 1..4: one workgroup per row (calculate_value_col_bin_each_tb :829-927).
template <int BS, int TMAX, int PMAX, int COOP = 0>
__global__ __launch_bounds__(BS) void k_num_tb(const int *__restrict__ arpt,
                                               const int *__restrict__ acol,
                                               const real *__restrict__ aval,
                                               const int *__restrict__ brpt,
                                               const int *__restrict__ bcol,
                                               const real *__restrict__ bval,
                                               const int *__restrict__ crpt,
                                               int *__restrict__ ccol, real *__restrict__ cval,
                                               const int *__restrict__ row_perm,
                                               const int *__restrict__ row_prod,
                                                  const int *__restrict__ row_maxb, int bin_off,
                                               int bin_size, int bnnz, int write_col,
                                               unsigned long long *prof = nullptr)
{
    // prof (a build with -DNSPARSE_TB_PROF_BUILD and NSPARSE_TB_PROF=1; compiled out otherwise: the pointer and
    // the clock cost three scalar registers, and with them the 256-thread bin loses a wavefront per SIMD),
    // 100 MHz ticks of thread 0: 0 row record + clear, 1 walk, 2 compaction, 3 sort, 4 read-out; 5 rows
#ifdef NSPARSE_TB_PROF_BUILD
    unsigned long long tk = prof ? wall_clock64() : 0;
    auto tick = [&](int phase) {
        if (prof && threadIdx.x == 0) {
            const unsigned long long now = wall_clock64();
            atomicAdd(prof + phase, now - tk);
            tk = now;
        }
    };
#else
    auto tick = [](int) {};
#endif
    __shared__ __attribute__((aligned(16))) acc_t vals[TMAX];
    __shared__ __attribute__((aligned(16))) int keys[TMAX];
    // the scratch of the product walk and the sort buffer are never alive together: one block of LDS for
    // both (a row of the 256-slot bin: 5.6 -> 4.6 KB, so the 32 wavefronts of a CU all get a row instead of 29)
    struct WalkScratch {
        int2 ext[BS];
        real av[BS];
        DeferList<true, (PMAX / 16 > 32 ? PMAX / 16 : 32)> defer;
        FlatScratch<(BS >= 256 ? BS : 64)> flat;
    };
    // write_col bit 2: NSPARSE_FLAT=0.  The flat walk keeps U chunks in flight per lane (26 more registers): only where the LDS of a row bounds
    // the occupancy anyway, not in the one-wavefront-per-row bins that live on rows in flight
#ifndef NSP_FLAT_NUM_MIN_BS
#define NSP_FLAT_NUM_MIN_BS 512
#endif
    constexpr bool FLAT = BS >= NSP_FLAT_NUM_MIN_BS;
    union Overlay {
        WalkScratch w;
        int srt[PMAX];
    };
    __shared__ __attribute__((aligned(16))) Overlay s_ov;
    int *srt = s_ov.srt;
    int2 *s_ext = s_ov.w.ext;
    real *s_av = s_ov.w.av;
    auto &s_defer = s_ov.w.defer;
    __shared__ int s_cnt;
    constexpr int NBK = 512;
    __shared__ int s_bk[FLAT ? 2 * NBK + 8 : 1];  // bucket sort of the big-table bins (common.h: table_to_sorted)
    // Big-table bins (BS >= 512): the launch may hold FEWER workgroups than rows (a multiple of 8: write_col bit 4
    // says so) and every workgroup strides over the rows of its XCD's eighth of the bin.  A workgroup per row is a
    // dispatch, a drain of the row's stores and a fresh LDS allocation per row, with one or two rows per CU and
    // nothing to hide them behind: R-MAT-22, 8192-slot bin, 41 us per row of which 27 are the kernel's own phases.
    const int nb8 = (bin_size + 7) >> 3;
    const bool persist = FLAT && (write_col & 16);
    for (int j8 = (int)(blockIdx.x >> 3);; j8 += (int)(gridDim.x >> 3)) {
    const int slot = persist ? (j8 < nb8 ? (int)(blockIdx.x & 7) * nb8 + j8 : bin_size) : xcd_row_slot(bin_size);
    if (slot < 0 || slot >= bin_size) return;
    const int rid = row_perm[bin_off + slot];
    // every row word in ONE round trip (the loads that stood behind the barrier below started a trip later)
    const int off = crpt[rid];
    const int n = crpt[rid + 1] - off;
    const int a_beg = arpt[rid], a_end = arpt[rid + 1];
    const int np_row = row_prod[rid], mb_row = row_maxb[rid];
    int T = pow2_ceil(n + (n >> 1));
    if (T < 64) T = 64;
    if (T > TMAX) T = TMAX;
    const int mask = T - 1;
    for (int i = threadIdx.x; i < T; i += BS) {
        keys[i] = -1;
        vals[i] = 0;
    }
    if (threadIdx.x == 0) {
        s_cnt = 0;
        s_defer.n = 0;
    }
    __syncthreads();
    tick(0);

    walk_products_mixed<BS, true>(acol, aval, brpt, bcol, bval, bnnz, a_beg, a_end, np_row, mb_row,
                                  s_ext, s_av, &s_defer,
                                  [&](const IVec &k, const RVec &v, int n, real sc) {
                                      int h[VW], fresh = 0;
                                      if (COOP) ht_insert_vec_coop(keys, mask, k, n, h, fresh, COOP);
                                      else ht_insert_vec(keys, mask, k, n, h, fresh);
#pragma unroll
                                      for (int i = 0; i < VW; i++)
                                          if (i < n) unsafeAtomicAdd(vals + h[i], (acc_t)(sc * v.v[i]));
                                  }, (FLAT && !(write_col & 4)) ? reinterpret_cast<FlatScratch<BS> *>(&s_ov.w.flat) : (FlatScratch<BS> *)nullptr,
                                  (write_col & 8) != 0);
    __syncthreads();
    tick(1);

    const int lane = threadIdx.x & 63;
    const int P = pow2_ceil(n);
    bool sorted_already = false;
    if constexpr (FLAT) {
        // big-table bins: the table's keys go straight into srt, sorted bucket by bucket (write_col bit 6: off)
        if (!(write_col & 2) && !(write_col & 64)) {
            sorted_already = table_to_sorted<BS, NBK, TMAX / BS>(keys, T, srt, s_bk);
            if (!sorted_already) {
                for (int i = n + threadIdx.x; i < P; i += BS) srt[i] = 0x7fffffff;
                __syncthreads();
                bitonic_sort_lds<BS>(srt, P);
                sorted_already = true;
            }
            tick(2);
        }
    }
    if (!sorted_already) {
    // compaction: ballot + popcount inside the wave, one LDS atomic per 64 slots
    for (int base = (threadIdx.x >> 6) * 64; base < T; base += BS) {
        const int key = keys[base + lane];
        const bool occ = key != -1;
        const unsigned long long m = __ballot(occ);
        if (m) {
            int start = 0;
            if (lane == 0) start = atomicAdd(&s_cnt, __popcll(m));
            start = __shfl(start, 0);
            if (occ) srt[start + __popcll(m & ((1ull << lane) - 1ull))] = key;
        }
    }
    for (int i = n + threadIdx.x; i < P; i += BS) srt[i] = 0x7fffffff;
    __syncthreads();
    tick(2);
    // write_col bit 1: unsorted output requested (cuda-cpp template<bool sort>,
    // HashSpGEMM_volta.hpp:585-604): columns leave in compaction order
    if (P > 1 && !(write_col & 2)) bitonic_sort_lds<BS>(srt, P);
    }
    tick(3);

    for (int i = threadIdx.x; i < n; i += BS) {
        const int key = srt[i];
        int h = hash_slot(key, mask);
        while (keys[h] != key) h = (h + 1) & mask;
        if (write_col & 1) ccol[off + i] = key;
        cval[off + i] = (real)vals[h];
    }
    tick(4);
#ifdef NSPARSE_TB_PROF_BUILD
    if (prof && threadIdx.x == 0) atomicAdd(prof + 5, 1ull);
#endif
    if (!persist) return;
    __syncthreads();  // the next row clears the tables this one has just read
    }
}

// bin 1 (17..170 non-zeros, 256-slot table), one wavefront per row, PERSISTENT wavefronts with the next rows' dependent
// loads in flight behind the current row.  k_num_tb<64, ...> is a chain of five dependent round trips per row (row
// number -> row words -> A entries -> B extents -> B entries) with one row per wavefront and 32 wavefronts per CU (the
// LDS of a row): on the 27-point stencil 15.8 us per row of which the hashing and the sort are a small part.  Here a
// wavefront strides over the rows of its XCD's eighth of the bin and keeps a three-stage pipeline in registers:
//   row i + 2: its six row words, ONE vector load (lanes 0..5 fetch one word each; a scalar load would share lgkmcnt
//              with the LDS traffic of the walk and be waited for at the first LDS fence)
//   row i + 1: the lane's A entry of the first batch (requested at the top of row i), then its B extent (requested
//              after the walk of row i)
//   row i    : walked from registers: only the B entries themselves are a round trip
// The row numbers come 64 at a time (lane l holds the l-th row of the wavefront's stride).
template <int TMAX, int PMAX>
__global__ __launch_bounds__(64) void k_num_wave(const int *__restrict__ arpt, const int *__restrict__ acol,
                                                 const real *__restrict__ aval, const int *__restrict__ brpt,
                                                 const int *__restrict__ bcol, const real *__restrict__ bval,
                                                 const int *__restrict__ crpt, int *__restrict__ ccol,
                                                 real *__restrict__ cval, const int *__restrict__ row_perm,
                                                 const int *__restrict__ row_prod, const int *__restrict__ row_maxb,
                                                 int bin_off, int bin_size, int bnnz, int write_col)
{
    constexpr int BS = 64;
    constexpr int DCAP = (PMAX / 16 > 32 ? PMAX / 16 : 32);
    __shared__ __attribute__((aligned(16))) acc_t vals[TMAX];
    __shared__ __attribute__((aligned(16))) int keys[TMAX];
    struct WalkScratch {
        int2 ext[BS];
        real av[BS];
        DeferList<true, DCAP> defer;
    };
    union Overlay {
        WalkScratch w;
        int srt[PMAX];
    };
    __shared__ __attribute__((aligned(16))) Overlay s_ov;
    int *srt = s_ov.srt;
    const int lane = threadIdx.x;
    const int nb8 = (bin_size + 7) >> 3;
    const int xcd = (int)(blockIdx.x & 7), stride = (int)(gridDim.x >> 3), j0 = (int)(blockIdx.x >> 3);
    // rows of this wavefront: slots xcd * nb8 + j0 + i * stride (i = 0, 1, ...) while inside the eighth and the bin
    auto rid_batch = [&](int i0) {
        const unsigned j = (unsigned)j0 + (unsigned)(i0 + lane) * (unsigned)stride;  // < nb8 + 64 * stride: no overflow
        const unsigned slot = (unsigned)xcd * (unsigned)nb8 + j;
        return (j < (unsigned)nb8 && slot < (unsigned)bin_size) ? row_perm[(unsigned)bin_off + slot] : -1;
    };
    // lanes 0..5 fetch crpt[rid], crpt[rid+1], arpt[rid], arpt[rid+1], row_prod[rid], row_maxb[rid]: the lane's array
    // is picked once (six lane masks alive in the row loop were six scalar register pairs)
    const int *lane_base = crpt;
    lane_base = lane == 1 ? crpt + 1 : lane_base;
    lane_base = lane == 2 ? arpt : lane_base;
    lane_base = lane == 3 ? arpt + 1 : lane_base;
    lane_base = lane == 4 ? row_prod : lane_base;
    lane_base = lane == 5 ? row_maxb : lane_base;
    auto row_words = [&](int rid) { return (rid >= 0 && lane < 6) ? lane_base[rid] : 0; };
    // width of the walk's lane groups for the row whose words are w (bit 8: "mixed", walk_products_mixed's rule)
    auto width_of = [&](int w) {
        const int alen = __builtin_amdgcn_readlane(w, 3) - __builtin_amdgcn_readlane(w, 2);
        const int np = __builtin_amdgcn_readlane(w, 4), mb = __builtin_amdgcn_readlane(w, 5);
        const bool mixed = alen > 1 && (long long)mb * alen > 8LL * np;
        return group_width(mixed ? np - mb : np, mixed ? alen - 1 : alen, BS, mixed ? 0 : mb, VW) | (mixed ? 256 : 0);
    };
    // the lane's A entry of the first batch of walk_products (b0 = 0: m = gl), -1: none
    auto first_entry = [&](int rid, int w, int gm) {
        const int g = gm & 255;
        const int a_beg = __builtin_amdgcn_readlane(w, 2), a_end = __builtin_amdgcn_readlane(w, 3);
        const int lg = 31 - __clz(g), lng = 6 - lg, ng = 1 << lng, gid = lane >> lg, gl = lane & (g - 1);
        const int first = a_beg + gid;
        const int cnt = first < a_end ? (a_end - first + ng - 1) >> lng : 0;
        return (rid >= 0 && gl < cnt) ? first + gl * ng : -1;
    };
    struct __attribute__((aligned(4))) I2 {
        int b, e;
    };

    int rids = rid_batch(0);
    if (__builtin_amdgcn_readlane(rids, 0) < 0) return;
    int w_cur = row_words(__builtin_amdgcn_readlane(rids, 0));
    int gm_cur = width_of(w_cur);
    int2 pre_e = make_int2(0, 0);
    real pre_av = 0;
    {
        const int j = first_entry(0, w_cur, gm_cur);
        if (j >= 0) {
            const int c = __builtin_nontemporal_load(acol + j);
            pre_av = __builtin_nontemporal_load(aval + j);
            const I2 r = *reinterpret_cast<const I2 *>(brpt + c);
            pre_e = make_int2(r.b, r.e);
        }
    }
    int rid_nxt = __builtin_amdgcn_readlane(rids, 1);
    int w_nxt = row_words(rid_nxt);
    for (int i = 0;; i++) {
        // ---- requests for the rows behind this one --------------------------------------------
        if (((i + 2) & 63) == 0) rids = rid_batch(i + 2);  // (one trip per 64 rows)
        const int rid2 = rid_nxt >= 0 ? __builtin_amdgcn_readlane(rids, (i + 2) & 63) : -1;
        const int w2 = row_words(rid2);
        const int gm_nxt = width_of(w_nxt);
        const int jn = first_entry(rid_nxt, w_nxt, gm_nxt);
        int cn = 0;
        real avn = 0;
        if (jn >= 0) {
            cn = __builtin_nontemporal_load(acol + jn);
            avn = __builtin_nontemporal_load(aval + jn);
        }
        // ---- this row ----------------------------------------------------------------------------
        int n = __builtin_amdgcn_readlane(w_cur, 1) - __builtin_amdgcn_readlane(w_cur, 0);
        int T = pow2_ceil(n + (n >> 1));
        T = T < 64 ? 64 : (T > TMAX ? TMAX : T);
        const int mask = T - 1;
        for (int q = lane; q < T; q += BS) {
            keys[q] = -1;
            vals[q] = 0;
        }
        if (lane == 0) s_ov.w.defer.n = 0;
        wave_lds_sync();
        auto consume = [&](const IVec &k, const RVec &v, int m, real sc) {
            int h[VW], fresh = 0;
            ht_insert_vec(keys, mask, k, m, h, fresh);
#pragma unroll
            for (int u = 0; u < VW; u++)
                if (u < m) unsafeAtomicAdd(vals + h[u], (acc_t)(sc * v.v[u]));
        };
        const bool mixed = (gm_cur & 256) != 0;
        walk_products<BS, true, VW, decltype(consume) &, DCAP>(acol, aval, brpt, bcol, bval, bnnz,
                                                               __builtin_amdgcn_readlane(w_cur, 2),
                                                               __builtin_amdgcn_readlane(w_cur, 3), gm_cur & 255,
                                                               s_ov.w.ext, s_ov.w.av, consume,
                                                               mixed ? &s_ov.w.defer : (DeferList<true, DCAP> *)nullptr,
                                                               8 * (gm_cur & 255) * VW, nullptr, &pre_e, &pre_av);
        if (mixed) {
            wave_lds_sync();
            const int nd = s_ov.w.defer.n < DCAP ? s_ov.w.defer.n : DCAP;
            for (int d = 0; d < nd; d++) {
                const int2 e = s_ov.w.defer.ext[d];
                const real av = s_ov.w.defer.av[d];
                for (int base = e.x + lane * VW; base < e.y; base += BS * VW) {
                    IVec k;
                    RVec v;
                    const int m = fetch_chunk<true, VW>(bcol, bval, base, e.y, bnnz, k, v);
                    if (m > 0) consume(k, v, m, av);
                }
            }
        }
        // ---- next row: B extent of the lane's entry (its column has had the whole walk to arrive) ----
        pre_e = make_int2(0, 0);
        pre_av = avn;
        if (jn >= 0) {
            const I2 r = *reinterpret_cast<const I2 *>(brpt + cn);
            pre_e = make_int2(r.b, r.e);
        }
        wave_lds_sync();
        // ---- compaction, sort, read-out (as k_num_tb) -------------------------------------------------
        const int off = __builtin_amdgcn_readlane(w_cur, 0);
        n = __builtin_amdgcn_readlane(w_cur, 1) - off;
        const int P = pow2_ceil(n);
        int filled = 0;  // one wavefront: the running count is a scalar
        for (int base = 0; base < T; base += BS) {
            const int key = keys[base + lane];
            const bool occ = key != -1;
            const unsigned long long m = __ballot(occ);
            if (occ) srt[filled + __popcll(m & ((1ull << lane) - 1ull))] = key;
            filled += __popcll(m);
        }
        for (int q = n + lane; q < P; q += BS) srt[q] = 0x7fffffff;
        wave_lds_sync();
        if (P > 1 && !(write_col & 2)) bitonic_sort_lds<BS>(srt, P);
        for (int q = lane; q < n; q += BS) {
            const int key = srt[q];
            int h = hash_slot(key, mask);
            while (keys[h] != key) h = (h + 1) & mask;
            if (write_col & 1) ccol[off + q] = key;
            cval[off + q] = (real)vals[h];
        }
        if (rid_nxt < 0) return;
        wave_lds_sync();  // the next row clears the tables this one has just read
        w_cur = w_nxt;
        gm_cur = gm_nxt;
        w_nxt = w2;
        rid_nxt = rid2;
    }
}

// bin 5: persistent workgroups, private (keys, values) slices of global slabs; the row is
// written UNSORTED into (tcol, tval) at its C offset and sorted afterwards by one rocprim
// segmented radix sort (calculate_value_col_bin_each_gl :929-1027).
template <int BS>
__global__ __launch_bounds__(BS) void k_num_global(const int *__restrict__ arpt,
                                                   const int *__restrict__ acol,
                                                   const real *__restrict__ aval,
                                                   const int *__restrict__ brpt,
                                                   const int *__restrict__ bcol,
                                                   const real *__restrict__ bval,
                                                   const int *__restrict__ crpt,
                                                   int *__restrict__ tcol, real *__restrict__ tval,
                                                   const int *__restrict__ row_perm, int bin_off,
                                                   int count, BinState *bs,
                                                   int *__restrict__ kslab, real *__restrict__ vslab,
                                                   long long slice, int *__restrict__ seg_beg,
                                                   int *__restrict__ seg_end)
{
    __shared__ int s_row;
    __shared__ int s_cnt;
    int *keys = kslab + (long long)blockIdx.x * slice;
    real *vals = vslab + (long long)blockIdx.x * slice;
    const int lane = threadIdx.x & 63;
    while (true) {
        __syncthreads();
        if (threadIdx.x == 0) {
            s_row = atomicAdd(&bs->queue_head, 1);
            s_cnt = 0;
        }
        __syncthreads();
        const int q = s_row;
        if (q >= count) break;
        const int rid = row_perm[bin_off + q];
        const int off = crpt[rid];
        const int n = crpt[rid + 1] - off;
        if (threadIdx.x == 0) {
            seg_beg[q] = off;
            seg_end[q] = off + n;
        }
        long long T = 64;
        while (T < 2LL * n) T <<= 1;
        if (T > slice) T = slice;
        const long long mask = T - 1;
        for (long long i = threadIdx.x; i < T; i += BS) {
            keys[i] = -1;
            vals[i] = 0;
        }
        __syncthreads();
        const int a_beg = arpt[rid], a_end = arpt[rid + 1];
        for (int j = a_beg + (threadIdx.x >> 6); j < a_end; j += BS / 64) {
            const int c = acol[j];
            const real av = aval[j];
            const int ke = brpt[c + 1];
            for (int k = brpt[c] + lane; k < ke; k += 64) {
                int fresh;
                const long long h = gt_find_or_insert(keys, mask, bcol[k], &fresh);
                unsafeAtomicAdd(vals + h, av * bval[k]);
            }
        }
        __syncthreads();
        for (long long base = (threadIdx.x >> 6) * 64; base < T; base += BS) {
            const int key = __hip_atomic_load(keys + base + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool occ = key != -1;
            const unsigned long long m = __ballot(occ);
            if (m) {
                int start = 0;
                if (lane == 0) start = atomicAdd(&s_cnt, __popcll(m));
                start = __shfl(start, 0);
                if (occ) {
                    const int pos = off + start + __popcll(m & ((1ull << lane) - 1ull));
                    tcol[pos] = key;
                    tval[pos] = __hip_atomic_load(vals + base + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
}

}  // namespace spgemm
}  // namespace nsp
