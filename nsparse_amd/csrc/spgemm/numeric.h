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
template <int BS, int TMAX, int PMAX>
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
    // prof (a build with -DNSPARSE_EXPERIMENTS and NSPARSE_TB_PROF=1; compiled out otherwise: the pointer and
    // the clock cost three scalar registers, and with them the 256-thread bin loses a wavefront per SIMD),
    // 100 MHz ticks of thread 0: 0 row record + clear, 1 walk, 2 compaction, 3 sort, 4 read-out; 5 rows
#ifdef NSPARSE_EXPERIMENTS
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
    // (persistent workgroups striding over the rows of the big-table bins, and a bucket sort instead of the bitonic
    //  network there, were measured in round 3 and lost: DESIGN 4.1)
    const int slot = xcd_row_slot(bin_size);
    if (slot < 0) return;
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
                                      ht_insert_vec(keys, mask, k, n, h, fresh);
#pragma unroll
                                      for (int i = 0; i < VW; i++)
                                          if (i < n) unsafeAtomicAdd(vals + h[i], (acc_t)(sc * v.v[i]));
                                  }, (FLAT && !(write_col & 4)) ? reinterpret_cast<FlatScratch<BS> *>(&s_ov.w.flat) : (FlatScratch<BS> *)nullptr,
                                  (write_col & 8) != 0);
    __syncthreads();
    tick(1);

    const int lane = threadIdx.x & 63;
    const int P = pow2_ceil(n);
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
    tick(3);

    for (int i = threadIdx.x; i < n; i += BS) {
        const int key = srt[i];
        int h = hash_slot(key, mask);
        while (keys[h] != key) h = (h + 1) & mask;
        if (write_col & 1) ccol[off + i] = key;
        cval[off + i] = (real)vals[h];
    }
    tick(4);
#ifdef NSPARSE_EXPERIMENTS
    if (prof && threadIdx.x == 0) atomicAdd(prof + 5, 1ull);
#endif
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

// nsparse_set_deterministic(1): the values of a finished structure once more, in ONE order.  One wavefront per row of
// C; lane l owns the entries l, l + 64, ... of the row and walks the A entries of the row in their stored order
// (every lane the same entry: broadcast loads), looking its column up in the B row -- by bisection when the rows of B
// ascend, else by a scan.  sum += a * b in that order: the result does not depend on scheduling, atomics or the bin a
// row went through.  The float build multiplies in float and sums in double like the accumulating kernels.
__global__ __launch_bounds__(256) void k_num_deterministic(const int *__restrict__ arpt, const int *__restrict__ acol,
                                                           const real *__restrict__ aval, const int *__restrict__ brpt,
                                                           const int *__restrict__ bcol, const real *__restrict__ bval,
                                                           const int *__restrict__ crpt, const int *__restrict__ ccol,
                                                           real *__restrict__ cval, int M, int b_sorted)
{
#pragma clang fp contract(off)  // multiply, round, add, round: what a sequential CPU loop without FMA computes
    const int row = (int)((blockIdx.x * 256u + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const int c0 = crpt[row], c1 = crpt[row + 1];
    const int a0 = arpt[row], a1 = arpt[row + 1];
    for (int p0 = c0; p0 < c1; p0 += 64) {
        const int p = p0 + lane;
        const int col = p < c1 ? ccol[p] : -1;
        acc_t sum = 0;
        for (int j = a0; j < a1; j++) {
            const int k = acol[j];
            const real av = aval[j];
            const int b0 = brpt[k], b1 = brpt[k + 1];
            if (col < 0 || b0 >= b1) continue;
            int hit = -1;
            if (b_sorted) {
                int lo = b0, hi = b1 - 1;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (bcol[mid] < col) lo = mid + 1;
                    else hi = mid;
                }
                if (bcol[lo] == col) hit = lo;
                if (hit >= 0) sum += (acc_t)(av * bval[hit]);
            } else {
                // unsorted B may even hold a column twice in a row: every occurrence counts, in stored order
                for (int q = b0; q < b1; q++)
                    if (bcol[q] == col) sum += (acc_t)(av * bval[q]);
            }
        }
        if (p < c1) cval[p] = (real)sum;
    }
}

}  // namespace spgemm
}  // namespace nsp
