// spgemm/listed.h -- heavy numeric rows against the column list of the symbolic phase (round 3).
// Part of the spgemm_hash.hip translation unit (kernels are launched from its host code).
#pragma once
#include "common.h"

namespace nsp {
namespace spgemm {

// ===================================================================================
//  heavy numeric rows with a column list: flat walk, rank by search, no tiles
// ===================================================================================
// A row of more than 5461 non-zeros does not fit an LDS hash table.  Rounds 1 / 2 cut its column WINDOW into
// tiles (dense: 12288 columns; ranked: a 2^19-column bitmap + 6144 accumulators) and walked the sorted rows of
// B with one cursor per A entry, tile after tile.  On R-MAT-22 (rows of ~16 K non-zeros spread over 4 M
// columns) a row takes 8.6 tiles because the BITMAP is full long before the accumulators are, and a tile is
// 13 us of mostly latency: two cursor walks (dependent loads, 56 of 1024 threads own a cursor), a scan of 16 K
// words, an emission that reads them again -- 3 products per thread.  41.6 of the call's 85 ms.
//
// Here the symbolic phase has already written the row's columns, sorted, to a slab (common.h: bits_to_list),
// so the numeric phase knows the structure and only has to ACCUMULATE:
//   * a slice of up to kListSlice consecutive ENTRIES of the list -- whatever width they span -- sits in LDS
//     (columns + double accumulators: 12 bytes per non-zero);
//   * every product of the row is read once per slice by the flat walk (common.h: walk_products_flat: three
//     dependent round trips for the whole row, every lane busy, any row-length mix);
//   * a product's accumulator is the rank of its column in the slice: binary search in LDS, four searches
//     interleaved per lane; products outside the slice's column range are dropped before the search;
//   * the slice leaves with coalesced copies: no bitmap, no scan, no compaction, no sort.
// A 16 K-entry row is two slices instead of 8.6 tiles.  The price is that every slice reads ALL products of
// the row, so rows of more than kListMaxSlices slices (hub rows: up to 460 K non-zeros, millions of products)
// stay with the cursor kernels, which see every product once (heavy_tiled.h, heavy_ranked.h).
// Numeric-only re-runs take the same kernel with C.col as the list (list_off == nullptr).
// (kListSlice, kListMaxSlices, list_wanted: common.h)
constexpr int kListBuckets = 2048;

template <int BS>
__global__ __launch_bounds__(BS) void k_num_listed(const int *__restrict__ arpt, const int *__restrict__ acol,
                                                   const real *__restrict__ aval, const int *__restrict__ brpt,
                                                   const int *__restrict__ bcol, const real *__restrict__ bval,
                                                   const int *__restrict__ crpt, int *__restrict__ ccol,
                                                   real *__restrict__ cval, const int *__restrict__ row_perm,
                                                   int bin_off, int count, BinState *bs,
                                                   const int *__restrict__ tcol,
                                                   const long long *__restrict__ list_off, long long list_work,
                                                   const int *__restrict__ row_prod, int bnnz, int write_col)
{
    __shared__ __attribute__((aligned(16))) int s_cols[kListSlice];
    __shared__ __attribute__((aligned(16))) acc_t s_vals[kListSlice];
    __shared__ int2 s_ext[BS];
    __shared__ real s_av[BS];
    __shared__ FlatScratch<BS> s_flat;
    // first / one-past-last entry of every bucket of 2^shift columns of the slice: the search of a product starts
    // in its bucket (only buckets that hold an entry are ever looked up: every product's column is in the list)
    __shared__ unsigned short s_bfirst[kListBuckets], s_bend[kListBuckets];
    __shared__ int s_row;
    while (true) {
        __syncthreads();
        if (threadIdx.x == 0) s_row = atomicAdd(&bs->queue_head3, 1);
        __syncthreads();
        const int q = s_row;
        if (q >= count) break;
        const int rid = row_perm[bin_off + q];
        const int pos = crpt[rid];
        const int n = crpt[rid + 1] - pos;
        const long long lo = list_off ? list_off[rid] : (long long)pos;
        if (lo < 0 || !list_wanted(n, row_prod[rid], list_work)) continue;  // the cursor kernels' row
        const int *__restrict__ list = tcol + lo;
        const int S = (n + kListSlice - 1) / kListSlice;
        const int per = (((n + S - 1) / S) + 3) & ~3;
        const int a_beg = arpt[rid], a_end = arpt[rid + 1];
        for (int s0 = 0; s0 < n; s0 += per) {
            const int m = n - s0 < per ? n - s0 : per;
            for (int i = threadIdx.x; i < m; i += BS) {
                s_cols[i] = list[s0 + i];
                s_vals[i] = 0;
            }
            __syncthreads();
            const int c_lo = s_cols[0], c_hi = s_cols[m - 1];
            int shift = 0;
            while (((unsigned int)(c_hi - c_lo) >> shift) >= (unsigned int)kListBuckets) shift++;
            for (int i = threadIdx.x; i < m; i += BS) {
                const int bk = (s_cols[i] - c_lo) >> shift;
                if (i == 0 || ((s_cols[i - 1] - c_lo) >> shift) != bk) s_bfirst[bk] = (unsigned short)i;
                if (i == m - 1 || ((s_cols[i + 1] - c_lo) >> shift) != bk) s_bend[bk] = (unsigned short)(i + 1);
            }
            __syncthreads();
            walk_products_flat<BS, true>(
                acol, aval, brpt, bcol, bval, bnnz, a_beg, a_end, s_ext, s_av, &s_flat,
                [&](const IVec &k, const RVec &v, int cn, real sc) {
                    int col[VW], b[VW], len[VW];
                    bool in[VW];
                    int longest = 0;
#pragma unroll
                    for (int i = 0; i < VW; i++) {
                        in[i] = i < cn && k.v[i] >= c_lo && k.v[i] <= c_hi;
                        col[i] = in[i] ? k.v[i] : c_lo;
                        const int bk = (col[i] - c_lo) >> shift;
                        b[i] = s_bfirst[bk];
                        len[i] = in[i] ? (int)s_bend[bk] - b[i] : 1;
                        longest = len[i] > longest ? len[i] : longest;
                    }
                    // rank = the last entry of the bucket that is not beyond the column (the column IS in the list):
                    // the four searches of a chunk advance together, one LDS read each per step
                    while (longest > 1) {
#pragma unroll
                        for (int i = 0; i < VW; i++) {
                            const int half = len[i] >> 1;
                            const int t = s_cols[b[i] + half];
                            b[i] = (half > 0 && t <= col[i]) ? b[i] + half : b[i];
                            len[i] -= half;
                        }
                        longest -= longest >> 1;
                    }
#pragma unroll
                    for (int i = 0; i < VW; i++)
                        if (in[i]) unsafeAtomicAdd(s_vals + b[i], (acc_t)(sc * v.v[i]));
                });
            // (the walk ends with a workgroup barrier)
            if (write_col & 1)
                for (int i = threadIdx.x; i < m; i += BS) ccol[pos + s0 + i] = s_cols[i];
            for (int i = threadIdx.x; i < m; i += BS) cval[pos + s0 + i] = (real)s_vals[i];
            __syncthreads();
        }
    }
}

}  // namespace spgemm
}  // namespace nsp
