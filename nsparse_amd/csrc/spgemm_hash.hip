// spgemm_hash.hip -- hash-table SpGEMM  C = A * B  (symbolic + numeric) for gfx950.
//
// Replaces (reference file:line):
//   spgemm_kernel_hash        cuda-c/src/kernel/kernel_spgemm_hash_d.cu:1035-1075  (and _s.cu)
//   set_max_bin / set_min_bin                                            :156-246
//   set_row_nnz (symbolic dispatch) + its kernels                        :266-622, :1077-1185
//   calculate_value_col_bin (numeric dispatch) + its kernels             :631-1027, :1187-1288
//   get_spgemm_flop           cuda-c/src/kernel/kernel_spgemm_cu_csr.cu:18-57
//   SpGEMM_Hash_Numeric       cuda-cpp/inc/HashSpGEMM_volta.hpp:1018-1031 (numeric-only re-run)
//
// What is kept from the reference: the two-phase algorithm (count distinct columns per
// row with a hash table, scan, then accumulate values and emit columns in ascending
// order), the multiplicative hash with linear probing, and binning of rows by size so
// that every bin gets its own LDS budget.  Everything else is designed for CDNA4:
//
//  * wave64.  Inside a row the threads of a workgroup are cut into groups of g lanes,
//    g = pow2_ceil(average length of the B rows this C row touches), 1 <= g <= 64, chosen
//    per row at run time.  A group walks one B row with coalesced loads; short B rows get
//    narrow groups (many A entries in flight per wave), long ones a whole wavefront.  The
//    reference needs two code paths for this (4-lane "pwarp" rows vs warp-per-A-entry).
//  * LDS ladder re-derived for 160 KiB/CU (spgemm_hash_kernel_gen.c:51-91 gives the rule:
//    largest table that still leaves the CU occupied, halve downwards).  Symbolic tables
//    reach 32768 keys (128 KiB); the reference stops at 8192 and sends everything above to
//    global memory.  The table actually cleared and probed is sized per ROW
//    (pow2 >= 1.5 n for numeric, pow2 >= n_prod for symbolic), not per bin, so a small row
//    in a big bin does not pay for the bin's worst case.
//  * compaction by ballot + popcount inside the wave (no global cursor, no d_row_nz
//    reuse: the reference's atomicAdd(d_nz+rid,1) compaction is racy without warp
//    lock-step), column sort by an LDS bitonic network on 32-bit keys whose
//    sub-wave stages run without workgroup barriers (the reference's rank sort is
//    O(nz^2), 16.7 M compares for a 4096-entry row), values fetched afterwards by probing
//    the still-intact table.
//  * rows that do not fit LDS: persistent workgroups pull rows from a queue and hash into a
//    private slice of one bounded global slab (tables sized per row, cleared per row), the
//    unsorted result is ordered by one rocprim::segmented_radix_sort_pairs.  Memory is
//    O(workgroups * largest table), not O(rows * largest table) as in the reference
//    (kernel_spgemm_hash_d.cu:1156-1170,1258-1281).
//  * one D2H of (bin sizes, max, nnz) per phase through pinned memory instead of the
//    reference's >= 9 blocking cudaMemcpy; all scratch comes from the block cache.
//
// Results: C.rpt / C.col are bit-identical to the reference by construction (distinct
// columns per row, ascending); C.val differs only by floating-point summation order
// (LDS atomics), same as the reference vs cuSPARSE (tolerance 1e-9 double / 1e-6 float,
// nsparse.cu:300-353).
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>

#include <algorithm>
#include <cstring>
#include <type_traits>

#include "internal.h"

namespace nsp {
namespace spgemm {

constexpr int NB = kMaxBins;

// ---- bin ladders ------------------------------------------------------------------
// Symbolic, n = intermediate products of the row (upper bound of its nnz):
//   bin 0  n <= 32      sub-wave rows, 4 lanes per row, 64-key table per row
//   bin 1  n <= 512     one workgroup per row,   64 threads, table <=   512 keys ( 2 KiB)
//   bin 2  n <= 2048                            128 threads,        <=  2048      ( 8 KiB)
//   bin 3  n <= 8192                            256 threads,        <=  8192      (32 KiB)
//   bin 4  n <= 32768                          1024 threads,        <= 32768      (128 KiB)
//   bin 5  n  > 32768   1024 threads, 32768 keys, row FAILS over to the global table when
//                       it holds more than 24576 distinct keys
// Numeric, n = exact nnz of the C row, table = pow2_ceil(1.5 n) (load factor <= 2/3):
//   bin 0  n <= 16      sub-wave rows, 4 lanes per row, 32 slots per row
//   bin 1  n <= 170     64 threads,  table <=  256 slots
//   bin 2  n <= 682     256 threads, table <= 1024
//   bin 3  n <= 2730    512 threads, table <= 4096
//   bin 4  n <= 5461    1024 threads, table <= 8192  (96 KiB fp64 + 32 KiB sort keys)
//   bin 5  n  > 5461    global-memory tables
// Row -> bin.  Bin 0: tiny rows (sub-wave kernels).  Bins 1..5: hash tables sized by n.
// Bins 9..10 (symbolic only): BIT WINDOW rows -- same idea with one bit per column, for rows
// with many products whose window is too wide for byte flags (up to 2^20 columns = 128 KiB).
// Bins 6..8: DENSE WINDOW rows -- the columns a C row can touch lie in [lo, lo+span) and
// span is small enough for an LDS array indexed by (col - lo): no probing, no compare-and-swap
// with return, no sort (see k_sym_dense / k_num_dense).  A row is dense-eligible when
// span <= dense_span[2] and span <= dense_ratio * n (clearing and scanning the window must not
// cost more than the products).
struct Thr {
    int tiny;            // n <= tiny           -> bin 0
    int hash_t[4];       // n <= hash_t[k]      -> bin 1 + k, above -> bin 5
    int dense_span[3];   // span <= dense_span[k] -> bin 6 + k
    int dense_ratio;     // 0 disables the dense bins
    int bits_span[2];    // symbolic only: span <= bits_span[k] -> bin 9 + k (1 bit per column)
    int bits_ratio;      // span <= bits_ratio * n; 0 disables
    int bits_min;        // only rows with n > bits_min (small rows hash faster than they clear)
    int bits_wide_min;   // rows with n > bits_wide_min (they would fill the two largest hash tables to
    int bits_wide_span;  // the brim, or overflow them) and span <= bits_wide_span -> bin 10, which
                         // then covers the window in pieces of bits_span[1] columns; 0 disables
};
constexpr Thr kSymThr = {32,   {512, 2048, 8192, 32768}, {4096, 16384, 65536}, 8, {262144, 1048576}, 64, 2048,
                         8192, 16 * 1048576};
constexpr Thr kNumThr = {16, {170, 682, 2730, 5461}, {1536, 4096, 12288}, 8, {0, 0}, 0, 0, 0, 0};
constexpr int kSymLargeBin = 5;
constexpr int kNumGlobalBin = 5;
// Setup kernels: rows longer than kLongFactor * W entries are not walked by their W-lane group
// (a 4700-entry row on 4 lanes is a millisecond of serial dependent gathers): the bulk pass
// appends them to a short device list and a second, fixed-size launch walks them with 64 lanes.
constexpr int kLongFactor = 32;
constexpr int kLongCap = 1 << 16;
constexpr int kDenseBin0 = 6;
constexpr int kBitsBin0 = 9;
constexpr int kSetupMaxGrid = 16384;
constexpr int kPartialStride = 16;  // long longs per block: hist[NB], max, total, bm, alen
constexpr int kSymLargeT = 32768;
constexpr int kSymLargeLimit = 24576;
static int g_dense_enabled = -1;  // -1: read NSPARSE_DENSE on first use
static int g_sorted = 1;          // 0: hash rows are written in table order (nsparse_spgemm_set_sorted)

// device-resident counters of one binning pass (lives in Context::d_scratch)
struct BinState {
    int hist[NB];
    int cursor[NB];
    int maxv;
    int fail_count;
    int queue_head;
    int nnz;
    long long total;
    long long bm_total;  // words of column bitmaps (dense window rows)
    long long max_alen;  // longest row of A
    int b_unsorted;      // some row of B does not have strictly ascending columns
    int queue_head2;     // second persistent-kernel queue of the heavy numeric bin
};

struct Stats {
    nsparse_spgemm_stats s;
};
static Stats g_stats;

__host__ __device__ __forceinline__ int bin_of(int n, int span, const Thr &thr)
{
    if (n <= thr.tiny) return 0;
    if (thr.dense_ratio > 0 && span > 0 && span <= thr.dense_span[2] &&
        (long long)span <= (long long)thr.dense_ratio * n)
        return kDenseBin0 + (span > thr.dense_span[0]) + (span > thr.dense_span[1]);
    if (thr.bits_ratio > 0 && n > thr.bits_min && span > 0 && span <= thr.bits_span[1] &&
        (long long)span <= (long long)thr.bits_ratio * n)
        return kBitsBin0 + (span > thr.bits_span[0]);
    if (thr.bits_ratio > 0 && thr.bits_wide_min > 0 && n > thr.bits_wide_min && span > 0 &&
        span <= thr.bits_wide_span)
        return kBitsBin0 + 1;
    int b = 1;
#pragma unroll
    for (int q = 0; q < 4; q++) b += (n > thr.hash_t[q]) ? 1 : 0;
    return b;
}

// Slot of a column id in a table of mask + 1 = 2^L slots: the TOP L bits of key * 2^32/phi
// (Fibonacci hashing).  The reference takes the LOW bits of key * 107
// (kernel_spgemm_hash_d.cu:30,296), which only see the low bits of the key: column ids that
// are multiples of a large power of two -- a large share of an R-MAT row, whose index bits are
// 0 with probability 0.76 -- all start probing at the same slot, and a scale-22 row of 25 K
// columns degenerates into long linear-probe clusters (measured, R-MAT scale 22: symbolic
// 665 -> 116 ms, whole call 1076 -> 458 ms).  The top bits depend on every bit of the key; consecutive
// columns land 0.618 * 2^L slots apart, so FEM rows spread evenly as well.  The table
// contents differ from the reference's, the rows that come out of them do not.
__device__ __forceinline__ int hash_slot(int key, int mask)
{
    return (int)(((unsigned)key * 0x9E3779B1u) >> __builtin_clz((unsigned)mask));
}

__device__ __forceinline__ int pow2_ceil(int v) { return v <= 1 ? 1 : (1 << (32 - __clz(v - 1))); }

__device__ __forceinline__ int lds_load(const int *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Insert `key` into an open-addressing table of (mask+1) ints (empty = -1), linear
// probing.  Returns the slot; *fresh = 1 when this call created the entry.
// One LDS compare-and-swap per probe and nothing else: CAS(slot, -1, key) returns -1 (we
// inserted), key (already there, nothing written) or another key (next slot).  The
// read-then-CAS form of the reference costs three times the instructions on CDNA (nested
// exec-mask regions) and the kernels are issue-bound, not LDS-bound.
__device__ __forceinline__ int ht_find_or_insert(int *tab, int mask, int key, int *fresh)
{
    int h = hash_slot(key, mask);
    while (true) {
        const int old = atomicCAS(tab + h, -1, key);
        if (old == -1 || old == key) {
            *fresh = old == -1;
            return h;
        }
        h = (h + 1) & mask;
    }
}

// Same on a table in global memory.  Only the value returned by the CAS decides, so a
// stale L1 line (another CU cannot touch this slice, but atomics execute in L2) can at
// worst cost one extra CAS.
__device__ __forceinline__ long long gt_find_or_insert(int *tab, long long mask, int key, int *fresh)
{
    long long h = (long long)(((unsigned long long)(unsigned)key * 0x9E3779B97F4A7C15ull) >>
                              __builtin_clzll((unsigned long long)mask));
    *fresh = 0;
    while (true) {
        const int cur = __hip_atomic_load(tab + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == key) return h;
        if (cur == -1) {
            const int old = atomicCAS(tab + h, -1, key);
            if (old == -1) { *fresh = 1; return h; }
            if (old == key) return h;
        }
        h = (h + 1) & mask;
    }
}

__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Every lane takes VW consecutive entries of a B row per step: one 16-byte column load and
// VW/2 16-byte value loads instead of VW scalar pairs, and the walk's bookkeeping (the kernels
// are instruction-issue-bound: rocprofv3 shows VALU 73 % / SALU 83 % busy, LDS 15 %) is paid
// once per VW products.
constexpr int VW = 4;
struct __attribute__((aligned(4))) IVec {
    int v[VW];
};
struct __attribute__((aligned(sizeof(real) < 8 ? 4 : 8))) RVec {
    real v[VW];
};

// Lanes per B row for a C row with `np` products spread over `alen` entries of A, for a
// workgroup of BS threads.  With g lanes per group the row takes
//     ceil(alen / (BS/g)) * ceil(avg_len / (g*VW))   group steps,
// so g trades padding of the B rows (small g pads less) against imbalance between groups
// (large g, few groups).  The largest g with the fewest steps wins.
__device__ __forceinline__ int group_width(int np, int alen, int BS, int maxb = 0)
{
    if (alen <= 0) return 64;
    const int avg = (np + alen - 1) / alen;
    int best_g = 64, best_t = 0x7fffffff;
#pragma unroll
    for (int g = 64; g >= 1; g >>= 1) {
        const int ng = BS / g;
        int t = ((alen + ng - 1) / ng) * ((avg + g * VW - 1) / (g * VW));
        // the group that owns the longest B row of this C row (maxb entries) cannot finish
        // earlier than that row alone takes: on power-law inputs (hub rows of hundreds of entries
        // among rows of three) this term, not the average, decides
        const int tl = (maxb + g * VW - 1) / (g * VW);
        t = t > tl ? t : tl;
        if (t < best_t) { best_t = t; best_g = g; }
    }
    return best_g;
}

// Workgroup b runs on XCD b % 8 (observed dispatch order, not a contract: only speed depends
// on it).  Rows of a bin are listed in roughly ascending order and neighbouring rows of A touch
// the same rows of B, so XCD x is given the x-th contiguous eighth of the bin: its private 4 MiB
// L2 then holds one window of B instead of all of it (measured before: 1.35 GB fetched per
// numeric launch for 0.1 GB of B).  Launch with 8 * ceil(n / 8) workgroups.
__device__ __forceinline__ int xcd_row_slot(int n)
{
    const int nb8 = (n + 7) >> 3;
    const int slot = (int)(blockIdx.x & 7) * nb8 + (int)(blockIdx.x >> 3);
    return slot < n ? slot : -1;
}

__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Walk every intermediate product of one C row with the threads of a workgroup and hand
// (column, aval * bval) to `consume`.
//
// The naive loop (per A entry: load A.col -> load B.rpt[c], B.rpt[c+1] -> load B.col/B.val)
// is a chain of three dependent global loads per A entry, 2-3 k cycles of latency for one or
// two wave-steps of work.  Here each lane of a group loads ONE A entry and its B row extent
// and parks them in LDS (`s_ext`, `s_av`: one slot per thread), so a batch of g entries costs
// the two dependent latencies once.  The group then runs a flat state machine over
// (entry, chunk-of-g) steps in which the loads of step s+1 are issued before step s is
// hashed, so every B chunk is in flight for a whole hashing step.
// VW entries of a B row starting at i0; returns how many of them belong to the row (< ke).
// The load is always the full 16-byte vector: elements past ke belong to the next row of B
// (valid memory, masked out by the returned count); only the last VW-1 entries of the whole
// array (i0 + VW > bnnz) take the element-wise path.
template <bool WITH_VAL>
__device__ __forceinline__ int fetch_chunk(const int *__restrict__ bcol, const real *__restrict__ bval,
                                           int i0, int ke, int bnnz, IVec &k, RVec &v)
{
    int n = ke - i0;
    n = n < 0 ? 0 : (n > VW ? VW : n);
    if (n > 0) {
        if (i0 + VW <= bnnz) {
            // unsigned index: zero-extension is free, so the loads use base + 32-bit offset
            k = *reinterpret_cast<const IVec *>(bcol + (unsigned)i0);
            if (WITH_VAL) v = *reinterpret_cast<const RVec *>(bval + (unsigned)i0);
        } else {
#pragma unroll
            for (int i = 0; i < VW; i++) {
                const int ii = i < n ? i0 + i : i0;
                k.v[i] = bcol[ii];
                if (WITH_VAL) v.v[i] = bval[ii];
            }
        }
    }
    return n;
}

template <int BS, bool WITH_VAL, typename F>
__device__ __forceinline__ void walk_products(const int *__restrict__ acol, const real *__restrict__ aval,
                                              const int *__restrict__ brpt, const int *__restrict__ bcol,
                                              const real *__restrict__ bval, int bnnz, int a_beg,
                                              int a_end, int g, int2 *s_ext, real *s_av, F &&consume)
{
    const int ngroups = BS / g;
    const int gid = threadIdx.x / g, gl = threadIdx.x % g;
    const int first = a_beg + gid;
    const int cnt = first < a_end ? (a_end - first + ngroups - 1) / ngroups : 0;
    int2 *ext = s_ext + gid * g;
    real *avs = s_av + gid * g;
    const int lane_off = gl * VW;
    const int stride = g * VW;
    for (int b0 = 0; b0 < cnt; b0 += g) {
        const int m = b0 + gl;
        int2 e = make_int2(0, 0);
        real av = 0;
        if (m < cnt) {
            const int j = first + m * ngroups;
            const int c = __builtin_nontemporal_load(acol + j);
            if (WITH_VAL) av = __builtin_nontemporal_load(aval + j);
            struct __attribute__((aligned(4))) I2 {
                int b, e;
            };
            const I2 r = *reinterpret_cast<const I2 *>(brpt + c);  // one 8-byte gather
            e.x = r.b;
            e.y = r.e;
        }
        ext[gl] = e;
        if (WITH_VAL) avs[gl] = av;
        wave_lds_sync();  // a group never spans wavefronts: in-order LDS is enough
        const int nb = cnt - b0 < g ? cnt - b0 : g;
        int t = 0;
        int2 cur = ext[0];
        real cav = WITH_VAL ? avs[0] : (real)0;
        int base = cur.x;
        IVec pk;
        RVec pv;
        int pn = fetch_chunk<WITH_VAL>(bcol, bval, base + lane_off, cur.y, bnnz, pk, pv);
        while (t < nb) {
            const IVec ck = pk;
            const RVec cv = pv;
            const int cn = pn;
            const real sc = cav;
            base += stride;
            if (base >= cur.y) {
                t++;
                if (t < nb) {
                    cur = ext[t];
                    if (WITH_VAL) cav = avs[t];
                    base = cur.x;
                }
            }
            pn = t < nb ? fetch_chunk<WITH_VAL>(bcol, bval, base + lane_off, cur.y, bnnz, pk, pv) : 0;
            if (cn > 0) consume(ck, cv, cn, sc);
        }
        wave_lds_sync();
    }
}

// VW find-or-insert operations of one lane, the first probes issued back to back
__device__ __forceinline__ void ht_insert_vec(int *tab, int mask, const IVec &k, int n, int (&h)[VW], int &fresh)
{
    int old[VW];
#pragma unroll
    for (int i = 0; i < VW; i++) h[i] = hash_slot(k.v[i], mask);
#pragma unroll
    for (int i = 0; i < VW; i++) old[i] = i < n ? atomicCAS(tab + h[i], -1, k.v[i]) : k.v[i];
#pragma unroll
    for (int i = 0; i < VW; i++) {
        while (old[i] != -1 && old[i] != k.v[i]) {
            h[i] = (h[i] + 1) & mask;
            old[i] = atomicCAS(tab + h[i], -1, k.v[i]);
        }
        fresh += old[i] == -1;
    }
}

// ===================================================================================
//  setup: products per row, histogram, bin-grouped row permutation
// ===================================================================================

// W lanes cooperate on one row of A (W = pow2 <= 64 chosen from the average row length so
// that the A.col loads of a wave coalesce).  Restates set_intprod_num (:70-86) fused with
// set_bin (:88-112) and with the flop sum of get_spgemm_flop.
// One 16-byte record per row of B: where it starts, how long it is, and its smallest / largest
// column id (rows need not be sorted).  Every later stage reaches a B row through ONE gather of
// this record instead of two B.rpt loads (+ two window loads): the column window of a C row is
// the union of the windows of the B rows it touches.
struct __attribute__((aligned(16))) BInfo {
    int start, len, lo, hi;
};

template <int W>
__global__ __launch_bounds__(256) void k_b_info(const int *__restrict__ brpt, const int *__restrict__ bcol,
                                                int K, BInfo *__restrict__ info, BinState *bs,
                                                int *__restrict__ long_list, int *long_cnt, int long_len,
                                                const int *__restrict__ todo)
{
    // W lanes per row of B (W from the average row length, so the column loads coalesce).
    // todo == nullptr: bulk pass over all rows; rows longer than long_len are deferred to
    // long_list.  todo != nullptr: pass over the deferred rows todo[0 .. min(*long_cnt, cap)).
    constexpr int RPB = 256 / W;
    const int lane = threadIdx.x % W;
    const int nrows = todo ? (*long_cnt < kLongCap ? *long_cnt : kLongCap) : K;
    for (int base = blockIdx.x * RPB; base < nrows; base += gridDim.x * RPB) {
        const int q = base + (int)threadIdx.x / W;
        const int r = q < nrows ? (todo ? todo[q] : q) : -1;
        int lo = 0x7fffffff, hi = -1, b = 0, e = 0;
        bool bad = false;
        if (r >= 0) {
            b = brpt[r];
            e = brpt[r + 1];
        }
        int ok = 0;
        if (r >= 0 && !todo && long_list && e - b > long_len && lane == 0) {
            const int idx = atomicAdd(long_cnt, 1);
            ok = idx < kLongCap;
            if (ok) long_list[idx] = r;
        }
        const bool defer = __shfl(ok, 0, W) != 0;
        if (r >= 0 && !defer) {
            for (int k = b + lane; k < e; k += W) {
                const int c = bcol[k];
                if (k > b) bad |= c <= bcol[k - 1];  // strictly ascending?  (neighbour is in cache)
                lo = c < lo ? c : lo;
                hi = c > hi ? c : hi;
            }
        }
#pragma unroll
        for (int o = W / 2; o >= 1; o >>= 1) {
            const int l = __shfl_xor(lo, o), h = __shfl_xor(hi, o);
            lo = l < lo ? l : lo;
            hi = h > hi ? h : hi;
        }
        if (bad) atomicOr(&bs->b_unsorted, 1);
        if (r >= 0 && !defer && lane == 0) {
            BInfo o;
            o.start = b;
            o.len = e - b;
            o.lo = lo;
            o.hi = hi;
            info[r] = o;
        }
    }
}

template <int W>
__global__ __launch_bounds__(256) void k_row_products(const int *__restrict__ arpt,
                                                      const int *__restrict__ acol,
                                                      const BInfo *__restrict__ binfo, int M,
                                                      int *__restrict__ row_prod,
                                                      int *__restrict__ row_lo,
                                                      int *__restrict__ row_span,
                                                      int *__restrict__ bm_words, int bm_span_max,
                                                      Thr thr, long long *__restrict__ partial,
                                                      int *__restrict__ row_span_num,
                                                      int *__restrict__ row_nz,
                                                      int *__restrict__ row_maxb,
                                                      int *__restrict__ long_list, int *long_cnt,
                                                      int long_len, const int *__restrict__ todo)
{
    // todo == nullptr: bulk pass, rows of A longer than long_len are deferred to long_list;
    // todo != nullptr: the deferred rows (see kLongFactor)
    const int nrows = todo ? (*long_cnt < kLongCap ? *long_cnt : kLongCap) : M;
    if (!todo && blockIdx.x == 0 && threadIdx.x == 0) {  // scan tails (instead of two memset launches)
        bm_words[M] = 0;
        row_nz[M] = 0;
    }
    __shared__ int s_hist[NB];
    __shared__ int s_max;
    __shared__ unsigned long long s_total;
    __shared__ unsigned long long s_bm;
    __shared__ int s_alen;
    if (threadIdx.x < NB) s_hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) { s_max = 0; s_total = 0; s_bm = 0; s_alen = 0; }
    __syncthreads();
    const int lane = threadIdx.x % W;
    constexpr int RPB = 256 / W;
    // per-thread statistics, folded once per wave at the end: one LDS atomic per row would
    // serialise when (almost) every row falls into the same bin (1 M-row power-law inputs)
    int t_max = 0, t_alen = 0;
    unsigned long long t_total = 0, t_bm = 0;
    // grid-stride over rows
    for (int base = blockIdx.x * RPB; base < nrows; base += gridDim.x * RPB) {
        const int q = base + (int)threadIdx.x / W;
        int row = q < nrows ? (todo ? todo[q] : q) : M;
        long long n = 0;
        int lo = 0x7fffffff, hi = -1, mb = 0;
        {
            int ok = 0;
            if (row < M && !todo && long_list && arpt[row + 1] - arpt[row] > long_len && lane == 0) {
                const int idx = atomicAdd(long_cnt, 1);
                ok = idx < kLongCap;
                if (ok) long_list[idx] = row;
            }
            if (__shfl(ok, 0, W) != 0) row = M;  // deferred: nothing to do for this group now
        }
        if (row < M) {
            const int e = arpt[row + 1];
            int j = arpt[row] + lane;
            for (; j + 3 * W < e; j += 4 * W) {  // four independent gathers in flight
                int c[4];
                BInfo bi[4];
#pragma unroll
                for (int u = 0; u < 4; u++) c[u] = __builtin_nontemporal_load(acol + j + u * W);
#pragma unroll
                for (int u = 0; u < 4; u++) bi[u] = binfo[c[u]];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    n += bi[u].len;
                    mb = bi[u].len > mb ? bi[u].len : mb;
                    lo = bi[u].lo < lo ? bi[u].lo : lo;
                    hi = bi[u].hi > hi ? bi[u].hi : hi;
                }
            }
            for (; j < e; j += W) {
                const int c = __builtin_nontemporal_load(acol + j);
                const BInfo bi = binfo[c];
                n += bi.len;
                mb = bi.len > mb ? bi.len : mb;
                lo = bi.lo < lo ? bi.lo : lo;
                hi = bi.hi > hi ? bi.hi : hi;
            }
        }
#pragma unroll
        for (int o = W / 2; o >= 1; o >>= 1) {
            n += __shfl_xor(n, o);
            const int l = __shfl_xor(lo, o), h = __shfl_xor(hi, o), m2 = __shfl_xor(mb, o);
            lo = l < lo ? l : lo;
            hi = h > hi ? h : hi;
            mb = m2 > mb ? m2 : mb;
        }
        int bin = -1;
        if (row < M && lane == 0) {
            const int ni = n > 0x7fffffffLL ? 0x7fffffff : (int)n;  // saturate (hub rows)
            const long long sp = hi >= lo ? (long long)hi - lo + 1 : 0;
            const int span = sp > 0x7fffffffLL ? 0x7fffffff : (int)sp;
            row_prod[row] = ni;
            row_lo[row] = hi >= lo ? lo : 0;
            row_span[row] = span;
            row_maxb[row] = mb;
            // words of the column bitmap the symbolic dense kernel hands to the numeric one
            const int bw = (span > 0 && span <= bm_span_max) ? (span + 31) >> 5 : 0;
            bm_words[row] = bw;
            row_span_num[row] = 0;  // set by k_sym_dense when it hands a bitmap over
            bin = bin_of(ni, span, thr);
            const int al = arpt[row + 1] - arpt[row];
            if (W >= 16) {  // at most 4 rows per wave: direct LDS atomics are cheapest
                atomicAdd(&s_hist[bin], 1);
                atomicMax(&s_max, ni);
                atomicMax(&s_alen, al);
                atomicAdd(&s_total, (unsigned long long)n);
                if (bw) atomicAdd(&s_bm, (unsigned long long)bw);
            } else {
                t_max = ni > t_max ? ni : t_max;
                t_alen = al > t_alen ? al : t_alen;
                t_total += (unsigned long long)n;
                t_bm += (unsigned long long)bw;
            }
        }
        if (W < 16) {
            // histogram: one LDS atomic per (wave, bin present in the wave)
            unsigned long long todo = __ballot(bin >= 0);
            while (todo) {
                const int leader = __ffsll((long long)todo) - 1;
                const int b = __shfl(bin, leader);
                const unsigned long long same = __ballot(bin == b);
                if ((threadIdx.x & 63) == leader) atomicAdd(&s_hist[b], __popcll(same));
                todo &= ~same;
            }
        }
    }
    if (W < 16) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const int m1 = __shfl_xor(t_max, o), m2 = __shfl_xor(t_alen, o);
            t_max = m1 > t_max ? m1 : t_max;
            t_alen = m2 > t_alen ? m2 : t_alen;
            t_total += __shfl_xor(t_total, o);
            t_bm += __shfl_xor(t_bm, o);
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMax(&s_max, t_max);
            atomicMax(&s_alen, t_alen);
            atomicAdd(&s_total, t_total);
            atomicAdd(&s_bm, t_bm);
        }
    }
    __syncthreads();
    // Per-block partials with plain stores; k_reduce_partials folds them.  (Same-address
    // device-scope atomics from thousands of workgroups serialise at ~20 ns each on the
    // 8-XCD part: 0.38 ms for 15 K blocks, 0.14 ms for 2 K -- measured.)
    long long *out = partial + (long long)blockIdx.x * kPartialStride;
    if (threadIdx.x < NB) out[threadIdx.x] = s_hist[threadIdx.x];
    if (threadIdx.x == 0) {
        out[NB] = s_max;
        out[NB + 1] = (long long)s_total;
        out[NB + 2] = (long long)s_bm;
        out[NB + 3] = s_alen;
    }
}

__global__ __launch_bounds__(256) void k_reduce_partials(const long long *__restrict__ partial, int nblocks,
                                                         BinState *bs)
{
    __shared__ unsigned long long s_acc[kPartialStride];
    __shared__ int s_max;
    __shared__ int s_alen;
    if (threadIdx.x < kPartialStride) s_acc[threadIdx.x] = 0;
    if (threadIdx.x == 0) { s_max = 0; s_alen = 0; }
    __syncthreads();
    // a few dozen workgroups, each folds a slice of the partials and issues one global atomic
    // per field: thread t handles field (t % 16) of partials t/16, t/16 + 16, ... of its slice
    const int per = (nblocks + gridDim.x - 1) / gridDim.x;
    const int b0 = blockIdx.x * per, b1 = b0 + per < nblocks ? b0 + per : nblocks;
    const int f = threadIdx.x & 15;
    if (f < NB + 4) {
        long long acc = 0;
        for (int b = b0 + (threadIdx.x >> 4); b < b1; b += 16) {
            const long long v = partial[(long long)b * kPartialStride + f];
            acc = (f == NB || f == NB + 3) ? (v > acc ? v : acc) : acc + v;
        }
        if (f == NB) atomicMax(&s_max, (int)acc);
        else if (f == NB + 3) atomicMax(&s_alen, (int)acc);
        else if (acc) atomicAdd(&s_acc[f], (unsigned long long)acc);
    }
    __syncthreads();
    if (threadIdx.x < NB && s_acc[threadIdx.x]) atomicAdd(&bs->hist[threadIdx.x], (int)s_acc[threadIdx.x]);
    if (threadIdx.x == 0) {
        if (s_max) atomicMax(&bs->maxv, s_max);
        if (s_acc[NB + 1]) atomicAdd((unsigned long long *)&bs->total, s_acc[NB + 1]);
        if (s_acc[NB + 2]) atomicAdd((unsigned long long *)&bs->bm_total, s_acc[NB + 2]);
        if (s_alen) atomicMax((unsigned long long *)&bs->max_alen, (unsigned long long)s_alen);
    }
}

// Copy a counter block to mapped host memory and raise a sequence flag: the host polls the
// flag instead of paying hipMemcpyAsync + hipStreamSynchronize (~40 us per round trip here).
__global__ __launch_bounds__(64) void k_publish(const BinState *__restrict__ src, int *dst, int words,
                                                const int *__restrict__ nnz_src, int *flag, int seq)
{
    const int *s = reinterpret_cast<const int *>(src);
    for (int i = threadIdx.x; i < words; i += 64) dst[i] = s[i];
    if (nnz_src && threadIdx.x == 0) reinterpret_cast<BinState *>(dst)->nnz = *nnz_src;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// histogram of an existing per-row count (numeric binning, set_min_bin :201-246)
__global__ __launch_bounds__(1024) void k_hist(const int *__restrict__ n, const int *__restrict__ span,
                                              int M, Thr thr, BinState *bs)
{
    __shared__ int s_hist[NB];
    __shared__ int s_max;
    __shared__ unsigned long long s_sum;  // 64-bit: the int scan of the same numbers may wrap
    if (threadIdx.x < NB) s_hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        s_max = 0;
        s_sum = 0;
    }
    __syncthreads();
    const int i = blockIdx.x * 1024 + threadIdx.x;  // 1024 rows per block: 4x fewer same-address
    int bin = -1, v = 0;                             // global atomics at the end
    if (i < M) {
        v = n[i];
        bin = bin_of(v, span[i], thr);
    }
    unsigned long long todo = __ballot(bin >= 0);
    while (todo) {  // one LDS atomic per (wave, bin present in the wave)
        const int leader = __ffsll((long long)todo) - 1;
        const int b = __shfl(bin, leader);
        const unsigned long long same = __ballot(bin == b);
        if ((threadIdx.x & 63) == leader) atomicAdd(&s_hist[b], __popcll(same));
        todo &= ~same;
    }
    unsigned long long sum = (unsigned long long)v;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const int m = __shfl_xor(v, o);
        v = m > v ? m : v;
        sum += __shfl_xor(sum, o);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&s_max, v);
        atomicAdd(&s_sum, sum);
    }
    __syncthreads();
    if (threadIdx.x < NB && s_hist[threadIdx.x]) atomicAdd(&bs->hist[threadIdx.x], s_hist[threadIdx.x]);
    if (threadIdx.x == 0) {
        atomicMax(&bs->maxv, s_max);
        if (s_sum) atomicAdd((unsigned long long *)&bs->total, s_sum);
    }
}

__global__ __launch_bounds__(256) void k_row_len(const int *__restrict__ rpt, int *__restrict__ len, int M)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < M) len[i] = rpt[i + 1] - rpt[i];
}

// rows grouped by bin (set_row_perm :125-154): one LDS pass ranks the rows of a block
// inside their bin, one global atomic per (block, bin) reserves the range.
__global__ __launch_bounds__(1024) void k_bin_scatter(const int *__restrict__ n,
                                                     const int *__restrict__ span, int M, Thr thr,
                                                     BinState *bs, int *__restrict__ perm)
{
    __shared__ int s_cnt[NB];
    __shared__ int s_base[NB];
    if (threadIdx.x < NB) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int i = blockIdx.x * 1024 + threadIdx.x;
    int b = -1, r = 0;
    if (i < M) b = bin_of(n[i], span[i], thr);
    // rank inside the block: ballot + popcount inside the wave, one LDS atomic per (wave, bin)
    unsigned long long todo = __ballot(b >= 0);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int bb = __shfl(b, leader);
        const unsigned long long same = __ballot(b == bb);
        int base = 0;
        if (lane == leader) base = atomicAdd(&s_cnt[bb], __popcll(same));
        base = __shfl(base, leader);
        if (b == bb) r = base + __popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    if (b < 0) b = 0;
    __syncthreads();
    if (threadIdx.x < NB) {
        int off = 0;
        for (int q = 0; q < (int)threadIdx.x; q++) off += bs->hist[q];
        const int c = s_cnt[threadIdx.x];
        s_base[threadIdx.x] = off + (c ? atomicAdd(&bs->cursor[threadIdx.x], c) : 0);
    }
    __syncthreads();
    if (i < M) perm[s_base[b] + r] = i;
}

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
    for (int i = threadIdx.x; i < RPB * TROW; i += BS) tab[i] = -1;
    __syncthreads();
    const int lrow = threadIdx.x / LPR;
    const int sub = threadIdx.x % LPR;
    const int q = blockIdx.x * RPB + lrow;
    int cnt = 0;
    int rid = 0;
    if (q < bin_size) {
        rid = row_perm[bin_off + q];
        int *t = tab + lrow * TROW;
        const int e = arpt[rid + 1];
        for (int j = arpt[rid] + sub; j < e; j += LPR) {
            const int c = __builtin_nontemporal_load(acol + j);
            const int ke = brpt[c + 1];
            for (int k = brpt[c]; k < ke; k++) {
                int fresh;
                ht_find_or_insert(t, TROW - 1, bcol[k], &fresh);
                cnt += fresh;
            }
        }
    }
#pragma unroll
    for (int o = LPR / 2; o >= 1; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (q < bin_size && sub == 0) row_nz[rid] = cnt;
}

// bins 1..5: one workgroup per row (set_row_nz_bin_each_tb :399-472; LARGE = the try-in-LDS
// kernel with a fail list, set_row_nz_bin_each_tb_large :474-554).
template <int BS, int TMAX, bool LARGE>
__global__ __launch_bounds__(BS) void k_sym_tb(const int *__restrict__ arpt,
                                               const int *__restrict__ acol,
                                               const int *__restrict__ brpt,
                                               const int *__restrict__ bcol,
                                               const int *__restrict__ row_perm,
                                               const int *__restrict__ row_prod,
                                                  const int *__restrict__ row_maxb,
                                               int *__restrict__ row_nz, int bin_off, int bin_size,
                                               int bnnz, BinState *bs, int *__restrict__ fail_list)
{
    __shared__ __attribute__((aligned(16))) int tab[TMAX];
    __shared__ int2 s_ext[LARGE ? 1 : BS];
    __shared__ int s_nz;
    const int slot = xcd_row_slot(bin_size);
    if (slot < 0) return;
    const int rid = row_perm[bin_off + slot];
    const int np = row_prod[rid];
    int T = LARGE ? TMAX : pow2_ceil(np);
    if (T < 64) T = 64;
    if (T > TMAX) T = TMAX;
    const int mask = T - 1;
    {
        int4 *t4 = reinterpret_cast<int4 *>(tab);
        const int4 m1 = make_int4(-1, -1, -1, -1);
        for (int i = threadIdx.x; i < T / 4; i += BS) t4[i] = m1;
    }
    if (threadIdx.x == 0) s_nz = 0;
    __syncthreads();

    const int a_beg = arpt[rid], a_end = arpt[rid + 1];
    const int g = group_width(np, a_end - a_beg, BS, row_maxb[rid]);
    int cnt = 0;
    if (!LARGE) {
        walk_products<BS, false>(acol, (const real *)nullptr, brpt, bcol, (const real *)nullptr, bnnz,
                                 a_beg, a_end, g, s_ext, (real *)nullptr,
                                 [&](const IVec &k, const RVec &, int n, real) {
                                     int h[VW];
                                     ht_insert_vec(tab, mask, k, n, h, cnt);
                                 });
    } else {
        // try-in-LDS: plain walk with early exit once the table holds kSymLargeLimit keys
        const int ngroups = BS / g;
        const int gid = threadIdx.x / g, gl = threadIdx.x % g;
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
    __shared__ real vals[RPB * TROW];
    for (int i = threadIdx.x; i < RPB * TROW; i += BS) {
        keys[i] = -1;
        vals[i] = 0;
    }
    __syncthreads();
    const int lrow = threadIdx.x / LPR;
    const int sub = threadIdx.x % LPR;
    const int q = blockIdx.x * RPB + lrow;
    const bool active = q < bin_size;
    int rid = 0;
    int *kt = keys + lrow * TROW;
    real *vt = vals + lrow * TROW;
    if (active) {
        rid = row_perm[bin_off + q];
        const int e = arpt[rid + 1];
        for (int j = arpt[rid] + sub; j < e; j += LPR) {
            const int c = __builtin_nontemporal_load(acol + j);
            const real av = __builtin_nontemporal_load(aval + j);
            const int ke = brpt[c + 1];
            for (int k = brpt[c]; k < ke; k++) {
                int fresh;
                const int h = ht_find_or_insert(kt, TROW - 1, bcol[k], &fresh);
                unsafeAtomicAdd(vt + h, av * bval[k]);
            }
        }
    }
    __syncthreads();  // uniform: every thread reaches it
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
            cval[off + rank] = vt[s];
        }
    }
}

// In-LDS bitonic sort of P (power of two) ints, ascending.  Stages whose partner distance
// is < 128 keep every compare-exchange pair inside one wavefront's 128-element segment and
// run back to back with wave-level ordering only; only the wider stages need a workgroup
// barrier.
template <int BS>
__device__ __forceinline__ void bitonic_sort_lds(int *s, int P)
{
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    constexpr int NW = BS / 64;
    // compare-exchange of pair number t at partner distance j inside merge size k
    auto cex = [&](int t, int j, int k) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int p = i | j;
        const int a = s[i], b = s[p];
        const bool up = (i & k) == 0;
        if ((a > b) == up) { s[i] = b; s[p] = a; }
    };
    auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    // phase 1: every merge size up to 128 stays inside a 128-element segment
    const int kmax1 = P < 128 ? P : 128;
    for (int seg = wid; seg * 128 < P; seg += NW) {
        const int t = seg * 64 + lane;
        for (int k = 2; k <= kmax1; k <<= 1)
            for (int j = k >> 1; j >= 1; j >>= 1) {
                if (t < P / 2) cex(t, j, k);
                wave_sync();
            }
    }
    __syncthreads();
    // phase 2: wide stages with workgroup barriers, then the sub-segment tail of each merge
    for (int k = 256; k <= P; k <<= 1) {
        for (int j = k >> 1; j >= 128; j >>= 1) {
            for (int t = threadIdx.x; t < P / 2; t += BS) cex(t, j, k);
            __syncthreads();
        }
        for (int seg = wid; seg * 128 < P; seg += NW) {
            const int t = seg * 64 + lane;
            for (int j = 64; j >= 1; j >>= 1) {
                cex(t, j, k);
                wave_sync();
            }
        }
        __syncthreads();
    }
}

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
                                               int bin_size, int bnnz, int write_col)
{
    __shared__ __attribute__((aligned(16))) real vals[TMAX];
    __shared__ __attribute__((aligned(16))) int keys[TMAX];
    __shared__ __attribute__((aligned(16))) int srt[PMAX];
    __shared__ int2 s_ext[BS];
    __shared__ real s_av[BS];
    __shared__ int s_cnt;
    const int slot = xcd_row_slot(bin_size);
    if (slot < 0) return;
    const int rid = row_perm[bin_off + slot];
    const int off = crpt[rid];
    const int n = crpt[rid + 1] - off;
    int T = pow2_ceil(n + (n >> 1));
    if (T < 64) T = 64;
    if (T > TMAX) T = TMAX;
    const int mask = T - 1;
    for (int i = threadIdx.x; i < T; i += BS) {
        keys[i] = -1;
        vals[i] = 0;
    }
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();

    const int a_beg = arpt[rid], a_end = arpt[rid + 1];
    const int g = group_width(row_prod[rid], a_end - a_beg, BS, row_maxb[rid]);
    walk_products<BS, true>(acol, aval, brpt, bcol, bval, bnnz, a_beg, a_end, g, s_ext, s_av,
                            [&](const IVec &k, const RVec &v, int n, real sc) {
                                int h[VW], fresh = 0;
                                ht_insert_vec(keys, mask, k, n, h, fresh);
#pragma unroll
                                for (int i = 0; i < VW; i++)
                                    if (i < n) unsafeAtomicAdd(vals + h[i], sc * v.v[i]);
                            });
    __syncthreads();

    // compaction: ballot + popcount inside the wave, one LDS atomic per 64 slots
    const int lane = threadIdx.x & 63;
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
    const int P = pow2_ceil(n);
    for (int i = n + threadIdx.x; i < P; i += BS) srt[i] = 0x7fffffff;
    __syncthreads();
    // write_col bit 1: unsorted output requested (cuda-cpp template<bool sort>,
    // HashSpGEMM_volta.hpp:585-604): columns leave in compaction order
    if (P > 1 && !(write_col & 2)) bitonic_sort_lds<BS>(srt, P);

    for (int i = threadIdx.x; i < n; i += BS) {
        const int key = srt[i];
        int h = hash_slot(key, mask);
        while (keys[h] != key) h = (h + 1) & mask;
        if (write_col & 1) ccol[off + i] = key;
        cval[off + i] = vals[h];
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
                                                  int *__restrict__ row_span_num)
{
    __shared__ __attribute__((aligned(16))) unsigned int flag4[SPAN_MAX / 4 + 8];
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
    const int a_beg = arpt[rid], a_end = arpt[rid + 1];
    const int g = group_width(row_prod[rid], a_end - a_beg, BS, row_maxb[rid]);
    walk_products<BS, false>(acol, (const real *)nullptr, brpt, bcol, (const real *)nullptr, bnnz, a_beg,
                             a_end, g, s_ext, (real *)nullptr,
                             [&](const IVec &k, const RVec &, int n, real) {
#pragma unroll
                                 for (int i = 0; i < VW; i++)
                                     if (i < n) flag[k.v[i] - lo] = 1;
                             });
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
                                 [&](const IVec &k, const RVec &, int n, real) {
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
    __shared__ __attribute__((aligned(16))) real dense[SPAN_MAX + 4];
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
    for (int i = threadIdx.x; i < 4 * Q; i += BS) dense[i] = 0;
    __syncthreads();
    const int a_beg = arpt[rid], a_end = arpt[rid + 1];
    const int g = group_width(row_prod[rid], a_end - a_beg, BS, row_maxb[rid]);
    walk_products<BS, true>(acol, aval, brpt, bcol, bval, bnnz, a_beg, a_end, g, s_ext, s_av,
                            [&](const IVec &k, const RVec &v, int n, real sc) {
#pragma unroll
                                for (int i = 0; i < VW; i++)
                                    if (i < n) {
                                        const int idx = k.v[i] - lo;
                                        unsafeAtomicAdd(dense + __mul24(idx & 3, Q) + (idx >> 2), sc * v.v[i]);
                                    }
                            });
    __syncthreads();
    if (MODE == 2) {
        const int n = crpt[rid + 1] - off;
        for (int p = threadIdx.x; p < n; p += BS) {
            const int idx = ccol[off + p] - lo;
            cval[off + p] = dense[__mul24(idx & 3, Q) + (idx >> 2)];
        }
        return;
    }
    // ordered emission: wave w owns the column range [w*R, (w+1)*R)
    const unsigned int *bits = bm + bm_off[rid];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int R = ((span + NW * 64 - 1) / (NW * 64)) * 64;
    const int rb = w * R, re = rb + R < span ? rb + R : span;
    int cnt = 0;
    for (int base = rb; base < re; base += 64) {
        const int idx = base + lane;
        const bool occ = idx < re && ((bits[idx >> 5] >> (idx & 31)) & 1u);
        cnt += __popcll(__ballot(occ));
    }
    if (lane == 0) s_wcnt[w] = cnt;
    __syncthreads();
    int pos = off;
    for (int u = 0; u < w; u++) pos += s_wcnt[u];
    for (int base = rb; base < re; base += 64) {
        const int idx = base + lane;
        const bool occ = idx < re && ((bits[idx >> 5] >> (idx & 31)) & 1u);
        const unsigned long long m = __ballot(occ);
        if (occ) {
            const int p = pos + __popcll(m & ((1ull << lane) - 1ull));
            ccol[p] = lo + idx;
            cval[p] = dense[__mul24(idx & 3, Q) + (idx >> 2)];
        }
        pos += __popcll(m);
    }
}

// ===================================================================================
//  heavy numeric rows: column-tiled dense windows
// ===================================================================================
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it
// waits for the acknowledgement of every global store issued before it; the tiled kernel
// emits a tile with global stores nobody in the workgroup reads back, so waiting for them
// once per tile (a full HBM round trip) is pure stall.
// Inclusive prefix sum over the 64 lanes in registers (DPP row shifts + row broadcasts); the
// __shfl_up form goes through the LDS crossbar six times.
__device__ __forceinline__ int wave_incl_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return v;
}

__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Rows with more than 5461 non-zeros do not fit an LDS hash table, and on power-law inputs
// they carry most of the products (R-MAT-18: 59 K such rows, 2.7 G products).  Hashing them in
// global memory means two random HBM round trips per product.  Instead the row's column window
// is cut into tiles of W columns that DO fit LDS as a dense array.  Rows of B are sorted, so
// the part of B row k that falls into a tile is contiguous: every A entry keeps a cursor
// (position, end, next column, a value) in a per-workgroup global scratch slice, and for each
// tile every lane advances the cursors of its entries while the column stays inside the tile,
// accumulating into LDS.  The tile is then emitted in ascending order (byte flags + ballot /
// popcount), so the row leaves sorted without a sort.  One pass over the products, no global
// atomics.  Persistent workgroups pull rows from a queue.  Needs sorted rows of B (checked by
// the caller through the B-info pass: unsorted B falls back to the global hash table).
template <int BS, int W>
__global__ __launch_bounds__(BS) void k_num_tiled(const int *__restrict__ arpt, const int *__restrict__ acol,
                                                  const real *__restrict__ aval,
                                                  const int *__restrict__ brpt, const int *__restrict__ bcol,
                                                  const real *__restrict__ bval,
                                                  const int *__restrict__ crpt, int *__restrict__ ccol,
                                                  real *__restrict__ cval,
                                                  const int *__restrict__ row_perm, int bin_off, int count,
                                                  BinState *bs, const int *__restrict__ row_lo,
                                                  const int *__restrict__ row_span, int *__restrict__ slab,
                                                  long long stride_ints, int amax, int write_col,
                                                  int LONG_LEN, unsigned long long *prof, int dens)
{
    // prof (NSPARSE_TILED_PROF=1): thread 0 adds 100 MHz ticks per phase -- 0 cursor set-up,
    // 2 register-fed accumulation, 3 overflow paths, 4 emission, 5 tiles, 6 rows
    // (kept in registers, one atomic per counter when the workgroup retires)
    unsigned long long tk = prof ? wall_clock64() : 0;
    unsigned long long t_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto tick = [&](int phase) {
        if (prof) {
            const unsigned long long now = wall_clock64();
            t_acc[phase] += now - tk;
            tk = now;
        }
    };
    constexpr int NW = BS / 64;
    constexpr int LCAP = 1024;      // long B rows tracked per C row (the rest stay lane-serial)
    constexpr int EPT = 4;          // lane-serial cursors per thread kept in registers
    constexpr int KS = 4;           // sweep slots per wavefront kept in registers
    constexpr int VMAX = 32;        // sweep slots one B row may be dealt out to
    constexpr int INF = 0x7fffffff;
    constexpr int R = W / NW;       // columns of a tile emitted by one wavefront
    constexpr int IT = R / 64;
    static_assert(W % (NW * 64) == 0, "tile width must split evenly over the wavefronts");
    // LONG_LEN: a B row longer than this is swept by a whole wavefront
    __shared__ __attribute__((aligned(16))) real dense[W];
    __shared__ __attribute__((aligned(16))) unsigned int flag4[W / 4];
    __shared__ int4 l_meta[LCAP];   // sweep list: (chunk position, row end, stride, -)
    __shared__ real l_av[LCAP];
    __shared__ int s_row;
    __shared__ int s_nlong;
    __shared__ int s_wcnt[NW];
    unsigned char *flag = reinterpret_cast<unsigned char *>(flag4);
    int *st_cur = slab + (long long)blockIdx.x * stride_ints;
    int *st_end = st_cur + amax;
    int *st_next = st_end + amax;
    real *st_av = reinterpret_cast<real *>(st_next + amax);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // The window is clean on entry to every tile: cleared here once, and the emission resets
    // exactly the slots it finds occupied.
    for (int i = threadIdx.x; i < W; i += BS) dense[i] = 0;
    for (int i = threadIdx.x; i < W / 4; i += BS) flag4[i] = 0;
    while (true) {
        __syncthreads();
        if (threadIdx.x == 0) {
            s_row = atomicAdd(&bs->queue_head, 1);
            s_nlong = 0;
        }
        __syncthreads();
        const int q = s_row;
        if (q >= count) break;
        tick(1);  // queue + barriers
        const int rid = row_perm[bin_off + q];
        const int lo = row_lo[rid], span = row_span[rid];
        // dens > 0: rows thinner than one non-zero per `dens` columns, or wider than 32 tiles, are
        // left to k_num_ranked
        if (dens > 0 && ((long long)(crpt[rid + 1] - crpt[rid]) * dens < span || span > 32 * W)) continue;
        const int a_beg = arpt[rid], alen = arpt[rid + 1] - a_beg;
        // ---- cursor set-up ------------------------------------------------------------
        // Entry e of the A row is always handled by thread e % BS.  Its cursor into B row
        // acol[e] keeps the NEXT TWO (column, value) pairs in registers, so that a tile in
        // which the entry has at most one product is served without waiting for memory: the
        // refill issued when a pair is consumed is only needed a tile later.  The first EPT
        // entries of a thread live in registers; the rest (A rows beyond EPT * BS entries) in
        // the workgroup's global slice with a one-column look-ahead.  B rows longer than
        // LONG_LEN go to the LDS list and are swept by whole wavefronts.
        int e_cur[EPT], e_end[EPT], e_c0[EPT], e_c1[EPT];
        real e_v0[EPT], e_v1[EPT], e_av[EPT];
        // A long B row is swept in chunks of 64 consecutive entries.  A row that would put more
        // than one chunk into a tile is dealt out chunk by chunk to V sweep slots (slot v takes
        // chunks v, v + V, ...), which land on different wavefronts: on power-law inputs the
        // longest B rows carry most of the products of a C row.
        const int split = 64 * ((span + W - 1) / W);
        auto init_entry = [&](int e, int &cur, int &end, real &av) -> bool {
            const int c = acol[a_beg + e];
            cur = brpt[c];
            end = brpt[c + 1];
            av = aval[a_beg + e];
            const int len = end - cur;
            if (len <= LONG_LEN) return true;
            int V = (len + split - 1) / split;
            V = V < 1 ? 1 : (V > VMAX ? VMAX : V);
            const int li = atomicAdd(&s_nlong, V);
            const bool fits = li + V <= LCAP;
            for (int v = 0; v < V && li + v < LCAP; v++) {
                l_meta[li + v] = fits ? make_int4(cur + 64 * v, end, 64 * V, 0) : make_int4(0, 0, 64, 0);
                l_av[li + v] = av;
            }
            return !fits;  // true: lane-serial
        };
#pragma unroll
        for (int u = 0; u < EPT; u++) {
            const int e = threadIdx.x + u * BS;
            e_cur[u] = e_end[u] = 0;
            e_c0[u] = e_c1[u] = INF;
            e_v0[u] = e_v1[u] = e_av[u] = 0;
            if (e < alen && init_entry(e, e_cur[u], e_end[u], e_av[u])) {
                const int k = e_cur[u], end = e_end[u];
                if (k < end) {
                    e_c0[u] = bcol[k];
                    e_v0[u] = bval[k];
                }
                if (k + 1 < end) {
                    e_c1[u] = bcol[k + 1];
                    e_v1[u] = bval[k + 1];
                }
            }
        }
        for (int e = threadIdx.x + EPT * BS; e < alen; e += BS) {
            int cur, end;
            real av;
            const bool serial = init_entry(e, cur, end, av);
            st_cur[e] = cur;
            st_end[e] = end;
            st_next[e] = (serial && cur < end) ? bcol[cur] : INF;
            st_av[e] = av;
        }
        __syncthreads();
        const int nlong = s_nlong < LCAP ? s_nlong : LCAP;
        // sweep slots w, w + NW, ...: the first KS of a wavefront keep their current chunk (A)
        // and the next one (B) in registers, 64 (column, value) pairs each
        int pa_col[KS], pb_col[KS];
        real pa_val[KS], pb_val[KS];
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const int i = w + s * NW;
            pa_col[s] = pb_col[s] = INF;
            pa_val[s] = pb_val[s] = 0;
            if (i < nlong) {
                const int4 mt = l_meta[i];
                const int ka = mt.x + lane, kb = ka + mt.z;
                if (ka < mt.y) {
                    pa_col[s] = bcol[ka];
                    pa_val[s] = bval[ka];
                }
                if (kb < mt.y) {
                    pb_col[s] = bcol[kb];
                    pb_val[s] = bval[kb];
                }
            }
        }
        tick(0);
        if (prof) t_acc[6]++;
        int pos = crpt[rid];
        for (int t0 = 0; t0 < span; t0 += W) {
            const int tw = span - t0 < W ? span - t0 : W;  // columns in this tile
            const int c0 = lo + t0, tile_end = c0 + tw;
            auto acc = [&](int col, real x) {
                const int idx = col - c0;
                flag[idx] = 1;
                unsafeAtomicAdd(dense + idx, x);
            };
            // ---- register-fed pass ---------------------------------------------------------
            // Everything inside the tile that is already in registers is accumulated and its
            // refill issued; a second trip is needed only by cursors that used up their whole
            // look-ahead (64 in-tile entries of a long row, 2 of a short one), and then all of
            // them wait for their refills together.
            // A chunk serves every tile it overlaps (the range test has two sides) and is
            // replaced only when its last column lies below the end of the tile.
            bool more;
            bool fresh[KS];
#pragma unroll
            for (int s = 0; s < KS; s++) fresh[s] = true;
            do {
                more = false;
                if (prof) t_acc[10]++;
#pragma unroll
                for (int s = 0; s < KS; s++) {
                    if (!fresh[s]) continue;  // wave-uniform
                    if ((unsigned)(pa_col[s] - c0) < (unsigned)tw) acc(pa_col[s], l_av[w + s * NW] * pa_val[s]);
                    fresh[s] = __builtin_amdgcn_readlane(pa_col[s], 63) < tile_end;
                    if (fresh[s]) {  // chunk A used up: B moves in, the one after B is requested
                        const int i = w + s * NW;
                        int4 mt = l_meta[i];
                        mt.x += mt.z;
                        if (lane == 0) l_meta[i].x = mt.x;
                        pa_col[s] = pb_col[s];
                        pa_val[s] = pb_val[s];
                        const int k = mt.x + mt.z + lane;
                        const bool ok = k < mt.y;
                        pb_col[s] = ok ? bcol[k] : INF;
                        pb_val[s] = ok ? bval[k] : (real)0;
                        more = true;
                    }
                }
#pragma unroll
                for (int u = 0; u < EPT; u++) {
                    if (e_c0[u] < tile_end) {
                        acc(e_c0[u], e_av[u] * e_v0[u]);
                        const bool two = e_c1[u] < tile_end;
                        if (two) {
                            acc(e_c1[u], e_av[u] * e_v1[u]);
                            e_cur[u] += 2;
                            const int k = e_cur[u];
                            const bool ok = k < e_end[u];
                            e_c0[u] = ok ? bcol[k] : INF;
                            e_v0[u] = ok ? bval[k] : (real)0;
                            more = true;
                        } else {
                            e_cur[u] += 1;
                            e_c0[u] = e_c1[u];
                            e_v0[u] = e_v1[u];
                        }
                        const int k1 = e_cur[u] + 1;
                        const bool ok1 = k1 < e_end[u];
                        e_c1[u] = ok1 ? bcol[k1] : INF;
                        e_v1[u] = ok1 ? bval[k1] : (real)0;
                    }
                }
            } while (__any(more));
            if (prof) {
                __syncthreads();
                tick(2);
            }
            // ---- overflow paths: state in LDS / global memory --------------------------------
            // sweep slots beyond the register ones: same chunk walk, one round trip per chunk
            for (int i = w + KS * NW; i < nlong; i += NW) {
                int4 mt = l_meta[i];
                const real av = l_av[i];
                while (true) {
                    const int k = mt.x + lane;
                    const int col = k < mt.y ? bcol[k] : INF;
                    const real bv = k < mt.y ? bval[k] : (real)0;
                    if ((unsigned)(col - c0) < (unsigned)tw) acc(col, av * bv);
                    if (__builtin_amdgcn_readlane(col, 63) >= tile_end) break;
                    mt.x += mt.z;
                }
                if (lane == 0) l_meta[i].x = mt.x;
            }
            // A entries beyond EPT * BS: one product per memory round trip
            for (int e = threadIdx.x + EPT * BS; e < alen; e += BS) {
                int col = st_next[e];
                if (col < tile_end) {
                    int cur = st_cur[e];
                    const int end = st_end[e];
                    const real av = st_av[e];
                    do {
                        const real bv = bval[cur];
                        cur++;
                        const int ncol = cur < end ? bcol[cur] : INF;  // issued with bv
                        acc(col, av * bv);
                        col = ncol;
                    } while (col < tile_end);
                    st_cur[e] = cur;
                    st_next[e] = col;
                }
            }
            lds_barrier();
            tick(3);
            // ---- ordered emission: wavefront w owns columns [w*R, (w+1)*R) of the tile -----
            // all IT flag reads are issued together; columns past tw are clean, hence empty
            const int r0 = w * R;
            unsigned long long msk[IT];
            int cnt = 0;
#pragma unroll
            for (int j = 0; j < IT; j++) {
                msk[j] = __ballot(flag[r0 + j * 64 + lane] != 0);
                cnt += __popcll(msk[j]);
            }
            if (lane == 0) s_wcnt[w] = cnt;
            tick(7);
            lds_barrier();
            tick(8);
            int wpos = pos, total = 0;
            for (int u = 0; u < NW; u++) {
                const int c = s_wcnt[u];
                if (u < w) wpos += c;
                total += c;
            }
#pragma unroll
            for (int j = 0; j < IT; j++) {
                const unsigned long long m = msk[j];
                if ((m >> lane) & 1ull) {
                    const int idx = r0 + j * 64 + lane;
                    const int p = wpos + __popcll(m & ((1ull << lane) - 1ull));
                    if (write_col & 1) ccol[p] = c0 + idx;
                    cval[p] = dense[idx];
                    dense[idx] = 0;  // leave the window clean for the next tile
                    flag[idx] = 0;
                }
                wpos += __popcll(m);
            }
            pos += total;
            tick(9);
            lds_barrier();
            tick(4);
            if (prof) t_acc[5]++;
        }
    }
    if (prof && threadIdx.x == 0)
        for (int i = 0; i < 12; i++) atomicAdd(prof + i, t_acc[i]);
}

// ===================================================================================
//  heavy numeric rows, sparse flavour: bitmap-ranked accumulator
// ===================================================================================
// A heavy row whose columns are spread thinly over a wide window (R-MAT scale 22: 18 K non-zeros
// over 4 M columns) would need hundreds of almost empty dense tiles.  Here LDS holds 8 bytes per
// NON-ZERO instead of 9 bytes per column: a tile is a bitmap over W columns plus a value array
// of CAP entries addressed by rank.
//   pass 1  walk the products of the tile, set the bit of every column           (ds_or)
//   scan    per-word exclusive prefix of the popcounts; if the tile holds more than CAP
//           columns it is cut at the word where the prefix crosses CAP
//   pass 2  walk again, accumulate a*b at  prefix[word] + popcount(bits below)    (ds_add)
//   emit    values are already in ascending column order and contiguous; the columns are
//           read off the bitmap
// Cursors advance only in pass 2, so the cut costs nothing but the re-walk of the columns
// beyond it.  Same cursor scheme as k_num_tiled: lane-serial entries (4 look-ahead loads per
// step), long B rows dealt out in 64-entry chunks to wavefront sweep slots.
template <int BS, int W, int CAP>
__global__ __launch_bounds__(BS) void k_num_ranked(const int *__restrict__ arpt, const int *__restrict__ acol,
                                                   const real *__restrict__ aval,
                                                   const int *__restrict__ brpt, const int *__restrict__ bcol,
                                                   const real *__restrict__ bval,
                                                   const int *__restrict__ crpt, int *__restrict__ ccol,
                                                   real *__restrict__ cval,
                                                   const int *__restrict__ row_perm, int bin_off, int count,
                                                   BinState *bs, const int *__restrict__ row_lo,
                                                   const int *__restrict__ row_span, int *__restrict__ slab,
                                                   long long stride_ints, int amax, int write_col,
                                                   int LONG_LEN, int dens, int tiled_w,
                                                   unsigned long long *prof)
{
    // prof (NSPARSE_TILED_PROF=1), 100 MHz ticks of thread 0: 0 set-up, 1 pass 1, 2 scan, 3 pass 2,
    // 4 emission; 5 tiles, 6 rows
    unsigned long long tk = prof ? wall_clock64() : 0;
    unsigned long long t_acc[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto tick = [&](int phase) {
        if (prof) {
            const unsigned long long now = wall_clock64();
            t_acc[phase] += now - tk;
            tk = now;
        }
    };
    constexpr int NW = BS / 64;
    constexpr int LCAP = BS, EPT = 4, VMAX = 32, LA = 4;
    constexpr int INF = 0x7fffffff;
    constexpr int NWORD = W / 32;
    static_assert(NWORD == 8 * BS, "eight bitmap words per thread");
    static_assert(CAP <= 65535, "ranks are kept in 16 bits");
    __shared__ __attribute__((aligned(16))) unsigned int bits[NWORD];
    __shared__ __attribute__((aligned(16))) unsigned short pref[NWORD];
    __shared__ __attribute__((aligned(16))) real vals[CAP];
    __shared__ int4 l_meta[LCAP];
    __shared__ real l_av[LCAP];
    __shared__ int s_row, s_nlong, s_cut, s_ntile, s_total;
    __shared__ int s_wsum[8 * NW];
    int *st_cur = slab + (long long)blockIdx.x * stride_ints;
    int *st_end = st_cur + amax;
    int *st_next = st_end + amax;
    real *st_av = reinterpret_cast<real *>(st_next + amax);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < NWORD; i += BS) bits[i] = 0;
    for (int i = threadIdx.x; i < CAP; i += BS) vals[i] = 0;
    while (true) {
        __syncthreads();
        if (threadIdx.x == 0) {
            s_row = atomicAdd(&bs->queue_head2, 1);
            s_nlong = 0;
        }
        __syncthreads();
        const int q = s_row;
        if (q >= count) break;
        const int rid = row_perm[bin_off + q];
        const int lo = row_lo[rid], span = row_span[rid];
        int pos = crpt[rid];
        // dens > 0: only rows thinner than one non-zero per `dens` columns or wider than 32 dense
        // tiles (the rest belong to k_num_tiled); dens <= 0: every row
        if (dens > 0 && (long long)(crpt[rid + 1] - pos) * dens >= span && span <= 32 * tiled_w) continue;
        const int a_beg = arpt[rid], alen = arpt[rid + 1] - a_beg;
        const int split = 64 * ((span + W - 1) / W + (crpt[rid + 1] - pos) / CAP + 1);
        auto init_entry = [&](int e, int &cur, int &end, real &av) -> bool {
            const int c = acol[a_beg + e];
            cur = brpt[c];
            end = brpt[c + 1];
            av = aval[a_beg + e];
            const int len = end - cur;
            if (len <= LONG_LEN) return true;
            int V = (len + split - 1) / split;
            V = V < 1 ? 1 : (V > VMAX ? VMAX : V);
            const int li = atomicAdd(&s_nlong, V);
            const bool fits = li + V <= LCAP;
            for (int v = 0; v < V && li + v < LCAP; v++) {
                l_meta[li + v] = fits ? make_int4(cur + 64 * v, end, 64 * V, 0) : make_int4(0, 0, 64, 0);
                l_av[li + v] = av;
            }
            return !fits;  // true: lane-serial
        };
        int e_cur[EPT], e_end[EPT], e_nc[EPT];
        real e_av[EPT];
#pragma unroll
        for (int u = 0; u < EPT; u++) {
            const int e = threadIdx.x + u * BS;
            e_cur[u] = e_end[u] = 0;
            e_nc[u] = INF;
            e_av[u] = 0;
            if (e < alen && init_entry(e, e_cur[u], e_end[u], e_av[u]) && e_cur[u] < e_end[u])
                e_nc[u] = bcol[e_cur[u]];
        }
        for (int e = threadIdx.x + EPT * BS; e < alen; e += BS) {
            int cur, end;
            real av;
            const bool serial = init_entry(e, cur, end, av);
            st_cur[e] = cur;
            st_end[e] = end;
            st_next[e] = (serial && cur < end) ? bcol[cur] : INF;
            st_av[e] = av;
        }
        __syncthreads();
        const int nlong = s_nlong < LCAP ? s_nlong : LCAP;
        const int row_end = lo + span;
        int t_lo = lo;
        tick(0);
        if (prof) {
            t_acc[6]++;
            t_acc[10] += nlong;
            t_acc[11] += alen;
            t_acc[12] += s_nlong > LCAP;
        }
        while (t_lo < row_end) {
            // Pass 1 runs to t_max and everything beyond the cut is walked again by the next tile,
            // so t_max aims at ~7/8 of CAP columns at the density of what is left of the row.
            int t_max;
            {
                const long long rem_nnz = crpt[rid + 1] - pos, rem_span = row_end - t_lo;
                long long wd_est = rem_nnz > 0 ? (long long)(CAP - CAP / 8) * rem_span / rem_nnz : rem_span;
                wd_est = (wd_est + 31) & ~31LL;
                if (wd_est > W) wd_est = W;
                if (wd_est < 1024) wd_est = 1024;
                t_max = rem_span <= wd_est ? row_end : t_lo + (int)wd_est;
            }
            // One walk of everything inside [t_lo, t_hi).  PASS2 = false: mark columns.
            // PASS2 = true: accumulate by rank and commit the cursors.
            auto touch = [&](auto pass2, int col, real x) {
                const unsigned int idx = (unsigned int)(col - t_lo);
                if (!decltype(pass2)::value) {
                    atomicOr(&bits[idx >> 5], 1u << (idx & 31));
                } else {
                    const unsigned int below = bits[idx >> 5] & ((1u << (idx & 31)) - 1u);
                    unsafeAtomicAdd(vals + (int)pref[idx >> 5] + __popc(below), x);
                }
            };
            // lane-serial entries: LA consecutive (column, value) pairs per round trip; the first
            // batch of all register entries is requested before any of it is used
            auto load_batch = [&](auto pass2, int k, int end, int c0, int(&c)[LA], real(&v)[LA]) {
                c[0] = c0;
#pragma unroll
                for (int j = 1; j < LA; j++) c[j] = k + j < end ? bcol[k + j] : INF;
                if (decltype(pass2)::value) {
#pragma unroll
                    for (int j = 0; j < LA; j++) v[j] = k + j < end ? bval[k + j] : (real)0;
                }
            };
            // uses the leading pairs that lie inside the tile; returns how many, and the column after them
            auto consume = [&](auto pass2, int t_hi, const int(&c)[LA], const real(&v)[LA], real av, int &next) -> int {
                int n = 0;
                next = INF;
#pragma unroll
                for (int j = 0; j < LA; j++) {
                    if (n == j) {
                        if (c[j] < t_hi) {
                            touch(pass2, c[j], decltype(pass2)::value ? av * v[j] : (real)0);
                            n = j + 1;
                        } else {
                            next = c[j];
                        }
                    }
                }
                return n;
            };
            // after a batch that was used up completely: keep going, one round trip per batch
            auto walk_rest = [&](auto pass2, int t_hi, int &k, int end, int &col, real av) {
                col = k < end ? bcol[k] : INF;
                while (col < t_hi) {
                    int c[LA];
                    real v[LA];
                    load_batch(pass2, k, end, col, c, v);
                    const int n = consume(pass2, t_hi, c, v, av, col);
                    k += n;
                    if (n < LA) return;
                    col = k < end ? bcol[k] : INF;
                }
            };
            auto walk = [&](auto pass2, int t_hi) {
                constexpr bool P2 = decltype(pass2)::value;
                // sweep slots: 64-entry chunks, two-sided range test (a chunk may straddle tiles);
                // the current chunks of SB slots are requested together, and the first such batch
                // together with the first pair of register entries: one round trip for both
                constexpr int SB = 4;
                auto sweep_load = [&](int i0, int4(&mt)[SB], int(&col)[SB], real(&bv)[SB]) {
#pragma unroll
                    for (int j = 0; j < SB; j++) {
                        const int i = i0 + j * NW;
                        mt[j] = i < nlong ? l_meta[i] : make_int4(0, 0, 64, 0);
                        const int kk = mt[j].x + lane;
                        col[j] = kk < mt[j].y ? bcol[kk] : INF;
                        bv[j] = 0;
                        if (P2) bv[j] = kk < mt[j].y ? bval[kk] : (real)0;
                    }
                };
                auto sweep_use = [&](int i0, const int4(&mt)[SB], const int(&col)[SB], const real(&bv)[SB]) {
#pragma unroll
                    for (int j = 0; j < SB; j++) {
                        const int i = i0 + j * NW;
                        if (i >= nlong) continue;
                        const real av = l_av[i];
                        int k = mt[j].x, c = col[j];
                        real x = bv[j];
                        while (true) {
                            if (c >= t_lo && c < t_hi) touch(pass2, c, av * x);
                            if (__builtin_amdgcn_readlane(c, 63) >= t_hi) break;
                            k += mt[j].z;
                            const int kk = k + lane;
                            c = kk < mt[j].y ? bcol[kk] : INF;
                            if (P2) x = kk < mt[j].y ? bval[kk] : (real)0;
                        }
                        if (P2 && lane == 0) l_meta[i].x = k;
                    }
                };
                auto entries_load = [&](int u0, int(&c)[2][LA], real(&v)[2][LA]) {
#pragma unroll
                    for (int d = 0; d < 2; d++)
                        if (e_nc[u0 + d] < t_hi) load_batch(pass2, e_cur[u0 + d], e_end[u0 + d], e_nc[u0 + d], c[d], v[d]);
                };
                auto entries_use = [&](int u0, const int(&c)[2][LA], const real(&v)[2][LA]) {
#pragma unroll
                    for (int d = 0; d < 2; d++) {
                        const int u = u0 + d;
                        if (e_nc[u] < t_hi) {
                            int col;
                            const int n = consume(pass2, t_hi, c[d], v[d], e_av[u], col);
                            int k = e_cur[u] + n;
                            if (n == LA) walk_rest(pass2, t_hi, k, e_end[u], col, e_av[u]);
                            if (P2) {
                                e_cur[u] = k;
                                e_nc[u] = col;
                            }
                        }
                    }
                };
                {
                    int4 mt[SB];
                    int col[SB];
                    real bv[SB];
                    int c[2][LA];
                    real v[2][LA];
                    if (w < nlong) sweep_load(w, mt, col, bv);
                    entries_load(0, c, v);
                    if (w < nlong) sweep_use(w, mt, col, bv);
                    entries_use(0, c, v);
                    static_assert(EPT == 4, "two pairs of register entries");
                    entries_load(2, c, v);
                    entries_use(2, c, v);
                    for (int i0 = w + SB * NW; i0 < nlong; i0 += SB * NW) {
                        sweep_load(i0, mt, col, bv);
                        sweep_use(i0, mt, col, bv);
                    }
                }
                for (int e = threadIdx.x + EPT * BS; e < alen; e += BS) {
                    int col = st_next[e];
                    if (col < t_hi) {
                        const int end = st_end[e];
                        const real av = st_av[e];
                        int c[LA];
                        real v[LA];
                        int k = st_cur[e];
                        load_batch(pass2, k, end, col, c, v);
                        const int n = consume(pass2, t_hi, c, v, av, col);
                        k += n;
                        if (n == LA) walk_rest(pass2, t_hi, k, end, col, av);
                        if (P2) {
                            st_cur[e] = k;
                            st_next[e] = col;
                        }
                    }
                }
            };
            walk(std::false_type{}, t_max);
            lds_barrier();
            tick(t_lo == lo ? 7 : 1);
            // ---- scan: thread t owns words t, t + BS, ..., t + 7 BS ----------------------------
            // (strided, so that the dense low-column stretch of a power-law row is shared by many
            // threads when the columns are written out)
            unsigned int wd[8];
            int pc[8];  // becomes the exclusive prefix of the word
#pragma unroll
            for (int j = 0; j < 8; j++) {
                wd[j] = bits[threadIdx.x + j * BS];
                const int c = __popc(wd[j]);
                const int incl = wave_incl_scan(c);
                pc[j] = incl - c;
                if (lane == 63) s_wsum[j * NW + w] = incl;
            }
            if (threadIdx.x == 0) s_cut = t_max;
            lds_barrier();
            if (w == 0) {  // 8 * NW partial sums in (segment, wavefront) order -> exclusive offsets
                static_assert(8 * NW <= 128, "two partial sums per lane");
                const int i0 = 2 * lane, i1 = 2 * lane + 1;
                const int a0 = i0 < 8 * NW ? s_wsum[i0] : 0, a1 = i1 < 8 * NW ? s_wsum[i1] : 0;
                const int incl = wave_incl_scan(a0 + a1);
                if (i0 < 8 * NW) s_wsum[i0] = incl - a0 - a1;
                if (i1 < 8 * NW) s_wsum[i1] = incl - a1;
                if (lane == 63) s_total = incl;
            }
            lds_barrier();
            const int total = s_total;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                pc[j] += s_wsum[j * NW + w];
                pref[threadIdx.x + j * BS] = (unsigned short)pc[j];
            }
            if (total > CAP) {  // cut at the word where the running count would pass CAP
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int p1 = pc[j] + __popc(wd[j]);
                    if (pc[j] <= CAP && p1 > CAP) {
                        s_cut = t_lo + 32 * (threadIdx.x + j * BS);
                        s_ntile = pc[j];
                    }
                }
            } else if (threadIdx.x == 0) {
                s_ntile = total;
            }
            lds_barrier();
            const int t_hi = s_cut, ntile = s_ntile;
            tick(2);
            if (prof && t_hi != t_max) t_acc[9]++;
            walk(std::true_type{}, t_hi);
            lds_barrier();
            tick(t_lo == lo ? 8 : 3);
            // ---- emission (words and prefixes are read back: not kept live across pass 2) ------
            if (write_col & 1) {
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    unsigned int m = bits[threadIdx.x + j * BS];
                    const int cbase = t_lo + 32 * (threadIdx.x + j * BS);
                    int p = pos + (int)pref[threadIdx.x + j * BS];
                    if (cbase < t_hi) {
                        while (m) {
                            ccol[p++] = cbase + __builtin_ctz(m);
                            m &= m - 1;
                        }
                    }
                }
            }
            for (int r = threadIdx.x; r < ntile; r += BS) {
                cval[pos + r] = vals[r];
                vals[r] = 0;
            }
#pragma unroll
            for (int j = 0; j < 8; j++) bits[threadIdx.x + j * BS] = 0;
            pos += ntile;
            t_lo = t_hi;
            lds_barrier();
            tick(4);
            if (prof) t_acc[5]++;
        }
    }
    if (prof && threadIdx.x == 0)
        for (int i = 0; i < 13; i++) atomicAdd(prof + 16 + i, t_acc[i]);
}

// ===================================================================================
//  host orchestration
// ===================================================================================

static void *scan_exclusive(const int *in, int *out, int n, hipStream_t st)
{
    size_t tmp_bytes = 0;
    NSP_CHECK(rocprim::exclusive_scan(nullptr, tmp_bytes, in, out, 0, (size_t)n, rocprim::plus<int>(), st));
    void *tmp = dev_alloc(tmp_bytes ? tmp_bytes : 1);
    NSP_CHECK(rocprim::exclusive_scan(tmp, tmp_bytes, in, out, 0, (size_t)n, rocprim::plus<int>(), st));
    // stream-ordered: the block returns to the cache only at the end of the call (dev_free
    // does not touch the device), and every later use of it is ordered after this scan
    return tmp;
}

static inline int pick_w(long long nnz, int M)
{
    const long long avg = M > 0 ? (nnz + M - 1) / M : 1;
    int w = 1;
    while (w < avg && w < 64) w <<= 1;
    return w;
}


static void launch_row_products(const sfCSR *a, const sfCSR *b, const BInfo *binfo,
                                int *row_prod, int *row_lo, int *row_span, int *bm_words,
                                int bm_span_max, const Thr &thr, BinState *d_bs, long long *partial,
                                int *row_span_num, int *row_nz, int *row_maxb, int *long_list,
                                int *long_cnt, hipStream_t st)
{
    const int M = a->M;
    const int w = pick_w(a->nnz, M);
    int grid = ceil_div((long long)M * w, 256);
    if (grid > kSetupMaxGrid - 256) grid = kSetupMaxGrid - 256;
    const int *no_todo = nullptr;
    // nnz_max in (0, threshold] => the host knows no row is long: no deferral, no second launch.
    // (A wrong nnz_max is harmless: without a list every row is simply walked in place.)
    if (a->nnz_max > 0 && a->nnz_max <= kLongFactor * w) long_list = nullptr;
#define NSP_RP(W)                                                                              \
    case W:                                                                                    \
        hipLaunchKernelGGL(k_row_products<W>, dim3(grid), dim3(256), 0, st, a->d_rpt, a->d_col, \
                           binfo, M, row_prod, row_lo, row_span, bm_words, bm_span_max, thr,   \
                           partial, row_span_num, row_nz, row_maxb, long_list, long_cnt,       \
                           kLongFactor * W, no_todo);                                          \
        break;
    switch (w) {
        NSP_RP(1) NSP_RP(2) NSP_RP(4) NSP_RP(8) NSP_RP(16) NSP_RP(32) NSP_RP(64)
    }
#undef NSP_RP
    if (long_list) {
        // the deferred long rows, 64 lanes each; their block partials follow the bulk pass's
        hipLaunchKernelGGL(k_row_products<64>, dim3(256), dim3(256), 0, st, a->d_rpt, a->d_col, binfo, M,
                           row_prod, row_lo, row_span, bm_words, bm_span_max, thr,
                           partial + (long long)grid * kPartialStride, row_span_num, row_nz, row_maxb,
                           (int *)nullptr, long_cnt, 0, (const int *)long_list);
        grid += 256;
    }
    hipLaunchKernelGGL(k_reduce_partials, dim3(grid < 32 ? grid : 32), dim3(256), 0, st, partial, grid, d_bs);
    NSP_LAUNCH_CHECK();
}

struct Timer {
    Context &cx;
    explicit Timer(Context &c) : cx(c) {}
    void mark(int i, hipStream_t st) { NSP_CHECK(hipEventRecord(cx.ev_t[i], st)); }
    float ms(int i, int j)
    {
        float v = 0;
        NSP_CHECK(hipEventSynchronize(cx.ev_t[j]));
        NSP_CHECK(hipEventElapsedTime(&v, cx.ev_t[i], cx.ev_t[j]));
        return v;
    }
};

// streams: bin b runs on cx.stream[b]; stream[0] is the main line.  In profiling mode
// everything is serialised on stream[0] and bracketed by events.
struct BinLauncher {
    Context *cx;
    hipEvent_t *ev;  // 2 * NB events: begin/end of every bin
    bool used[NB] = {};
    bool serial;     // profiling mode: one stream, bins back to back
    int main_bin;    // the bin with the most rows runs on the main stream itself (no fork/join)
    BinLauncher(Context &c, int phase, const int *hist = nullptr)
        : cx(&c), ev(c.ev_bin + phase * 2 * NB), serial(c.profiling), main_bin(-1)
    {
        if (hist) {
            int best = 0;
            for (int b = 0; b < NB; b++)
                if (hist[b] > best) { best = hist[b]; main_bin = b; }
        }
    }
    hipStream_t stream_of(int b) const { return (serial || b == main_bin) ? cx->stream[0] : cx->stream[b]; }
    void fork()
    {
        if (!serial) NSP_CHECK(hipEventRecord(cx->ev_fork, cx->stream[0]));
    }
    // The begin/end events sit on the stream the bin's kernels are launched on, so their
    // difference is the duration of those kernels whether or not other bins overlap.
    hipStream_t begin(int b)
    {
        hipStream_t st = stream_of(b);
        if (st != cx->stream[0] && !used[b]) NSP_CHECK(hipStreamWaitEvent(st, cx->ev_fork, 0));
        used[b] = true;
        NSP_CHECK(hipEventRecord(ev[2 * b], st));
        return st;
    }
    void end(int b) { NSP_CHECK(hipEventRecord(ev[2 * b + 1], stream_of(b))); }
    void join()
    {
        for (int b = 0; b < NB; b++) {
            if (!used[b] || stream_of(b) == cx->stream[0]) continue;
            NSP_CHECK(hipEventRecord(cx->ev_join[b], cx->stream[b]));
            NSP_CHECK(hipStreamWaitEvent(cx->stream[0], cx->ev_join[b], 0));
        }
    }
    void collect(float *out)  // call after the device is idle
    {
        for (int b = 0; b < NB; b++) {
            out[b] = 0;
            if (used[b]) {
                // the flag poll proves the GPU is done, not that the runtime has retired the event
                NSP_CHECK(hipEventSynchronize(ev[2 * b + 1]));
                NSP_CHECK(hipEventElapsedTime(&out[b], ev[2 * b], ev[2 * b + 1]));
            }
        }
    }
};

static int global_slab_groups(long long slice_elems, size_t bytes_per_elem, int rows)
{
    // persistent workgroups for the overflow path: bounded by rows, by 2 per CU, and by a
    // slab budget of 16 GiB (HBM3E is 288 GB; the budget only matters for multi-million
    // column matrices).
    const long long budget = 16LL << 30;
    long long g = budget / (slice_elems * (long long)bytes_per_elem);
    if (g > 512) g = 512;
    if (g > rows) g = rows;
    if (g < 1) g = 1;
    return (int)g;
}

static BinLauncher symbolic_phase(const sfCSR *a, const sfCSR *b, int *row_prod, const int *row_maxb,
                                  const int *row_lo,
                                  const int *row_span, int *row_nz, int *row_perm, const int *hist,
                                  int max_prod, BinState *d_bs, Context &cx, float *ms_bin,
                                  int *fail_rows, const int *bm_off, unsigned int *bm,
                                  int *row_span_num)
{
    BinLauncher L(cx, 0, hist);
    int off[NB + 1];
    off[0] = 0;
    for (int q = 0; q < NB; q++) off[q + 1] = off[q] + hist[q];
    const int *arpt = a->d_rpt, *acol = a->d_col, *brpt = b->d_rpt, *bcol = b->d_col;
    *fail_rows = 0;
    int *fail_list = nullptr;
    L.fork();
#define NSP_SYM_TB(BIN, BS, TMAX)                                                              \
    if (hist[BIN] > 0) {                                                                       \
        hipStream_t st = L.begin(BIN);                                                         \
        hipLaunchKernelGGL((k_sym_tb<BS, TMAX, false>), dim3(8 * ceil_div(hist[BIN], 8)), dim3(BS), 0, \
                           st, arpt, acol, brpt, bcol, row_perm, row_prod, row_maxb, row_nz, off[BIN],   \
                           hist[BIN], b->nnz, d_bs,                                            \
                           (int *)nullptr);                                                    \
        NSP_LAUNCH_CHECK();                                                                    \
        L.end(BIN);                                                                            \
    }
#define NSP_SYM_DENSE(BIN, BS, SPAN)                                                            \
    if (hist[BIN] > 0) {                                                                       \
        hipStream_t st = L.begin(BIN);                                                         \
        hipLaunchKernelGGL((k_sym_dense<BS, SPAN>), dim3(8 * ceil_div(hist[BIN], 8)), dim3(BS), 0, st, \
                           arpt, acol, brpt, bcol, row_perm, row_prod, row_maxb, row_lo, row_span, row_nz, \
                           off[BIN], hist[BIN], b->nnz, bm_off, bm, row_span_num);             \
        NSP_LAUNCH_CHECK();                                                                    \
        L.end(BIN);                                                                            \
    }
#define NSP_SYM_BITS(BIN, BS, WORDS)                                                           \
    if (hist[BIN] > 0) {                                                                       \
        hipStream_t st = L.begin(BIN);                                                         \
        hipLaunchKernelGGL((k_sym_bits<BS, WORDS>), dim3(8 * ceil_div(hist[BIN], 8)), dim3(BS), 0, st, \
                           arpt, acol, brpt, bcol, row_perm, row_prod, row_maxb, row_lo, row_span, row_nz, \
                           off[BIN], hist[BIN], b->nnz);                                       \
        NSP_LAUNCH_CHECK();                                                                    \
        L.end(BIN);                                                                            \
    }
    NSP_SYM_BITS(10, 1024, 32768)
    NSP_SYM_BITS(9, 512, 8192)
#undef NSP_SYM_BITS
    static const int tune_d6 = getenv("NSPARSE_SYMD6_BS") ? atoi(getenv("NSPARSE_SYMD6_BS")) : 128;
    NSP_SYM_DENSE(8, 1024, 65536)
    NSP_SYM_DENSE(7, 512, 16384)
    if (tune_d6 == 128) { NSP_SYM_DENSE(6, 128, 4096) } else if (tune_d6 == 512) { NSP_SYM_DENSE(6, 512, 4096) } else { NSP_SYM_DENSE(6, 256, 4096) }
#undef NSP_SYM_DENSE
    static const int tune_s3 = getenv("NSPARSE_SYM3_BS") ? atoi(getenv("NSPARSE_SYM3_BS")) : 512;
    static const int tune_s2 = getenv("NSPARSE_SYM2_BS") ? atoi(getenv("NSPARSE_SYM2_BS")) : 128;
    NSP_SYM_TB(4, 1024, 32768)
    if (tune_s3 == 512) { NSP_SYM_TB(3, 512, 8192) } else if (tune_s3 == 1024) { NSP_SYM_TB(3, 1024, 8192) } else { NSP_SYM_TB(3, 256, 8192) }
    if (tune_s2 == 256) { NSP_SYM_TB(2, 256, 2048) } else { NSP_SYM_TB(2, 128, 2048) }
    NSP_SYM_TB(1, 64, 512)
#undef NSP_SYM_TB
    if (hist[0] > 0) {
        hipStream_t st = L.begin(0);
        constexpr int BS = 256, LPR = 4;
        hipLaunchKernelGGL((k_sym_small<BS, LPR, 64>), dim3(ceil_div(hist[0], BS / LPR)), dim3(BS), 0,
                           st, arpt, acol, brpt, bcol, row_perm, row_nz, off[0], hist[0]);
        NSP_LAUNCH_CHECK();
        L.end(0);
    }
    // the overflow bin needs a host round trip (fail count), so it is issued last: by then
    // every other bin is already queued on its own stream.
    if (hist[5] > 0) {
        hipStream_t st = L.begin(5);
        fail_list = (int *)dev_alloc(sizeof(int) * (size_t)hist[5]);
        hipLaunchKernelGGL((k_sym_tb<1024, kSymLargeT, true>), dim3(8 * ceil_div(hist[5], 8)), dim3(1024), 0, st, arpt,
                           acol, brpt, bcol, row_perm, row_prod, row_maxb, row_nz, off[5], hist[5], b->nnz, d_bs, fail_list);
        NSP_LAUNCH_CHECK();
        NSP_CHECK(hipMemcpyAsync(cx.h_pinned + 128, &d_bs->fail_count, sizeof(int), hipMemcpyDeviceToHost, st));
        NSP_CHECK(hipStreamSynchronize(st));
        const int fails = cx.h_pinned[128];
        *fail_rows = fails;
        if (fails > 0) {
            long long bound = std::min<long long>(max_prod, b->N);
            long long slice = 64;
            while (slice < 2 * bound) slice <<= 1;
            const int groups = global_slab_groups(slice, sizeof(int), fails);
            int *slab = (int *)dev_alloc(sizeof(int) * (size_t)slice * groups);
            hipLaunchKernelGGL((k_sym_global<512>), dim3(groups), dim3(512), 0, st, arpt, acol, brpt,
                               bcol, fail_list, fails, row_prod, row_maxb, row_nz, b->N, d_bs, slab, slice);
            NSP_LAUNCH_CHECK();
            NSP_CHECK(hipStreamSynchronize(st));
            dev_free(slab);
        }
        L.end(5);
    }
    L.join();
    if (fail_list) dev_free(fail_list);  // its kernels have completed (bin 5 synchronises)
    return L;
}

static BinLauncher numeric_phase(const sfCSR *a, const sfCSR *b, sfCSR *c, const int *row_prod,
                                 const int *row_maxb,
                                 const int *row_lo, const int *row_span, const int *row_perm,
                                 const int *hist, int max_nz, BinState *d_bs, Context &cx,
                                 float *ms_bin, int write_col, const int *bm_off,
                                 const unsigned int *bm, int max_alen, bool b_sorted)
{
    BinLauncher L(cx, 1, hist);
    int off[NB + 1];
    off[0] = 0;
    for (int q = 0; q < NB; q++) off[q + 1] = off[q] + hist[q];
    const int *arpt = a->d_rpt, *acol = a->d_col, *brpt = b->d_rpt, *bcol = b->d_col;
    const real *aval = a->d_val, *bval = b->d_val;
    L.fork();
#define NSP_NUM_TB(BIN, BS, TMAX, PMAX)                                                        \
    if (hist[BIN] > 0) {                                                                       \
        hipStream_t st = L.begin(BIN);                                                         \
        hipLaunchKernelGGL((k_num_tb<BS, TMAX, PMAX>), dim3(8 * ceil_div(hist[BIN], 8)), dim3(BS), 0, st, arpt,  \
                           acol, aval, brpt, bcol, bval, c->d_rpt, c->d_col, c->d_val, row_perm, \
                           row_prod, row_maxb, off[BIN], hist[BIN], b->nnz, write_col);                                     \
        NSP_LAUNCH_CHECK();                                                                    \
        L.end(BIN);                                                                            \
    }
#define NSP_NUM_DENSE(BIN, BS, SPAN)                                                            \
    if (hist[BIN] > 0) {                                                                       \
        hipStream_t st = L.begin(BIN);                                                         \
        if (write_col & 1)                                                                     \
            hipLaunchKernelGGL((k_num_dense<BS, SPAN, 1>), dim3(8 * ceil_div(hist[BIN], 8)), dim3(BS), \
                               0, st, arpt, acol, aval, brpt, bcol, bval, c->d_rpt, c->d_col,  \
                               c->d_val, row_perm, row_prod, row_maxb, row_lo, row_span, off[BIN],       \
                               hist[BIN], b->nnz, bm_off, bm);                                 \
        else                                                                                   \
            hipLaunchKernelGGL((k_num_dense<BS, SPAN, 2>), dim3(8 * ceil_div(hist[BIN], 8)), dim3(BS), \
                               0, st, arpt, acol, aval, brpt, bcol, bval, c->d_rpt, c->d_col,  \
                               c->d_val, row_perm, row_prod, row_maxb, row_lo, row_span, off[BIN],       \
                               hist[BIN], b->nnz, bm_off, bm);                                 \
        NSP_LAUNCH_CHECK();                                                                    \
        L.end(BIN);                                                                            \
    }
    static const int tune_nd6 = getenv("NSPARSE_NUMD6_BS") ? atoi(getenv("NSPARSE_NUMD6_BS")) : 256;
    NSP_NUM_DENSE(8, 512, 12288)
    NSP_NUM_DENSE(7, 256, 4096)
    if (tune_nd6 == 128) { NSP_NUM_DENSE(6, 128, 1536) } else if (tune_nd6 == 512) { NSP_NUM_DENSE(6, 512, 1536) } else { NSP_NUM_DENSE(6, 256, 1536) }
#undef NSP_NUM_DENSE
    static const int tune_n2 = getenv("NSPARSE_NUM2_BS") ? atoi(getenv("NSPARSE_NUM2_BS")) : 256;
    static const int tune_n1 = getenv("NSPARSE_NUM1_BS") ? atoi(getenv("NSPARSE_NUM1_BS")) : 64;
    NSP_NUM_TB(4, 1024, 8192, 8192)
    NSP_NUM_TB(3, 512, 4096, 4096)
    if (tune_n2 == 128) { NSP_NUM_TB(2, 128, 1024, 1024) } else if (tune_n2 == 512) { NSP_NUM_TB(2, 512, 1024, 1024) } else { NSP_NUM_TB(2, 256, 1024, 1024) }
    if (tune_n1 == 128) { NSP_NUM_TB(1, 128, 256, 256) } else { NSP_NUM_TB(1, 64, 256, 256) }
#undef NSP_NUM_TB
    if (hist[0] > 0) {
        hipStream_t st = L.begin(0);
        constexpr int BS = 256, LPR = 4;
        hipLaunchKernelGGL((k_num_small<BS, LPR, 32>), dim3(ceil_div(hist[0], BS / LPR)), dim3(BS), 0,
                           st, arpt, acol, aval, brpt, bcol, bval, c->d_rpt, c->d_col, c->d_val,
                           row_perm, off[0], hist[0], write_col);
        NSP_LAUNCH_CHECK();
        L.end(0);
    }
    // The heavy bin synchronises on the host (its scratch slab is freed here), so it is issued
    // last: every other bin is already queued on its own stream and overlaps with it.
    constexpr int kTileW = sizeof(real) == 8 ? 12288 : 24576;
    static const int tiled_on = !(getenv("NSPARSE_TILED") && getenv("NSPARSE_TILED")[0] == '0');
    static const int long_len = getenv("NSPARSE_TILED_LONG") ? atoi(getenv("NSPARSE_TILED_LONG")) : 128;
    static const int tile_sel = getenv("NSPARSE_TILED_W") ? atoi(getenv("NSPARSE_TILED_W")) : 0;
    // rows with fewer than one non-zero per ranked_dens columns of their window take the ranked
    // kernel (0: none, < 0: all)
    static const int ranked_dens = getenv("NSPARSE_RANKED_DENS") ? atoi(getenv("NSPARSE_RANKED_DENS")) : 12;
    // dense tiles alone: at most 1024 per row, wider matrices hash globally; with the ranked kernel
    // taking the wide rows there is no limit
    const bool use_tiled = tiled_on && b_sorted && hist[kNumGlobalBin] > 0 && max_alen > 0 &&
                           (ranked_dens != 0 || (long long)b->N <= (long long)kTileW * 1024);
    if (use_tiled) {
        hipStream_t st = L.begin(kNumGlobalBin);
        const int rows = hist[kNumGlobalBin];
        const int amax = (max_alen + 1) & ~1;  // even: the value slice stays 8-byte aligned
        const long long stride_ints = 3LL * amax + (long long)amax * (sizeof(real) / sizeof(int));
        const int groups = rows < 1024 ? rows : 1024;
        int *slab = (int *)dev_alloc(sizeof(int) * (size_t)stride_ints * groups);
        static const int tiled_prof = getenv("NSPARSE_TILED_PROF") ? atoi(getenv("NSPARSE_TILED_PROF")) : 0;
        unsigned long long *d_prof = nullptr;
        if (tiled_prof) {
            d_prof = (unsigned long long *)dev_alloc(32 * sizeof(unsigned long long));
            NSP_CHECK(hipMemsetAsync(d_prof, 0, 32 * sizeof(unsigned long long), st));
        }
#define NSP_TILED(BSX, WX)                                                                     \
    hipLaunchKernelGGL((k_num_tiled<BSX, WX>), dim3(groups), dim3(BSX), 0, st, arpt, acol, aval, brpt, \
                       bcol, bval, c->d_rpt, c->d_col, c->d_val, row_perm, off[kNumGlobalBin], rows,   \
                       d_bs, row_lo, row_span, slab, stride_ints, amax, write_col, long_len, d_prof, ranked_dens)
        if (ranked_dens >= 0) {
            if (tile_sel == 1) { NSP_TILED(1024, kTileW / 2); }
            else if (tile_sel == 2) { NSP_TILED(512, kTileW / 2); }
            else if (tile_sel == 3) { NSP_TILED(512, kTileW / 4); }
            else { NSP_TILED(1024, kTileW); }
        }
#undef NSP_TILED
        NSP_LAUNCH_CHECK();
        // thin rows: bitmap-ranked accumulator (same stream: both kernels want the whole LDS of a CU)
        if (ranked_dens != 0) {
            constexpr int kRankCap = sizeof(real) == 8 ? 10240 : 20480;
            hipLaunchKernelGGL((k_num_ranked<1024, 262144, kRankCap>), dim3(groups), dim3(1024), 0, st, arpt, acol,
                               aval, brpt, bcol, bval, c->d_rpt, c->d_col, c->d_val, row_perm, off[kNumGlobalBin],
                               rows, d_bs, row_lo, row_span, slab, stride_ints, amax, write_col, long_len,
                               ranked_dens, tile_sel == 0 ? kTileW : (tile_sel == 3 ? kTileW / 4 : kTileW / 2), d_prof);
            NSP_LAUNCH_CHECK();
        }
        NSP_CHECK(hipStreamSynchronize(st));
        L.end(kNumGlobalBin);
        if (d_prof) {
            unsigned long long h[32];
            NSP_CHECK(hipMemcpy(h, d_prof, sizeof(h), hipMemcpyDeviceToHost));
            const double us = 0.01 / groups;  // 100 MHz ticks summed over the workgroups
            fprintf(stderr, "[tiled] groups %d rows %llu tiles %llu | per-group us: setup %.0f queue %.0f regs %.0f overflow %.0f emit %.0f+%.0f+%.0f+%.0f | trips %llu (long %llu)\n",
                    groups, h[6], h[5], h[0] * us, h[1] * us, h[2] * us, h[3] * us, h[7] * us, h[8] * us, h[9] * us, h[4] * us, h[10], h[11]);
            fprintf(stderr, "[ranked] rows %llu tiles %llu | per-group us: setup %.0f pass1 %.0f (first tile %.0f) scan %.0f pass2 %.0f (first %.0f) emit %.0f | cuts %llu | sum nlong %llu alen %llu list-overflow rows %llu\n",
                    h[22], h[21], h[16] * us, h[17] * us, h[23] * us, h[18] * us, h[19] * us, h[24] * us, h[20] * us, h[25], h[26], h[27], h[28]);
            dev_free(d_prof);
        }
        dev_free(slab);
    } else if (hist[kNumGlobalBin] > 0) {
        hipStream_t st = L.begin(kNumGlobalBin);
        const int rows = hist[kNumGlobalBin];
        long long slice = 64;
        while (slice < 2LL * max_nz) slice <<= 1;
        const int groups = global_slab_groups(slice, sizeof(int) + sizeof(real), rows);
        int *kslab = (int *)dev_alloc(sizeof(int) * (size_t)slice * groups);
        real *vslab = (real *)dev_alloc(sizeof(real) * (size_t)slice * groups);
        int *tcol = (int *)dev_alloc(sizeof(int) * (size_t)c->nnz);
        real *tval = (real *)dev_alloc(sizeof(real) * (size_t)c->nnz);
        int *seg = (int *)dev_alloc(sizeof(int) * 2 * (size_t)rows);
        // when the structure is kept (numeric-only re-run) sort into scratch columns
        const bool unsorted = (write_col & 2) != 0;  // only honoured together with bit 0
        int *out_col = (write_col & 1) ? c->d_col : (int *)dev_alloc(sizeof(int) * (size_t)c->nnz);
        hipLaunchKernelGGL((k_num_global<512>), dim3(groups), dim3(512), 0, st, arpt, acol, aval, brpt,
                           bcol, bval, c->d_rpt, unsorted ? c->d_col : tcol, unsorted ? c->d_val : tval,
                           row_perm, off[kNumGlobalBin], rows, d_bs, kslab, vslab, slice, seg, seg + rows);
        NSP_LAUNCH_CHECK();
        size_t tmp_bytes = 0;
        if (!unsorted) {
        NSP_CHECK(rocprim::segmented_radix_sort_pairs(nullptr, tmp_bytes, tcol, out_col, tval, c->d_val,
                                                      (unsigned)c->nnz, (unsigned)rows, seg, seg + rows,
                                                      0, 32, st));
        void *tmp = dev_alloc(tmp_bytes ? tmp_bytes : 1);
        NSP_CHECK(rocprim::segmented_radix_sort_pairs(tmp, tmp_bytes, tcol, out_col, tval, c->d_val,
                                                      (unsigned)c->nnz, (unsigned)rows, seg, seg + rows,
                                                      0, 32, st));
        dev_free(tmp);
        }
        NSP_CHECK(hipStreamSynchronize(st));
        L.end(kNumGlobalBin);
        if (!(write_col & 1)) dev_free(out_col);
        dev_free(seg);
        dev_free(tval);
        dev_free(tcol);
        dev_free(vslab);
        dev_free(kslab);
    }
    L.join();
    return L;
}

static void run(sfCSR *a, sfCSR *b, sfCSR *c, bool numeric_only)
{
    clear_error();
    Context &cx = ctx();
    Timer tm(cx);
    hipStream_t s0 = cx.stream[0];
    const int M = a->M;
    nsparse_spgemm_stats &S = g_stats.s;
    memset(&S, 0, sizeof(S));
    bool too_big = false;
    tm.mark(0, s0);

    BinState *d_sym = reinterpret_cast<BinState *>(cx.d_scratch);
    BinState *d_num = d_sym + 1;
    BinState *h_sym = reinterpret_cast<BinState *>(cx.h_mapped);
    BinState *h_num = h_sym + 1;
    BinState *h_num_dev = reinterpret_cast<BinState *>(cx.d_mapped) + 1;
    static_assert(2 * sizeof(BinState) <= 120 * sizeof(int), "scratch layout");

    void *scan_tmp = nullptr;
    unsigned int *bm = nullptr;
    BinLauncher sym_used(cx, 0);
    int *row_prod = (int *)dev_alloc(sizeof(int) * (size_t)(M + 1));
    int *row_nz = (int *)dev_alloc(sizeof(int) * (size_t)(M + 1));
    int *row_perm = (int *)dev_alloc(sizeof(int) * (size_t)(M > 0 ? M : 1));
    int *row_lo = (int *)dev_alloc(sizeof(int) * (size_t)(M > 0 ? M : 1));
    int *row_span = (int *)dev_alloc(sizeof(int) * (size_t)(M > 0 ? M : 1));
    int *row_maxb = (int *)dev_alloc(sizeof(int) * (size_t)(M > 0 ? M : 1));
    const int K = b->M;
    BInfo *binfo = (BInfo *)dev_alloc(sizeof(BInfo) * (size_t)(K > 0 ? K : 1));
    int *long_list = (int *)dev_alloc(sizeof(int) * kLongCap);  // reused: B rows first, then A rows
    int *long_cnt = cx.d_scratch + 240;                         // [0] B pass, [1] A pass
    if (g_dense_enabled < 0) {
        const char *e = getenv("NSPARSE_DENSE");
        g_dense_enabled = !(e && e[0] == '0');
    }
    Thr sym_thr = kSymThr, num_thr = kNumThr;
    if (!g_dense_enabled) sym_thr.dense_ratio = num_thr.dense_ratio = sym_thr.bits_ratio = 0;
    NSP_CHECK(hipStreamSynchronize(0));  // inputs queued on the null stream by the caller
    NSP_CHECK(hipMemsetAsync(d_sym, 0, 2 * sizeof(BinState), s0));
    NSP_CHECK(hipMemsetAsync(long_cnt, 0, 2 * sizeof(int), s0));

    // ---- setup: column window of every B row, products + window per C row, symbolic bins ----
    {
        const int wb = pick_w(b->nnz, K);
        const int gb = ceil_div((long long)K * wb, 256);
        int *blist = (b->nnz_max > 0 && b->nnz_max <= kLongFactor * wb) ? nullptr : long_list;
#define NSP_BI(W)                                                                              \
    case W:                                                                                    \
        hipLaunchKernelGGL(k_b_info<W>, dim3(gb), dim3(256), 0, s0, b->d_rpt, b->d_col, K, binfo, d_sym, \
                           blist, long_cnt, kLongFactor * W, (const int *)nullptr);            \
        break;
        switch (wb) {
            NSP_BI(1) NSP_BI(2) NSP_BI(4) NSP_BI(8) NSP_BI(16) NSP_BI(32) NSP_BI(64)
        }
#undef NSP_BI
        if (blist)
            hipLaunchKernelGGL(k_b_info<64>, dim3(256), dim3(256), 0, s0, b->d_rpt, b->d_col, K, binfo, d_sym,
                               (int *)nullptr, long_cnt, 0, (const int *)long_list);
    }
    long long *partial = (long long *)dev_alloc(sizeof(long long) * kPartialStride * kSetupMaxGrid);
    // column bitmaps handed from the symbolic to the numeric dense kernels
    int *bm_words = (int *)dev_alloc(sizeof(int) * (size_t)(M + 1));
    int *bm_off = (int *)dev_alloc(sizeof(int) * (size_t)(M + 1));
    int *row_span_num = (int *)dev_alloc(sizeof(int) * (size_t)(M > 0 ? M : 1));
    const bool use_bm = !numeric_only && num_thr.dense_ratio > 0;
    launch_row_products(a, b, binfo, row_prod, row_lo, row_span, bm_words,
                        use_bm ? num_thr.dense_span[2] : 0, sym_thr, d_sym, partial, row_span_num, row_nz, row_maxb, long_list, long_cnt + 1, s0);
    void *bm_scan_tmp = nullptr;
    if (use_bm) bm_scan_tmp = scan_exclusive(bm_words, bm_off, M + 1, s0);
    const int grid_m = ceil_div(M, 1024);
    if (!numeric_only) {
        hipLaunchKernelGGL(k_bin_scatter, dim3(grid_m), dim3(1024), 0, s0, row_prod, row_span, M, sym_thr, d_sym, row_perm);
        NSP_LAUNCH_CHECK();
    }
    {
        const int seq = ++cx.seq;
        hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, s0, d_sym, cx.d_mapped, (int)(sizeof(BinState) / 4),
                           (const int *)nullptr, cx.d_mapped + 120, seq);
        NSP_LAUNCH_CHECK();
        wait_published(120, seq, s0);
    }
    S.n_prod = h_sym->total;
    S.max_prod_row = h_sym->maxv;
    for (int q = 0; q < NB; q++) S.sym_bin_size[q] = h_sym->hist[q];
    tm.mark(1, s0);

    // ---- symbolic: nnz of every row of C, then C.rpt ----------------------------------
    if (!numeric_only) {
        c->M = M;
        c->N = b->N;
        // bitmaps only when they fit int offsets comfortably; otherwise the dense window is
        // used by the symbolic phase alone and the numeric phase hashes
        if (use_bm && h_sym->bm_total > 0 && h_sym->bm_total < (1LL << 30))
            bm = (unsigned int *)dev_alloc(sizeof(unsigned int) * (size_t)h_sym->bm_total);
        BinLauncher LS = symbolic_phase(a, b, row_prod, row_maxb, row_lo, row_span, row_nz, row_perm, h_sym->hist,
                                        h_sym->maxv, d_sym, cx, S.ms_sym_bin, &S.sym_fail_rows,
                                        bm_off, bm, row_span_num);
        sym_used = LS;
        c->d_rpt = (int *)dev_alloc(sizeof(int) * (size_t)(M + 1));
        scan_tmp = scan_exclusive(row_nz, c->d_rpt, M + 1, s0);
    } else {
        // structure given: row_nz[i] = rpt[i+1] - rpt[i] is recovered inside the kernels
        // from C.rpt; for binning we need it explicitly.
        hipLaunchKernelGGL(k_row_len, dim3(ceil_div(M, 256)), dim3(256), 0, s0, c->d_rpt, row_nz, M);
        NSP_LAUNCH_CHECK();
    }
    tm.mark(2, s0);

    // ---- numeric binning ------------------------------------------------------------
    // numeric window: full call -> rows whose bitmap was written; re-run -> every eligible row
    const int *num_span = numeric_only ? row_span : row_span_num;
    if (!numeric_only && bm == nullptr) num_thr.dense_ratio = 0;
    hipLaunchKernelGGL(k_hist, dim3(grid_m), dim3(1024), 0, s0, row_nz, num_span, M, num_thr, d_num);
    hipLaunchKernelGGL(k_bin_scatter, dim3(grid_m), dim3(1024), 0, s0, row_nz, num_span, M, num_thr, d_num, row_perm);
    NSP_LAUNCH_CHECK();
    {
        const int seq = ++cx.seq;
        hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, s0, d_num, reinterpret_cast<int *>(h_num_dev),
                           (int)(sizeof(BinState) / 4), c->d_rpt + M, cx.d_mapped + 121, seq);
        NSP_LAUNCH_CHECK();
        wait_published(121, seq, s0);
    }
    for (int q = 0; q < NB; q++) S.num_bin_size[q] = h_num->hist[q];
    S.max_nnz_row = h_num->maxv;
    if (!numeric_only) {
        // upstream keeps nnz(C) and C.rpt in int (nsparse.h:62-75) and wraps silently; refuse
        too_big = h_num->total > 0x7fffffffLL;
    }
    if (!numeric_only && !too_big) {
        c->nnz = h_num->nnz;
        c->d_col = (int *)dev_alloc(sizeof(int) * (size_t)(c->nnz > 0 ? c->nnz : 1));
        c->d_val = (real *)dev_alloc(sizeof(real) * (size_t)(c->nnz > 0 ? c->nnz : 1));
    }
    S.nnz_c = too_big ? h_num->total : c->nnz;

    // ---- numeric --------------------------------------------------------------------
    if (!too_big) {
    BinLauncher LN = numeric_phase(a, b, c, row_prod, row_maxb, row_lo, row_span, row_perm, h_num->hist,
                                   h_num->maxv, d_num, cx, S.ms_num_bin,
                                   numeric_only ? 0 : (g_sorted ? 1 : 3), bm_off, bm,
                                   (int)h_sym->max_alen, h_sym->b_unsorted == 0);
    tm.mark(3, s0);
    {   // synchronous on return, like upstream (:1287): poll a flag raised behind the last kernel
        const int seq = ++cx.seq;
        hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, s0, d_num, cx.d_mapped + 128, 0, (const int *)nullptr,
                           cx.d_mapped + 122, seq);
        NSP_LAUNCH_CHECK();
        wait_published(122, seq, s0);
    }
    LN.collect(S.ms_num_bin);
    sym_used.collect(S.ms_sym_bin);
    S.ms_setup = tm.ms(0, 1);
    S.ms_symbolic = tm.ms(1, 2);
    S.ms_numeric = tm.ms(2, 3);
    S.ms_total = tm.ms(0, 3);
    } else {
        NSP_CHECK(hipStreamSynchronize(s0));
        dev_free(c->d_rpt);
        c->d_rpt = nullptr;
        c->d_col = nullptr;
        c->d_val = nullptr;
        c->nnz = 0;
    }

    dev_free(scan_tmp);
    dev_free(bm);
    dev_free(bm_scan_tmp);
    dev_free(row_span_num);
    dev_free(bm_off);
    dev_free(bm_words);
    dev_free(partial);
    dev_free(long_list);
    dev_free(binfo);
    dev_free(row_maxb);
    dev_free(row_span);
    dev_free(row_lo);
    dev_free(row_perm);
    dev_free(row_nz);
    dev_free(row_prod);
    if (too_big) {
        char msg[160];
        snprintf(msg, sizeof(msg), "nnz(C) = %lld does not fit the int row pointers of sfCSR", (long long)S.nnz_c);
        set_error(-40, msg, __FILE__, __LINE__);
    }
}

__global__ __launch_bounds__(256) void k_flop(const int *__restrict__ arpt, const int *__restrict__ acol,
                                              const int *__restrict__ brpt, int M,
                                              unsigned long long *total)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    unsigned long long n = 0;
    if (i < M)
        for (int j = arpt[i]; j < arpt[i + 1]; j++) n += (unsigned long long)(brpt[acol[j] + 1] - brpt[acol[j]]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) n += __shfl_xor(n, o);
    if ((threadIdx.x & 63) == 0 && n) atomicAdd(total, n);
}

}  // namespace spgemm
}  // namespace nsp

extern "C" {

void spgemm_kernel_hash(sfCSR *a, sfCSR *b, sfCSR *c) { nsp::spgemm::run(a, b, c, false); }

void nsparse_spgemm_hash_numeric(sfCSR *a, sfCSR *b, sfCSR *c) { nsp::spgemm::run(a, b, c, true); }

void nsparse_get_spgemm_stats(nsparse_spgemm_stats *out) { *out = nsp::spgemm::g_stats.s; }

int nsparse_spgemm_set_sorted(int on)
{
    const int old = nsp::spgemm::g_sorted;
    nsp::spgemm::g_sorted = on ? 1 : 0;
    return old;
}

void nsparse_get_spgemm_bins(int *sym, int *num)
{
    // 15 ints each: tiny, hash_t[4], dense_span[3], dense_ratio, bits_span[2], bits_ratio,
    // bits_min, bits_wide_min, bits_wide_span (ratios are 0 when NSPARSE_DENSE=0)
    const nsp::spgemm::Thr *t[2] = {&nsp::spgemm::kSymThr, &nsp::spgemm::kNumThr};
    int *out[2] = {sym, num};
    if (nsp::spgemm::g_dense_enabled < 0) {
        const char *e = getenv("NSPARSE_DENSE");
        nsp::spgemm::g_dense_enabled = !(e && e[0] == '0');
    }
    for (int p = 0; p < 2; p++) {
        out[p][0] = t[p]->tiny;
        for (int q = 0; q < 4; q++) out[p][1 + q] = t[p]->hash_t[q];
        for (int q = 0; q < 3; q++) out[p][5 + q] = t[p]->dense_span[q];
        out[p][8] = nsp::spgemm::g_dense_enabled ? t[p]->dense_ratio : 0;
        out[p][9] = t[p]->bits_span[0];
        out[p][10] = t[p]->bits_span[1];
        out[p][11] = nsp::spgemm::g_dense_enabled ? t[p]->bits_ratio : 0;
        out[p][12] = t[p]->bits_min;
        out[p][13] = t[p]->bits_wide_min;
        out[p][14] = t[p]->bits_wide_span;
    }
}

void get_spgemm_flop(sfCSR *a, sfCSR *b, int M, long long int *flop)
{
    nsp::clear_error();
    nsp::Context &cx = nsp::ctx();
    unsigned long long *d_total = reinterpret_cast<unsigned long long *>(cx.d_scratch + 192);
    NSP_CHECK(hipMemsetAsync(d_total, 0, sizeof(unsigned long long), cx.stream[0]));
    hipLaunchKernelGGL(nsp::spgemm::k_flop, dim3(nsp::ceil_div(M, 256)), dim3(256), 0, cx.stream[0],
                       a->d_rpt, a->d_col, b->d_rpt, M, d_total);
    NSP_LAUNCH_CHECK();
    unsigned long long h = 0;
    NSP_CHECK(hipMemcpyAsync(&h, d_total, sizeof(h), hipMemcpyDeviceToHost, cx.stream[0]));
    NSP_CHECK(hipStreamSynchronize(cx.stream[0]));
    *flop = (long long)(2 * h);
}

}  // extern "C"
