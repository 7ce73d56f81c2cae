// spgemm/setup.h -- products per row, column windows, histogram, bin-grouped row permutation.
// Part of the spgemm_hash.hip translation unit (kernels are launched from its host code).
#pragma once
#include "common.h"

namespace nsp {
namespace spgemm {

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

// Rows of B that A can reach: [0x7fffffff - range[0], range[1]), both kept as maxima so that
// zero means "nothing seen".  When A is a row block of a larger problem (1-D row partition, B
// replicated) only that stretch of B needs its records.  Two steps without same-address atomics
// (they serialise across the 8 XCDs at ~20 ns each): per-workgroup partials, then one workgroup
// folds them.
__global__ __launch_bounds__(256) void k_col_range(const int *__restrict__ acol, int nnz,
                                                   unsigned int *__restrict__ part)
{
    __shared__ unsigned int s_a[4], s_b[4];
    unsigned int a = 0, b = 0;
    auto see = [&](int col) {
        const unsigned int c = (unsigned int)col;
        a = a > 0x7fffffffu - c ? a : 0x7fffffffu - c;
        b = b > c + 1 ? b : c + 1;
    };
    const int n4 = nnz >> 2;
    const int4 *c4 = reinterpret_cast<const int4 *>(acol);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
        const int4 v = c4[i];
        see(v.x);
        see(v.y);
        see(v.z);
        see(v.w);
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < (nnz & 3)) see(acol[4 * n4 + threadIdx.x]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned int x = __shfl_xor(a, o), y = __shfl_xor(b, o);
        a = x > a ? x : a;
        b = y > b ? y : b;
    }
    if ((threadIdx.x & 63) == 0) {
        s_a[threadIdx.x >> 6] = a;
        s_b[threadIdx.x >> 6] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int u = 1; u < 4; u++) {
            a = s_a[u] > a ? s_a[u] : a;
            b = s_b[u] > b ? s_b[u] : b;
        }
        part[2 * blockIdx.x] = a;
        part[2 * blockIdx.x + 1] = b;
    }
}

__global__ __launch_bounds__(256) void k_col_range_fold(const unsigned int *__restrict__ part, int nparts,
                                                        unsigned int *__restrict__ range)
{
    __shared__ unsigned int s_a[4], s_b[4];
    unsigned int a = 0, b = 0;
    for (int i = threadIdx.x; i < nparts; i += 256) {
        a = part[2 * i] > a ? part[2 * i] : a;
        b = part[2 * i + 1] > b ? part[2 * i + 1] : b;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned int x = __shfl_xor(a, o), y = __shfl_xor(b, o);
        a = x > a ? x : a;
        b = y > b ? y : b;
    }
    if ((threadIdx.x & 63) == 0) {
        s_a[threadIdx.x >> 6] = a;
        s_b[threadIdx.x >> 6] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int u = 1; u < 4; u++) {
            a = s_a[u] > a ? s_a[u] : a;
            b = s_b[u] > b ? s_b[u] : b;
        }
        range[0] = a;
        range[1] = b;
    }
}

// contribution of one column id to the order-independent key of a row pattern (splitmix64 finaliser)
__device__ __forceinline__ unsigned long long col_key(int c)
{
    unsigned long long z = (unsigned long long)(unsigned)c + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}


// Is twin detection worth its price on this matrix?  On a matrix without twin rows (a scalar stencil) the pattern map
// costs k_row_products one returning 64-bit atomic per row and more than doubles its time (27-point stencil, 10^6 rows:
// 121 -> 263 us), for nothing.  So for matrices of kTwinSampleMin rows and more k_b_info -- which runs first anyway --
// looks at a SAMPLE of the rows of A: the first 64 rows of every 1024 (consecutive rows, so that neighbouring twins
// are seen; a sixteenth of the matrix, so that scattered ones are seen by chance), rows of 1 .. 4096 entries, one
// wavefront per row.  Their pattern keys go into a table (48 bits of the key + a 16-bit tag of this call, so the table
// needs no clearing: words with another tag are free); the first key that is already there raises bs->twin_sample,
// and k_row_products probes only if it is raised.  Exact on the sample (tests/gpu_util.py: twin_rows mirrors the
// rule); a miss only costs the optimisation, never the result.
constexpr int kTwinSampleMin = 131072;
struct TwinSample {
    const int *arpt;
    const int *acol;
    int M;
    unsigned long long *tab;  // nullptr: no sampling (small matrices: always probe)
    unsigned int mask;
    unsigned int tag;         // 1 .. 65535
};

template <int W>
__global__ __launch_bounds__(256) void k_b_info(const int *__restrict__ brpt, const int *__restrict__ bcol,
                                                int K, BInfo *__restrict__ info, BinState *bs,
                                                int *__restrict__ long_list, int *long_cnt, int long_len,
                                                const int *__restrict__ todo,
                                                const unsigned int *__restrict__ range,
                                                unsigned char *__restrict__ btwin,
                                                unsigned long long *__restrict__ fill, long long fill_words,
                                                TwinSample ts = TwinSample{nullptr, nullptr, 0, nullptr, 0u, 0u})
{
    if (ts.tab) {  // (first: the workgroups that find no rows of B below must not leave before it)
        const long long ns = (long long)((ts.M + 1023) >> 10) * 64;
        constexpr int SL = 16;  // lanes per sampled row (a wavefront per row was 29 us of the stencil's k_b_info, this is 8)
        const int wl = threadIdx.x & (SL - 1);
        for (long long q = (long long)blockIdx.x * (256 / SL) + (threadIdx.x / SL); q < ns;
             q += (long long)gridDim.x * (256 / SL)) {
            // (the answer is in: a web graph or a finite-element matrix raises the flag within the first rows, and the
            //  rest of the sample -- 25 us of a 0.65 ms call on the 1 M-row web graph -- is skipped)
            if (__hip_atomic_load(&bs->twin_sample, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
            const long long r64 = ((q >> 6) << 10) + (q & 63);
            int b = 0, len = 0;
            if (r64 < ts.M) {
                b = ts.arpt[r64];
                len = ts.arpt[r64 + 1] - b;
            }
            if (len > 4096) len = 0;  // (not sampled)
            unsigned long long key = 0;
            for (int k = wl; k < len; k += SL) key += col_key(ts.acol[b + k]);
#pragma unroll
            for (int o = SL / 2; o >= 1; o >>= 1) key += __shfl_xor(key, o);
            if (len < 1) continue;
            if (wl == 0) {
                key += 0x9E3779B97F4A7C15ull * (unsigned long long)len;
                const unsigned long long mine = (key & ~0xFFFFull) | ts.tag;
                unsigned int sl = (unsigned int)(key >> 20) & ts.mask;
                for (unsigned int tries = 0; tries <= ts.mask; tries++) {
                    unsigned long long cur = __hip_atomic_load(ts.tab + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned int)(cur & 0xFFFFull) != ts.tag) {  // free: a word of another call (or none)
                        const unsigned long long old = atomicCAS(ts.tab + sl, cur, mine);
                        if (old == cur) break;
                        cur = old;
                        if ((unsigned int)(cur & 0xFFFFull) != ts.tag) continue;  // (cannot happen: only this call writes)
                    }
                    if (cur == mine) {
                        // a look first: on a web graph half the sampled rows repeat, and same-address device-scope
                        // stores serialise at ~20 ns each (measured: 30 K of them, +0.55 ms on the webbase-1M class)
                        if (__hip_atomic_load(&bs->twin_sample, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                            __hip_atomic_store(&bs->twin_sample, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                    sl = (sl + 1) & ts.mask;
                }
            }
        }
    }
    // fill: fill_words 64-bit words set to all ones on the way (the twin map of k_row_products, which runs
    // next: a fill launch less).
    for (long long w = (long long)blockIdx.x * 256 + threadIdx.x; w < fill_words; w += (long long)gridDim.x * 256)
        fill[w] = ~0ull;
    // btwin != nullptr: btwin[r] = 1 when row r of B has exactly the column pattern of row r - 1 (the
    // degrees of freedom of one mesh node): the numeric window kernel then treats consecutive A entries
    // that point at such rows as one run (block.h).  Long rows (second pass) are never marked.
    // W lanes per row of B (W from the average row length, so the column loads coalesce).
    // todo == nullptr: bulk pass over all rows (over the rows k_col_range found when range is
    // given); rows longer than long_len are deferred to long_list.  todo != nullptr: pass over
    // the deferred rows todo[0 .. min(*long_cnt, cap)).
    constexpr int RPB = 256 / W;
    const int lane = threadIdx.x % W;
    int r0 = 0, r1 = K;
    if (range) {
        r0 = (int)(0x7fffffffu - range[0]);
        r1 = (int)range[1] < K ? (int)range[1] : K;
        if (r1 <= r0) return;
    }
    const int nrows = todo ? (*long_cnt < kLongCap ? *long_cnt : kLongCap) : r1 - r0;
    for (int base = blockIdx.x * RPB; base < nrows; base += gridDim.x * RPB) {
        const int q = base + (int)threadIdx.x / W;
        const int r = q < nrows ? (todo ? todo[q] : r0 + q) : -1;
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
        int differs = 1;
        if (r >= 0 && !defer) {
            const int len = e - b;
            // candidate twin of the row before: same length (that row ends where this one starts)
            const bool cand = btwin && !todo && r > 0 && len > 0 && b - brpt[r - 1] == len;
            differs = cand ? 0 : 1;
            for (int k = b + lane; k < e; k += W) {
                const int c = bcol[k];
                if (k > b) bad |= c <= bcol[k - 1];  // strictly ascending?  (neighbour is in cache)
                if (cand) differs |= bcol[k - len] != c;
                lo = c < lo ? c : lo;
                hi = c > hi ? c : hi;
            }
        }
#pragma unroll
        for (int o = W / 2; o >= 1; o >>= 1) {
            const int l = __shfl_xor(lo, o), h = __shfl_xor(hi, o);
            differs |= __shfl_xor(differs, o);
            lo = l < lo ? l : lo;
            hi = h > hi ? h : hi;
        }
        if (bad) atomicOr(&bs->b_unsorted, 1);
        if (r >= 0 && !defer && lane == 0) {
            if (btwin) btwin[r] = differs == 0 ? 1 : 0;
            BInfo o;
            o.start = b;
            o.len = e - b;
            o.lo = lo;
            o.hi = hi;
            info[r] = o;
        }
    }
}

// Rows of A with EQUAL column patterns ("twin rows": the degrees of freedom of one mesh node), wherever
// they are: a hash map from pattern to the first row that claimed it, probed by the W lanes that have
// just walked the row in k_row_products.  An empty slot is claimed with one CAS (the row becomes the
// LEADER of its pattern); an occupied one is compared column by column with its owner -- equal: the row
// is a twin of that leader, else the next slot.  A twin does not go through the symbolic phase
// (k_twin_copy hands it the leader's result); the lowest and the highest twin that signed up are recorded as
// its group members for the numeric window kernel (block.h).  table / fcnt / members start as all ones.
struct TwinMap {
    unsigned long long *table;  // slot -> (first entry of the leader's row << 32 | leader row); ~0 = free
    unsigned int mask;           // slots - 1
    int a_nnz;
    int *twin_of;  // row -> leader or -1
    unsigned char *twin;
    int *fcnt;     // leader -> followers that signed up - 1 (only a brake, see below)
    int *members;  // leader -> the lowest and the highest of the followers that signed up (-1: none)
    const int *sample_flag;  // nullptr: always probe; else probe only if *sample_flag != 0 (TwinSample, k_b_info)
};

template <int W>
__device__ __forceinline__ int twin_probe(const int *__restrict__ arpt, const int *__restrict__ acol, int r,
                                          unsigned long long key, const TwinMap &tm, int lane)
{
    const int beg = arpt[r], len = arpt[r + 1] - beg;
    int leader = -1;
    if (len > 0) {
        key += 0x9E3779B97F4A7C15ull * (unsigned long long)len;
        unsigned int b = (unsigned int)(key >> 20) & tm.mask;
        const unsigned long long mine = ((unsigned long long)(unsigned int)beg << 32) | (unsigned int)r;
        while (true) {
            unsigned long long cur = 0;
            if (lane == 0) {
                // short rows look before they claim: thousands of them may share one pattern (one-entry rows
                // that point at a hub page) and would queue on the slot; rows of kTwinEager entries and more
                // claim at once -- one round trip instead of two for a leader
                cur = len >= kTwinEager ? ~0ull
                                        : __hip_atomic_load(tm.table + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cur == ~0ull) {
                    const unsigned long long old = atomicCAS(tm.table + b, ~0ull, mine);
                    cur = old == ~0ull ? mine : old;
                }
            }
            cur = __shfl(cur, 0, W);
            if (cur == mine) break;  // this row leads its pattern
            const int lrow = (int)(unsigned int)cur, cb = (int)(cur >> 32);
            // the owner's length and its columns in one round of loads (indices clamped: a shorter owner
            // must not be read past the end of A)
            const int ce = arpt[lrow + 1];
            int differs = 0;
            for (int j = lane; j < len; j += 4 * W) {
                int x[4], y[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int jj = j + u * W < len ? j + u * W : len - 1;
                    const int k2 = cb + jj < tm.a_nnz ? cb + jj : tm.a_nnz - 1;
                    x[u] = acol[beg + jj];
                    y[u] = acol[k2];
                }
#pragma unroll
                for (int u = 0; u < 4; u++) differs |= x[u] != y[u];
            }
            differs |= (ce - cb) != len;
#pragma unroll
            for (int o = W / 2; o >= 1; o >>= 1) differs |= __shfl_xor(differs, o);
            if (!differs) {
                leader = lrow;
                break;
            }
            b = (b + 1) & tm.mask;
        }
    }
    if (lane == 0) {
        tm.twin_of[r] = leader;
        tm.twin[r] = leader >= 0 ? 1 : 0;
        // sign up as a group member: no value comes back, nothing waits.  The counter is only a brake:
        // the thousands of one-entry rows that point at the same hub page would otherwise queue on
        // three addresses.
        if (leader >= 0 &&
            (len >= kTwinEager ||
             __hip_atomic_load(tm.fcnt + leader, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < kGroupMembers - 1)) {
            __hip_atomic_fetch_add(tm.fcnt + leader, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_min((unsigned int *)tm.members + kGroupMembers * leader, (unsigned int)r,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_max(tm.members + kGroupMembers * leader + 1, r, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    return leader;
}

// Pattern leaders of the rows of B (round 3).  The keyed runs of the node-block kernel (block.h) group the
// entries of an A row by the pattern leader of their row of B.  For C = A * A those are A's own leaders
// (k_row_products); for a general B -- a row block of a partitioned C = A * A is that case: A has fewer rows
// than B -- the rows of B that A reaches get their own pattern map here: W lanes sum the order-independent key
// of a row's columns and probe the map exactly as k_row_products does for A.  tm.twin_of starts as all -1
// (rows outside the reach keep that), table / fcnt / members as all ones.
template <int W>
__global__ __launch_bounds__(256) void k_b_twins(const int *__restrict__ brpt, const int *__restrict__ bcol, int K,
                                                 const unsigned int *__restrict__ range, TwinMap tm)
{
    constexpr int RPB = 256 / W;
    const int lane = threadIdx.x % W;
    int r0 = 0, r1 = K;
    if (range) {
        r0 = (int)(0x7fffffffu - range[0]);
        r1 = (int)range[1] < K ? (int)range[1] : K;
        if (r1 <= r0) return;
    }
    const int nrows = r1 - r0;
    for (int base = blockIdx.x * RPB; base < nrows; base += gridDim.x * RPB) {
        const int q = base + (int)threadIdx.x / W;
        const int r = q < nrows ? r0 + q : -1;
        unsigned long long key = 0;
        if (r >= 0) {
            const int e = brpt[r + 1];
            for (int k = brpt[r] + lane; k < e; k += W) key += col_key(bcol[k]);
        }
#pragma unroll
        for (int o = W / 2; o >= 1; o >>= 1) key += __shfl_xor(key, o);
        if (r >= 0) (void)twin_probe<W>(brpt, bcol, r, key, tm, lane);
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
                                                      int long_len, const int *__restrict__ todo,
                                                      TwinMap tw)
{
    // todo == nullptr: bulk pass, rows of A longer than long_len are deferred to long_list;
    // todo != nullptr: the deferred rows (see kLongFactor)
    // tw.table != nullptr: an order-independent 64-bit key of the row's column pattern is summed on the
    // way and twin_probe groups the rows with EQUAL patterns (the degrees of freedom of one node of a
    // finite-element mesh, wherever the numbering put them).
    const int nrows = todo ? (*long_cnt < kLongCap ? *long_cnt : kLongCap) : M;
    // big matrices: the pattern map is used only when the sample of k_b_info saw a pattern twice (TwinSample)
    // (written by the kernel before this one: a plain, scalar load -- it travels with the kernel arguments instead of
    //  being one more vector round trip in front of every workgroup's two batches of rows)
    const bool probe = tw.table != nullptr && (tw.sample_flag == nullptr || *tw.sample_flag != 0);
    if (!todo && blockIdx.x == 0 && threadIdx.x == 0) {  // scan tails (instead of two memset launches)
        bm_words[M] = 0;
        row_nz[M] = 0;
    }
    __shared__ int s_hist[NB];
    __shared__ int s_max;
    __shared__ unsigned long long s_total;
    __shared__ unsigned long long s_bm;
    __shared__ unsigned long long s_list;
    __shared__ int s_alen;
    if (threadIdx.x < NB) s_hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) { s_max = 0; s_total = 0; s_bm = 0; s_alen = 0; s_list = 0; }
    __syncthreads();
    const int lane = threadIdx.x % W;
    constexpr int RPB = 256 / W;
    // per-thread statistics, folded once per wave at the end: one LDS atomic per row would
    // serialise when (almost) every row falls into the same bin (1 M-row power-law inputs)
    int t_max = 0, t_alen = 0;
    unsigned long long t_total = 0, t_bm = 0, t_list = 0;
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
        unsigned long long key = 0;
        if (row < M) {
            const int e = arpt[row + 1];
            int j = arpt[row] + lane;
            for (; j + 3 * W < e; j += 4 * W) {  // four independent gathers in flight
                int c[4];
                BInfo bi[4];
#pragma unroll
                for (int u = 0; u < 4; u++) c[u] = __builtin_nontemporal_load(acol + j + u * W);
                if (probe) {
#pragma unroll
                    for (int u = 0; u < 4; u++) key += col_key(c[u]);
                }
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
                if (probe) key += col_key(c);
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
            key += __shfl_xor(key, o);
            lo = l < lo ? l : lo;
            hi = h > hi ? h : hi;
            mb = m2 > mb ? m2 : mb;
        }
        int leader = -1;
        // (rows of the tiny symbolic bin are not worth a probe: nothing to skip there, and they are no
        //  candidates for the node-block groups -- on a web graph that is most of a million rows)
        if (tw.table && row < M) {
            if (probe && n > thr.tiny) {
                leader = twin_probe<W>(arpt, acol, row, key, tw, lane);
            } else if (lane == 0) {
                tw.twin_of[row] = -1;
                tw.twin[row] = 0;
            }
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
            bin = leader < 0 ? bin_of(ni, span, thr, ni) : -1;
            // words of the column bitmap the symbolic dense kernel hands to the numeric one (twins share
            // their leader's and stay out of the symbolic bins).  Only k_sym_dense (bins 6-8) writes one, so
            // only its rows reserve one: a row with a window of up to bm_span_max = 65536 columns that hashes
            // (A^2 of a 7-point stencil on 64^3: 2 KB per row, 0.5 GB never written) reserves nothing
            const bool dense_row = bin >= kDenseBin0 && bin < kDenseBin0 + 3;
            const int bw = (dense_row && span > 0 && span <= bm_span_max) ? (span + 31) >> 5 : 0;
            bm_words[row] = bw;
            row_span_num[row] = 0;  // set by k_sym_dense when it hands a bitmap over
            // bit-window rows can hand their sorted column list to the numeric phase: room for it
            // (also the rows of the two big-table hash bins that can be heavy in the numeric phase: symbolic.h)
            const int lcap = ((bin == kBitsBin0 || bin == kBitsBin0 + 1 || bin == 3 || bin == 4) && ni > kListMinNnz)
                                 ? (ni < span ? ni : span) : 0;
            const int al = arpt[row + 1] - arpt[row];
            if (W >= 16) {  // at most 4 rows per wave: direct LDS atomics are cheapest
                if (bin >= 0) atomicAdd(&s_hist[bin], 1);
                atomicMax(&s_max, ni);
                atomicMax(&s_alen, al);
                atomicAdd(&s_total, (unsigned long long)n);
                if (bw) atomicAdd(&s_bm, (unsigned long long)bw);
                if (lcap) atomicAdd(&s_list, (unsigned long long)lcap);
            } else {
                t_max = ni > t_max ? ni : t_max;
                t_alen = al > t_alen ? al : t_alen;
                t_total += (unsigned long long)n;
                t_bm += (unsigned long long)bw;
                t_list += (unsigned long long)lcap;
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
            t_list += __shfl_xor(t_list, o);
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMax(&s_max, t_max);
            atomicMax(&s_alen, t_alen);
            atomicAdd(&s_total, t_total);
            atomicAdd(&s_bm, t_bm);
            if (t_list) atomicAdd(&s_list, t_list);
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
        out[NB + 4] = (long long)s_list;
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
    // per field: thread t handles field (t % 32) of partials t/32, t/32 + 8, ... of its slice
    const int per = (nblocks + gridDim.x - 1) / gridDim.x;
    const int b0 = blockIdx.x * per, b1 = b0 + per < nblocks ? b0 + per : nblocks;
    static_assert(kPartialStride == 32, "field = thread & 31");
    const int f = threadIdx.x & 31;
    if (f < NB + 5) {
        long long acc = 0;
        for (int b = b0 + (threadIdx.x >> 5); b < b1; b += 8) {
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
        if (s_acc[NB + 4]) atomicAdd((unsigned long long *)&bs->list_total, s_acc[NB + 4]);
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
    // lane 0's store below must land after the copy of the same word by another lane of this (one) wavefront: program
    // order of the wavefront -- said aloud (no instruction: a scheduling barrier, and the sync point of tests/emu)
    __builtin_amdgcn_wave_barrier();
    if (nnz_src && threadIdx.x == 0) reinterpret_cast<BinState *>(dst)->nnz = *nnz_src;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Is the STRUCTURE of B that of A (same shape and nnz given)?  C = A * A is usually called with two copies of
// A (the reference's sample uploads both); what was learnt about the rows of A -- their pattern classes --
// then holds for the rows of B (keyed runs, block.h).  Launched behind the set-up tail: it runs while the
// host is busy with the first flag, so it costs the call nothing.  bs->ab_differ is raised on a mismatch
// (a look first: when B is another matrix every workgroup finds one).
__global__ __launch_bounds__(256) void k_ab_compare(const int *__restrict__ arpt, const int *__restrict__ acol,
                                                    const int *__restrict__ brpt, const int *__restrict__ bcol,
                                                    int M, int nnz, BinState *bs, const int *__restrict__ need = nullptr)
{
    // need (big matrices, TwinSample): the answer only matters when there are twin rows; no repeat in the sample = no
    // probing = no twins, and 26 M entries of the stencil need not be compared (39 us behind the set-up tail)
    if (need && __hip_atomic_load(need, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
    bool diff = false;
    const int stride = gridDim.x * 256;
    for (int i = blockIdx.x * 256 + threadIdx.x; i <= M; i += stride) diff |= arpt[i] != brpt[i];
    const int n4 = ((reinterpret_cast<size_t>(acol) | reinterpret_cast<size_t>(bcol)) & 15) == 0 ? nnz >> 2 : 0;
    const int4 *a4 = reinterpret_cast<const int4 *>(acol), *b4 = reinterpret_cast<const int4 *>(bcol);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const int4 x = a4[i], y = b4[i];
        diff |= x.x != y.x || x.y != y.y || x.z != y.z || x.w != y.w;
    }
    for (int i = 4 * n4 + blockIdx.x * 256 + threadIdx.x; i < nnz; i += stride) diff |= acol[i] != bcol[i];
    if (__ballot(diff) != 0 && (threadIdx.x & 63) == 0 &&
        __hip_atomic_load(&bs->ab_differ, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
        atomicOr(&bs->ab_differ, 1);
}

// End of a call: zero the counter blocks for the next one (a fill launch less at its start -- the host
// keeps Context::counters_clean), then raise the flag the host polls.
__global__ __launch_bounds__(256) void k_finish(int *scratch, int *flag, int seq)
{
    if (threadIdx.x < 120) scratch[threadIdx.x] = 0;
    if (threadIdx.x >= 240 && threadIdx.x < 248) scratch[threadIdx.x] = 0;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Rows left out of the symbolic bins as twins take the result of the row they repeat.
__global__ __launch_bounds__(256) void k_twin_copy(const int *__restrict__ twin_of, int M,
                                                   int *__restrict__ row_nz, int *__restrict__ row_span_num,
                                                   int *__restrict__ bm_off)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= M) return;
    const int l = twin_of[r];
    if (l < 0) return;
    row_nz[r] = row_nz[l];
    row_span_num[r] = row_span_num[l];
    if (bm_off) bm_off[r] = bm_off[l];
}

// histogram of an existing per-row count (numeric binning, set_min_bin :201-246)
__global__ __launch_bounds__(1024) void k_hist(const int *__restrict__ n, const int *__restrict__ span,
                                              const int *__restrict__ work, int M, Thr thr, BinState *bs)
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
    // grid-stride over the rows with a bounded grid: the global atomics at the end all go to a dozen
    // addresses and serialise at ~20 ns each (a block per 1024 rows of a 1 M-row matrix: 26 us)
    int v = 0;
    unsigned long long sum = 0;
    for (int i0 = blockIdx.x * 1024; i0 < M; i0 += gridDim.x * 1024) {
        const int i = i0 + threadIdx.x;
        int bin = -1;
        if (i < M) {
            const int vi = n[i];
            bin = bin_of(vi, span[i], thr, work ? work[i] : vi);
            v = vi > v ? vi : v;
            sum += (unsigned long long)vi;
        }
        unsigned long long todo = __ballot(bin >= 0);
        while (todo) {  // one LDS atomic per (wave, bin present in the wave)
            const int leader = __ffsll((long long)todo) - 1;
            const int b = __shfl(bin, leader);
            const unsigned long long same = __ballot(bin == b);
            if ((threadIdx.x & 63) == leader) atomicAdd(&s_hist[b], __popcll(same));
            todo &= ~same;
        }
    }
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
// RPT rows per thread (1024 * RPT per workgroup): the cursor claims at the end are global atomics on
// a dozen addresses, ~20 ns each one after the other -- a workgroup per 1024 rows of a 1 M-row matrix
// spends 15 us there, a workgroup per 4096 rows 8.
template <int RPT>
__global__ __launch_bounds__(1024) void k_bin_scatter(const int *__restrict__ n,
                                                     const int *__restrict__ span,
                                                     const int *__restrict__ work, int M, Thr thr,
                                                     BinState *bs, int *__restrict__ perm,
                                                     const unsigned char *__restrict__ skip, int skip_mask)
{
    // skip: rows left out of the lists -- twin rows in the symbolic phase (mask 0xff), rows that follow
    // a group head in the numeric phase (grp, mask 3).  hist counts every row; cursor[bin] ends as the
    // number of rows actually listed.
    __shared__ int s_cnt[NB];
    __shared__ int s_base[NB];
    __shared__ int s_span[NB];
    if (threadIdx.x < NB) {
        s_cnt[threadIdx.x] = 0;
        s_span[threadIdx.x] = 0;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    int b[RPT], r[RPT];
    bool in[RPT];
#pragma unroll
    for (int k = 0; k < RPT; k++) {
        const int i = (blockIdx.x * RPT + k) * 1024 + threadIdx.x;
        b[k] = -1;
        r[k] = 0;
        in[k] = i < M && !(skip && (skip[i] & skip_mask));
        if (in[k]) {
            const int ni = n[i];
            b[k] = bin_of(ni, span[i], thr, work ? work[i] : ni);
            if (b[k] >= kDenseBin0) atomicMax(&s_span[b[k]], span[i]);  // window bins only
        }
        // rank inside the block: ballot + popcount inside the wave, one LDS atomic per (wave, bin)
        unsigned long long todo = __ballot(b[k] >= 0);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int bb = __shfl(b[k], leader);
            const unsigned long long same = __ballot(b[k] == bb);
            int base = 0;
            if (lane == leader) base = atomicAdd(&s_cnt[bb], __popcll(same));
            base = __shfl(base, leader);
            if (b[k] == bb) r[k] = base + __popcll(same & ((1ull << lane) - 1ull));
            todo &= ~same;
        }
        if (b[k] < 0) b[k] = 0;
    }
    __syncthreads();
    if (threadIdx.x < NB) {
        int off = 0;
        for (int q = 0; q < (int)threadIdx.x; q++) off += bs->hist[q];
        const int c = s_cnt[threadIdx.x];
        s_base[threadIdx.x] = off + (c ? atomicAdd(&bs->cursor[threadIdx.x], c) : 0);
        if (s_span[threadIdx.x]) atomicMax(&bs->max_span[threadIdx.x], s_span[threadIdx.x]);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RPT; k++)
        if (in[k]) perm[s_base[b[k]] + r[k]] = (blockIdx.x * RPT + k) * 1024 + threadIdx.x;
}

}  // namespace spgemm
}  // namespace nsp
