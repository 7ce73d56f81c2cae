// spgemm/common.h -- bin ladders, shared device helpers, the product walk.
// Part of the spgemm_hash.hip translation unit (kernels are launched from its host code).
#pragma once
#include <type_traits>

#include "../internal.h"

namespace nsp {
namespace spgemm {

constexpr int NB = kMaxBins;

// -DNSPARSE_EXPERIMENTS (make EXTRA=-DNSPARSE_EXPERIMENTS): the in-kernel phase timers, the LDS padding knobs and the
// workgroup-size / ladder overrides of the measurements in DESIGN.md 4.1.  The product build has none of them: the
// timers cost registers in the latency-bound row kernels, every override is a kernel instantiation nobody runs.
#ifdef NSPARSE_EXPERIMENTS
constexpr bool kExperiments = true;
#else
constexpr bool kExperiments = false;
#endif

// tests/emu only: census of the B entries a heavy-row kernel loads against the products it accumulates (family =
// FC_* below; what = 0 B.col loads, 1 B.val loads, 2 products, 3 tiles).  The product build compiles it away.
enum { FC_TILED = 0, FC_RANKED = 1, FC_RANKED_SYM = 2, FC_FLAT = 3, FC_WALK = 4, FC_RANKED_FLAT = 5,
       FC_FLAT_EXTENT = 6 /* what 0 / 1 / 2: entries inside their extent, k_num_flat / k_num_ranked_flat / k_sym_flat */,
       FC_SYM_FLAT = 7,
       FC_HASH_TB = 8, FC_HASH_LEAN = 9 /* what 0: keys handed to find-or-insert, 1: CAS probes issued, 2: keys that needed a retry */ };
#ifdef NSP_EMU
extern "C" void nsp_emu_count(int family, int what, long long n);
#define NSP_COUNT(family, what, n) nsp_emu_count((family), (what), (long long)(n))
#else
#define NSP_COUNT(family, what, n) ((void)0)
#endif
// a counted load of one B column index / one B value (heavy-row kernels; FC = the kernel's family)
#define NSP_LDC(FC, p, k) (NSP_COUNT(FC, 0, 1), (p)[k])
#define NSP_LDV(FC, p, k) (NSP_COUNT(FC, 1, 1), (p)[k])

// ---- bin ladders ------------------------------------------------------------------
// Symbolic, n = intermediate products of the row (upper bound of its nnz):
//   bin 0  n <= 32      sub-wave rows, 4 lanes per row, 64-key table per row
//   bin 1  n <= 870     one workgroup per row,   64 threads, table <=  1024 keys ( 4 KiB; round 3: was 435 / 512 --
//                       5 KB of LDS per one-wavefront row still means 32 rows per CU, bin 2 holds 15)
//   bin 2  n <= 1740                            128 threads,        <=  2048      ( 8 KiB)
//   bin 3  n <= 6963                            256 threads,        <=  8192      (32 KiB)
//   bin 4  n <= 27852                          1024 threads,        <= 32768      (128 KiB)
//   bin 5  n  > 27852   1024 threads, 32768 keys, row FAILS over to the global table when
//                       it holds more than 24576 distinct keys
//   (thresholds = 85 % of the largest table of the bin, and the table of a row is
//   pow2_ceil(1.5 n) up to that size: the reference fills a bin up to its table size, and a row
//   whose products hardly repeat -- power-law inputs -- then probes a full table.  Measured on
//   the webbase class: rows with 253 or 500 products in 256- and 512-key tables took 150-250 us
//   each and were the whole duration of their kernel, 0.22 ms; now 0.05 ms.)
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
    // numeric only: RANKED WINDOW rows (bin 9 of the numeric ladder).  The node-block kernel keeps nnz
    // accumulators addressed by bitmap rank, not one per column, so the width of the window only costs
    // bitmap words: rows with span <= rank_span, at most rank_max_nz non-zeros and a window that is not
    // absurdly sparse (span <= rank_ratio * n or span <= 4 * products) take it instead of hashing and
    // sorting.  0 disables.
    int rank_span, rank_ratio, rank_max_nz;
};
constexpr Thr kSymThr = {32,   {870, 1740, 6963, 27852}, {4096, 16384, 65536}, 8, {262144, 1048576}, 64, 2048,
                         8192, 16 * 1048576, 0, 0, 0};
constexpr Thr kNumThr = {16, {170, 682, 2730, 5461}, {1536, 4096, 12288}, 8, {0, 0}, 0, 0, 0, 0, 65536, 64, 4096};
constexpr int kRankBin = 9;  // numeric ladder only (the symbolic ladder's bins 9 / 10 are the bit windows)
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
constexpr int kGroupMembers = 2;  // twins of a leader that are recorded as its numeric group (block.h: 3 rows)
constexpr int kTwinEager = 8;     // rows this long claim their map slot / sign up without looking first (twin_probe)
constexpr int kPartialStride = 32;  // long longs per block: hist[NB], max, total, bm, alen, list (17 used)
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
    int max_span[NB];    // widest column window among the rows of each bin (sizes the LDS of the window kernels)
    int far_twins;       // twin rows more than two rows away from their pattern leader (k_numeric_setup)
    int ab_differ;       // the structure of B is not that of A (k_b_info, when asked to compare)
    // column lists of the bit-window rows (symbolic bins 9 / 10), handed to the numeric listed kernel:
    long long list_total;            // sum of min(products, window) over those rows: capacity of the list slab
    unsigned long long list_cursor;  // bump allocator of the symbolic kernels (one returning atomic per row)
    int queue_head3;                 // row queue of the listed numeric kernel
    int twin_sample;                 // some pattern occurs twice among the sampled rows of A (k_b_info; big matrices only)
};
static_assert(sizeof(BinState) <= 64 * sizeof(int), "k_publish and the fused tails copy one word per lane");

struct Stats {
    nsparse_spgemm_stats s;
    int fused_fallbacks = 0;  // calls repeated with the kernel chains after a grid barrier timed out
};
static Stats g_stats;

// `work` = products of the row.  Symbolic: the same number as n.  Numeric: n is the nnz of the C
// row, and a window is also worth its clearing and scanning when the row has many products for
// few non-zeros (FEM: 6561 products, 375 non-zeros, a window of 5 K columns once the mesh
// cross-section is 20 x 20 nodes): dense when span <= dense_ratio * n or span <= dense_ratio/4 * work.
__host__ __device__ __forceinline__ int bin_of(int n, int span, const Thr &thr, int work)
{
    if (n <= thr.tiny) return 0;
    if (thr.dense_ratio > 0 && span > 0 && span <= thr.dense_span[2] &&
        ((long long)span <= (long long)thr.dense_ratio * n ||
         (long long)span * 4 <= (long long)thr.dense_ratio * work))
        return kDenseBin0 + (span > thr.dense_span[0]) + (span > thr.dense_span[1]);
    if (thr.rank_span > 0 && span > 0 && span <= thr.rank_span && n <= thr.rank_max_nz &&
        ((long long)span <= (long long)thr.rank_ratio * n || (long long)span <= 4LL * work))
        return kRankBin;
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
    NSP_COUNT(FC_HASH_TB, 0, 1);
    while (true) {
        const int old = atomicCAS(tab + h, -1, key);
        NSP_COUNT(FC_HASH_TB, 1, 1);
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

// Every lane takes several consecutive entries of a B row per step: 16-byte column / value loads
// instead of scalar pairs, and the walk's bookkeeping (the kernels are instruction-issue-bound:
// rocprofv3 shows VALU 73 % / SALU 83 % busy, LDS 15 %) is paid once per vector.  The numeric
// walks take VW = 4 (columns + values: 12 registers per buffer); the symbolic dense-window kernel, which
// carries columns only and serves long regular rows, takes VWS = 8 (-16 % there; 8 on the numeric side
// costs occupancy and is 17 % slower, and in the hash symbolic kernels it pads the two- and
// three-entry B rows of power-law inputs: webbase class +12 %).
// LDS accumulators are double in both builds: ds_add_f32 runs at about one lane per clock on
// gfx950 (SQ_LDS_IDX_ACTIVE: 74 cycles per instruction against 12.5 for ds_add_f64 -- the fp32
// build of the numeric window kernel was 4x slower than the fp64 one), so the float build
// multiplies in float, converts, and adds in double.  Tables, tiles and ladders are therefore
// the fp64 ones for both precisions, and fp32 results carry less rounding noise than upstream's.
using acc_t = double;

constexpr int VW = 4;   // numeric walks and the hash symbolic kernels
constexpr int VWS = 8;  // symbolic dense-window kernel (k_sym_dense)
template <int V>
struct __attribute__((aligned(4))) IVecT {
    int v[V];
};
template <int V>
struct __attribute__((aligned(sizeof(real) < 8 ? 4 : 8))) RVecT {
    real v[V];
};
using IVec = IVecT<VW>;
using RVec = RVecT<VW>;
using IVecS = IVecT<VWS>;
using RVecS = RVecT<1>;  // symbolic walks carry no values

// Lanes per B row for a C row with `np` products spread over `alen` entries of A, for a
// workgroup of BS threads.  With g lanes per group the row takes
//     ceil(alen / (BS/g)) * ceil(avg_len / (g*VW))   group steps,
// so g trades padding of the B rows (small g pads less) against imbalance between groups
// (large g, few groups).  The largest g with the fewest steps wins.
__device__ __forceinline__ int group_width(int np, int alen, int BS, int maxb = 0, int vw = VW)
{
    if (alen <= 0) return 64;
    // (the average only steers the choice: a float reciprocal instead of an integer division's ~25 scalar steps)
    const int avg = (int)((float)np * __builtin_amdgcn_rcpf((float)alen) + 0.998f);  // ~ceil; exact multiples stay exact
    int best_g = 64, best_t = 0x7fffffff;
#pragma unroll
    for (int g = 64; g >= 1; g >>= 1) {
        const int ng = BS / g;
        int t = ((alen + ng - 1) / ng) * ((avg + g * vw - 1) / (g * vw));
        // the group that owns the longest B row of this C row (maxb entries) cannot finish
        // earlier than that row alone takes: on power-law inputs (hub rows of hundreds of entries
        // among rows of three) this term, not the average, decides
        const int tl = (maxb + g * vw - 1) / (g * vw);
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
template <bool WITH_VAL, int V>
__device__ __forceinline__ int fetch_chunk(const int *__restrict__ bcol, const real *__restrict__ bval,
                                           int i0, int ke, int bnnz, IVecT<V> &k, RVecT<WITH_VAL ? V : 1> &v)
{
    int n = ke - i0;
    n = n < 0 ? 0 : (n > V ? V : n);
    if (n > 0) {
        if (i0 + V <= bnnz) {
            // unsigned index: zero-extension is free, so the loads use base + 32-bit offset
            k = *reinterpret_cast<const IVecT<V> *>(bcol + (unsigned)i0);
            if (WITH_VAL) v = *reinterpret_cast<const RVecT<WITH_VAL ? V : 1> *>(bval + (unsigned)i0);
        } else {
#pragma unroll
            for (int i = 0; i < V; i++) {
                const int ii = i < n ? i0 + i : i0;
                k.v[i] = bcol[ii];
                if (WITH_VAL) v.v[WITH_VAL ? i : 0] = bval[ii];
            }
        }
    }
    return n;
}

// B rows parked by the group walk for a pass by the whole workgroup (walk_products_mixed)
// CAPX: a row whose products fit the bin's table cannot have more than table / 32 B rows of more than
// 32 entries, so the kernels size the list by their table (a fixed 32 was the whole duration of the
// webbase-1M-class symbolic phase: index pages that link to hundreds of directories left all but 32 of
// those long rows to single lanes, ~1000 dependent steps each)
template <bool WITH_VAL, int CAPX = 32>
struct DeferList {
    static constexpr int CAP = CAPX;
    int n;
    int2 ext[CAP];
    real av[WITH_VAL ? CAP : 1];
};

template <int BS, bool WITH_VAL, int V = VW, typename F, int DCAP = 32>
__device__ __forceinline__ void walk_products(const int *__restrict__ acol, const real *__restrict__ aval,
                                              const int *__restrict__ brpt, const int *__restrict__ bcol,
                                              const real *__restrict__ bval, int bnnz, int a_beg,
                                              int a_end, int g, int2 *s_ext, real *s_av, F &&consume,
                                              DeferList<WITH_VAL, DCAP> *dl = nullptr, int defer_len = 0x7fffffff,
                                              const unsigned char *skip_twin = nullptr, const int2 *pre_e = nullptr,
                                              const real *pre_av = nullptr)
{
    // pre_e / pre_av (k_num_wave): the lane's B extent and A value of the FIRST batch, fetched by the caller
    // while it was busy with the row before this one
    // skip_twin (symbolic walks only): rows of B flagged as twins of the row before them have that row's
    // columns; an A entry that points at such a row right after an entry that points at the row before it
    // adds no column and is not walked (the 3 dof of a mesh node: a third of the products remains).
    // g is a power of two (group_width, or 64): shifts, not the four integer divisions the compiler would emit
    static_assert((BS & (BS - 1)) == 0, "workgroup size is a power of two");
    const int lg = 31 - __clz(g), lng = (31 - __clz(BS)) - lg;
    const int ngroups = 1 << lng;
    const int gid = (int)threadIdx.x >> lg, gl = (int)threadIdx.x & (g - 1);
    const int first = a_beg + gid;
    const int cnt = first < a_end ? (a_end - first + ngroups - 1) >> lng : 0;
    int2 *ext = s_ext + gid * g;
    real *avs = s_av + gid * g;
    const int lane_off = gl * V;
    const int stride = g * V;
    for (int b0 = 0; b0 < cnt; b0 += g) {
        const int m = b0 + gl;
        int2 e = make_int2(0, 0);
        real av = 0;
        if (m < cnt) {
            if (pre_e && b0 == 0) {
                e = *pre_e;
                if (WITH_VAL) av = *pre_av;
            } else {
            const int j = first + m * ngroups;
            const int c = __builtin_nontemporal_load(acol + j);
            if (WITH_VAL) av = __builtin_nontemporal_load(aval + j);
            struct __attribute__((aligned(4))) I2 {
                int b, e;
            };
            const I2 r = *reinterpret_cast<const I2 *>(brpt + c);  // one 8-byte gather
            e.x = r.b;
            e.y = r.e;
            if (!WITH_VAL && skip_twin && j > a_beg && skip_twin[c] && acol[j - 1] == c - 1) e = make_int2(0, 0);
            }
            if (dl && e.y - e.x > defer_len) {  // far longer than the rows g was chosen for
                const int i = atomicAdd(&dl->n, 1);
                if (i < DCAP) {
                    dl->ext[i] = e;
                    if (WITH_VAL) dl->av[i] = av;
                    e = make_int2(0, 0);
                }
            }
        }
        ext[gl] = e;
        if (WITH_VAL) avs[gl] = av;
        wave_lds_sync();  // a group never spans wavefronts: in-order LDS is enough
        const int nb = cnt - b0 < g ? cnt - b0 : g;
        int t = 0;
        int2 cur = ext[0];
        real cav = WITH_VAL ? avs[0] : (real)0;
        int base = cur.x;
        IVecT<V> pk;
        RVecT<WITH_VAL ? V : 1> pv;
        int pn = fetch_chunk<WITH_VAL, V>(bcol, bval, base + lane_off, cur.y, bnnz, pk, pv);
        while (t < nb) {
            const IVecT<V> ck = pk;
            const RVecT<WITH_VAL ? V : 1> cv = pv;
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
            pn = t < nb ? fetch_chunk<WITH_VAL, V>(bcol, bval, base + lane_off, cur.y, bnnz, pk, pv) : 0;
            if (cn > 0) consume(ck, cv, cn, sc);
        }
        wave_lds_sync();
    }
}

__device__ __forceinline__ int wave_incl_scan(int v);

// Scratch of the flat walk below: the exclusive prefix of the chunk counts of one batch of A entries.
template <int BS>
struct FlatScratch {
    int pref[BS];
    int wsum[BS / 64];
};

// FLAT walk for rows whose B rows differ wildly in length (power-law inputs: hundreds of B rows of two or
// three entries and a few of hundreds or thousands).  Whatever group width the group walk picks, such a row
// is bound by its longest B row walked one chunk per round trip, or -- with the long rows parked for a pass of
// the whole workgroup, the round-2 form -- by one dependent round trip PER PARKED ROW (R-MAT-22, 8192-slot
// numeric bin: 36 of the 58 us of a row were this walk).  Here the products of a batch of BS entries of A are
// ONE sequence of V-element chunks: every thread parks one A entry and its B extent, a workgroup scan of the
// chunk counts gives every chunk its owner (binary search in LDS), thread t takes chunks t, t + BS, ... and
// requests U of them before it consumes the first.  Three dependent round trips per batch (A entry -> B
// extent -> B entries) whatever the lengths, and every lane busy.  Must be called by every thread.
template <int BS, bool WITH_VAL, typename F>
__device__ __forceinline__ void walk_products_flat(const int *__restrict__ acol, const real *__restrict__ aval,
                                                   const int *__restrict__ brpt, const int *__restrict__ bcol,
                                                   const real *__restrict__ bval, int bnnz, int a_beg, int a_end,
                                                   int2 *s_ext, real *s_av, FlatScratch<BS> *fs, F &&consume)
{
    constexpr int V = VW, NW = BS / 64, U = 4;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int b0 = a_beg; b0 < a_end; b0 += BS) {
        const int nb = a_end - b0 < BS ? a_end - b0 : BS;
        int2 e = make_int2(0, 0);
        real av = 0;
        if ((int)threadIdx.x < nb) {
            const int j = b0 + threadIdx.x;
            const int c = __builtin_nontemporal_load(acol + j);
            if (WITH_VAL) av = __builtin_nontemporal_load(aval + j);
            struct __attribute__((aligned(4))) I2 {
                int b, e;
            };
            const I2 r = *reinterpret_cast<const I2 *>(brpt + c);
            e.x = r.b;
            e.y = r.e;
        }
        const int nch = (e.y - e.x + V - 1) / V;
        s_ext[threadIdx.x] = e;
        if (WITH_VAL) s_av[threadIdx.x] = av;
        const int incl = wave_incl_scan(nch);
        if (lane == 63) fs->wsum[w] = incl;
        __syncthreads();
        int base = 0, total = 0;
#pragma unroll
        for (int u = 0; u < NW; u++) {
            const int c = fs->wsum[u];
            base += u < w ? c : 0;
            total += c;
        }
        fs->pref[threadIdx.x] = base + incl - nch;
        __syncthreads();
        for (int c0 = threadIdx.x; c0 < total; c0 += BS * U) {
            IVecT<V> pk[U];
            RVecT<WITH_VAL ? V : 1> pv[U];
            int pn[U];
            real sc[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int c = c0 + u * BS;
                pn[u] = 0;
                sc[u] = 0;
                if (c < total) {
                    int i = 0;  // the last entry whose first chunk is not beyond c
#pragma unroll
                    for (int step = BS / 2; step >= 1; step >>= 1) {
                        const int j = i + step;
                        if (j < nb && fs->pref[j] <= c) i = j;
                    }
                    const int2 x = s_ext[i];
                    if (WITH_VAL) sc[u] = s_av[i];
                    pn[u] = fetch_chunk<WITH_VAL, V>(bcol, bval, x.x + (c - fs->pref[i]) * V, x.y, bnnz, pk[u], pv[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++)
                if (pn[u] > 0) consume(pk[u], pv[u], pn[u], sc[u]);
        }
        __syncthreads();  // the next batch overwrites the parked entries
    }
}

// Rows of power-law matrices mix hundreds of B rows of two or three entries with one or two of
// hundreds: whatever single group width is chosen, either the short rows pad 64 lanes or the long
// row is walked 4 entries at a time by a narrow group (webbase class: 59 such rows held the whole
// symbolic phase for 0.26 ms).  Round 2: the width chosen for the rows without the longest one, B rows more
// than 8 steps long parked in LDS and walked by the whole workgroup afterwards, one after the other.  Round 3:
// such rows take the flat walk above (NSPARSE_FLAT=0 keeps the parked-rows form).  Must be called by every
// thread of the workgroup; dl->n zeroed and visible (barrier) beforehand.
template <int BS, bool WITH_VAL, typename F, int DCAP>
__device__ __forceinline__ void walk_products_mixed(const int *__restrict__ acol, const real *__restrict__ aval,
                                                    const int *__restrict__ brpt, const int *__restrict__ bcol,
                                                    const real *__restrict__ bval, int bnnz, int a_beg,
                                                    int a_end, int np, int maxb, int2 *s_ext, real *s_av,
                                                    DeferList<WITH_VAL, DCAP> *dl, F &&consume,
                                                    FlatScratch<BS> *fs = nullptr, bool flat_always = false)
{
    const int alen = a_end - a_beg;
    // longest row > 8 x the average (workgroup-uniform): width for the others, the long ones parked
    const bool mixed = alen > 1 && (long long)maxb * alen > 8LL * np;
    constexpr int V = VW;
    if ((mixed || flat_always) && fs != nullptr) {
        walk_products_flat<BS, WITH_VAL, F &>(acol, aval, brpt, bcol, bval, bnnz, a_beg, a_end, s_ext, s_av, fs, consume);
        return;
    }
    const int g = group_width(mixed ? np - maxb : np, mixed ? alen - 1 : alen, BS, mixed ? 0 : maxb, V);  // once
    walk_products<BS, WITH_VAL, V, F &, DCAP>(acol, aval, brpt, bcol, bval, bnnz, a_beg, a_end, g, s_ext, s_av, consume,
                                              mixed ? dl : (DeferList<WITH_VAL, DCAP> *)nullptr, 8 * g * V);
    if (!mixed) return;
    __syncthreads();
    const int nd = dl->n < DCAP ? dl->n : DCAP;
    for (int d = 0; d < nd; d++) {
        const int2 e = dl->ext[d];
        const real av = WITH_VAL ? dl->av[d] : (real)0;
        for (int base = e.x + (int)threadIdx.x * V; base < e.y; base += BS * V) {
            IVecT<V> k;
            RVecT<WITH_VAL ? V : 1> v;
            const int n = fetch_chunk<WITH_VAL, V>(bcol, bval, base, e.y, bnnz, k, v);
            if (n > 0) consume(k, v, n, av);
        }
    }
}

// one find-or-insert per element of the vector, the first probes issued back to back
template <int V>
__device__ __forceinline__ void ht_insert_vec(int *tab, int mask, const IVecT<V> &k, int n, int (&h)[V], int &fresh)
{
    int old[V];
#pragma unroll
    for (int i = 0; i < V; i++) h[i] = hash_slot(k.v[i], mask);
#pragma unroll
    for (int i = 0; i < V; i++) old[i] = i < n ? atomicCAS(tab + h[i], -1, k.v[i]) : k.v[i];
    // collisions of all V elements are resolved in ONE loop (its exit test is the only branch: V
    // separate probe loops cost V exec-mask save / restore sequences per step even when nothing collides)
    bool pend[V], any = false;
    NSP_COUNT(FC_HASH_TB, 0, n);
    NSP_COUNT(FC_HASH_TB, 1, n);
#pragma unroll
    for (int i = 0; i < V; i++) {
        pend[i] = old[i] != -1 && old[i] != k.v[i];
        any |= pend[i];
        if (pend[i]) NSP_COUNT(FC_HASH_TB, 2, 1);
    }
    while (any) {
        any = false;
#pragma unroll
        for (int i = 0; i < V; i++) {
            if (pend[i]) {
                h[i] = (h[i] + 1) & mask;
                NSP_COUNT(FC_HASH_TB, 1, 1);
                old[i] = atomicCAS(tab + h[i], -1, k.v[i]);
                pend[i] = old[i] != -1 && old[i] != k.v[i];
                any |= pend[i];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < V; i++) fresh += old[i] == -1;
}


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
#ifndef NSP_EMU
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
    __syncthreads();  // tests/emu: the CPU emulation knows one workgroup barrier
#endif
}

// ---- column lists (round 3) -----------------------------------------------------------------------------
// The bit-window symbolic kernels hold the exact column set of (a piece of) a C row as a bitmap in LDS.  Until
// round 2 they only counted it; now they also WRITE it out, ascending, into a slab (`tcol`, room for
// min(products, window) entries per row, bump-allocated with one returning atomic per row), and the numeric
// phase accumulates such a row against its list (listed.h) instead of finding the structure a second time:
// no bitmap tiles whose number grows with the WIDTH of the window (R-MAT-22: 8.6 tiles of 2^19 columns per
// 16 K-entry row, each two cursor walks, a scan and an emission), no hash probing, no sort.
//
// Which rows get a list: more non-zeros than the largest LDS hash table takes (n = the exact count where the
// kernel knows it, else the bound min(products, window)), at most kListMaxSlices slices of kListSlice entries,
// and slices x products -- what the listed kernel reads -- within `work_max`: a hub row with millions of
// products stays with the cursor kernels, which see every product once.
constexpr int kListSlice = 10240;
constexpr int kListMaxSlices = 4;
constexpr int kListMinNnz = 5461;  // = kNumThr.hash_t[3]
__host__ __device__ __forceinline__ bool list_wanted(int n, int products, long long work_max)
{
    const int S = (n + kListSlice - 1) / kListSlice;
    return n > kListMinNnz && S <= kListMaxSlices && (long long)S * products <= work_max;
}

// bits: `words` bitmap words in LDS.  Returns the number of set bits (uniform).  dst != nullptr: the columns
// col0 + (bit index) leave in ascending order to dst[0 .. count).  CLEAR: the words are zeroed on the way.
// Wavefront w owns a contiguous range of 64-word blocks; inside a block lane l holds word l, so the lanes' runs
// of output are ADJACENT in memory and a store instruction of a sparse stretch covers one or two cache lines
// (the first version gave every thread 16-32 consecutive words: 64 far-apart runs per store instruction, and
// the symbolic bit-window kernel of R-MAT-18 went from 6.8 to 12.7 ms).  Two workgroup barriers; s_wsum: BS / 64 ints.
template <int BS, bool CLEAR>
__device__ __forceinline__ int bits_to_list(unsigned int *bits, int words, int col0, int *__restrict__ dst, int *s_wsum,
                                            bool dry = false)
{
    // (Staging the columns in LDS so that a store instruction writes 64 consecutive entries was tried: the symbolic
    //  cursor kernel of R-MAT-22 took 16.9 ms with it as without; with the stores left out altogether -- `dry` --
    //  11.1, without lists 9.4.  The stores cost by their volume and their acknowledgements, which the next tile's
    //  first load waits for, not by their shape.)
    // (blocks of 128 words, lane l holds the 64-bit pair 2l, 2l + 1: one wave scan per 4096 columns)
    constexpr int NW = BS / 64;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nblk = (words + 127) >> 7;
    const int per = (nblk + NW - 1) / NW;
    const int b0 = wv * per, b1 = b0 + per < nblk ? b0 + per : nblk;
    auto pair_at = [&](int w) -> unsigned long long {  // words w, w + 1 (w even), zero beyond the bitmap
        const unsigned long long lo = w < words ? bits[w] : 0u, hi = w + 1 < words ? bits[w + 1] : 0u;
        return lo | (hi << 32);
    };
    int mine = 0;
    for (int b = b0; b < b1; b++) mine += __popcll(pair_at(b * 128 + 2 * lane));
    mine = wave_sum(mine);
    if (lane == 0) s_wsum[wv] = mine;
    lds_barrier();  // (LDS only: a __syncthreads() would also wait for the list stores of the previous piece)
    int base = 0, total = 0;
#pragma unroll
    for (int u = 0; u < NW; u++) {
        const int c = s_wsum[u];
        base += u < wv ? c : 0;
        total += c;
    }
    if (dst != nullptr || CLEAR) {
        int running = base;
        for (int b = b0; b < b1; b++) {
            const int w = b * 128 + 2 * lane;
            unsigned long long m = pair_at(w);
            if (CLEAR) {
                if (w < words) bits[w] = 0u;
                if (w + 1 < words) bits[w + 1] = 0u;
            }
            if (dst != nullptr) {
                const int c = __popcll(m);
                const int incl = wave_incl_scan(c);
                int p = running + incl - c;
                const int cb = col0 + 32 * w;
                while (m && !dry) {  // (dry: diagnostics, everything but the stores)
                    dst[p++] = cb + __builtin_ctzll(m);  // (nontemporal stores: 16.9 -> 22.6 ms)
                    m &= m - 1;
                }
                running += __builtin_amdgcn_readlane(incl, 63);
            }
        }
    }
    lds_barrier();  // not __syncthreads(): nobody in the workgroup reads the list back, why wait for its stores
    return total;
}

// Value of lane (l ^ J): upper / lower half exchange by v_permlane32_swap (VALU), smaller
// distances by ds_swizzle (LDS crossbar, no memory access).
template <int J>
__device__ __forceinline__ int lane_xor(int v)
{
    if constexpr (J == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap((unsigned int)v, (unsigned int)v, false, false);
        return (threadIdx.x & 32) ? (int)r[0] : (int)r[1];
    } else {
        return __builtin_amdgcn_ds_swizzle(v, (J << 10) | 0x1f);  // bit-mask mode: and 0x1f, xor J
    }
}

// The compare-exchange stages at partner distances J, J/2, ..., 1 of merge size k on the 128
// elements of one wavefront's segment, held two per lane: r0 = element e0 = seg*128 + lane,
// r1 = element e0 + 64.
template <int J>
__device__ __forceinline__ void bitonic_stages_reg(int &r0, int &r1, int e0, int k, int lane)
{
    if constexpr (J == 64) {
        const bool up = (e0 & k) == 0;
        const int lo = r0 < r1 ? r0 : r1, hi = r0 < r1 ? r1 : r0;
        r0 = up ? lo : hi;
        r1 = up ? hi : lo;
    } else {
        const bool lower = (lane & J) == 0;
        const int q0 = lane_xor<J>(r0), q1 = lane_xor<J>(r1);
        const bool min0 = lower == ((e0 & k) == 0), min1 = lower == (((e0 + 64) & k) == 0);
        r0 = min0 ? (r0 < q0 ? r0 : q0) : (r0 < q0 ? q0 : r0);
        r1 = min1 ? (r1 < q1 ? r1 : q1) : (r1 < q1 ? q1 : r1);
    }
    if constexpr (J > 1) bitonic_stages_reg<J / 2>(r0, r1, e0, k, lane);
}

// The same for big sorts: 512 elements per wavefront, eight consecutive ones per lane
// (element = seg*512 + lane*8 + i).  Distances 4, 2, 1 are inside the thread, 8 ... 256 are lane
// exchanges (lane ^ J/8): only distances of 512 and more cross wavefronts.
template <int J>
__device__ __forceinline__ void bitonic_stages_reg8(int (&r)[8], int ebase, int k, int lane)
{
    if constexpr (J >= 8) {
        constexpr int L = J / 8;
        const bool lower = (lane & L) == 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int q = lane_xor<L>(r[i]);
            const bool mn = lower == (((ebase + i) & k) == 0);
            r[i] = mn ? (r[i] < q ? r[i] : q) : (r[i] < q ? q : r[i]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if ((i & J) == 0) {
                const bool up = ((ebase + i) & k) == 0;
                const int a = r[i], b = r[i | J];
                const int lo = a < b ? a : b, hi = a < b ? b : a;
                r[i] = up ? lo : hi;
                r[i | J] = up ? hi : lo;
            }
        }
    }
    if constexpr (J > 1) bitonic_stages_reg8<J / 2>(r, ebase, k, lane);
}

// In-LDS bitonic sort of P (power of two) ints, ascending.  Stages whose partner distance is below
// 128 stay inside one wavefront's 128-element segment: the segment is taken into registers (two
// elements per lane), the stages run on lane exchanges, and it is written back -- the first
// version did every one of them as two LDS reads and two conditional LDS writes.  Only the wider
// stages go through LDS with a workgroup barrier.
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
    if (P >= 1024) {  // big sorts: 512-element register segments, LDS only for distances >= 512
        auto load8 = [&](int ebase, int(&r)[8]) {
            const int4 a = *reinterpret_cast<const int4 *>(s + ebase), b = *reinterpret_cast<const int4 *>(s + ebase + 4);
            r[0] = a.x, r[1] = a.y, r[2] = a.z, r[3] = a.w, r[4] = b.x, r[5] = b.y, r[6] = b.z, r[7] = b.w;
        };
        auto store8 = [&](int ebase, const int(&r)[8]) {
            *reinterpret_cast<int4 *>(s + ebase) = make_int4(r[0], r[1], r[2], r[3]);
            *reinterpret_cast<int4 *>(s + ebase + 4) = make_int4(r[4], r[5], r[6], r[7]);
        };
        for (int seg = wid; seg * 512 < P; seg += NW) {
            const int ebase = seg * 512 + lane * 8;
            int r[8];
            load8(ebase, r);
            bitonic_stages_reg8<1>(r, ebase, 2, lane);
            bitonic_stages_reg8<2>(r, ebase, 4, lane);
            bitonic_stages_reg8<4>(r, ebase, 8, lane);
            bitonic_stages_reg8<8>(r, ebase, 16, lane);
            bitonic_stages_reg8<16>(r, ebase, 32, lane);
            bitonic_stages_reg8<32>(r, ebase, 64, lane);
            bitonic_stages_reg8<64>(r, ebase, 128, lane);
            bitonic_stages_reg8<128>(r, ebase, 256, lane);
            bitonic_stages_reg8<256>(r, ebase, 512, lane);
            store8(ebase, r);
        }
        __syncthreads();
        for (int k = 1024; k <= P; k <<= 1) {
            for (int j = k >> 1; j >= 512; j >>= 1) {
                for (int t = threadIdx.x; t < P / 2; t += BS) cex(t, j, k);
                __syncthreads();
            }
            for (int seg = wid; seg * 512 < P; seg += NW) {
                const int ebase = seg * 512 + lane * 8;
                int r[8];
                load8(ebase, r);
                bitonic_stages_reg8<256>(r, ebase, k, lane);
                store8(ebase, r);
            }
            __syncthreads();
        }
        return;
    }
    // phase 1: every merge size up to 128 stays inside a 128-element segment
    for (int seg = wid; seg * 128 < P; seg += NW) {
        const int e0 = seg * 128 + lane;
        int r0 = e0 < P ? s[e0] : 0x7fffffff, r1 = e0 + 64 < P ? s[e0 + 64] : 0x7fffffff;
        if (P >= 2) bitonic_stages_reg<1>(r0, r1, e0, 2, lane);
        if (P >= 4) bitonic_stages_reg<2>(r0, r1, e0, 4, lane);
        if (P >= 8) bitonic_stages_reg<4>(r0, r1, e0, 8, lane);
        if (P >= 16) bitonic_stages_reg<8>(r0, r1, e0, 16, lane);
        if (P >= 32) bitonic_stages_reg<16>(r0, r1, e0, 32, lane);
        if (P >= 64) bitonic_stages_reg<32>(r0, r1, e0, 64, lane);
        if (P >= 128) bitonic_stages_reg<64>(r0, r1, e0, 128, lane);
        if (e0 < P) s[e0] = r0;
        if (e0 + 64 < P) s[e0 + 64] = r1;
    }
    __syncthreads();
    // phase 2: wide stages with workgroup barriers, then the sub-segment tail of each merge
    for (int k = 256; k <= P; k <<= 1) {
        for (int j = k >> 1; j >= 128; j >>= 1) {
            for (int t = threadIdx.x; t < P / 2; t += BS) cex(t, j, k);
            __syncthreads();
        }
        for (int seg = wid; seg * 128 < P; seg += NW) {
            const int e0 = seg * 128 + lane;
            int r0 = s[e0], r1 = s[e0 + 64];
            bitonic_stages_reg<64>(r0, r1, e0, k, lane);
            s[e0] = r0;
            s[e0 + 64] = r1;
        }
        __syncthreads();
    }
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

}  // namespace spgemm
}  // namespace nsp
