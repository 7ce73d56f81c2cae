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
// Source map (one translation unit; this file holds the host pipeline):
//   spgemm/common.h        bin ladders, BinState, hash probe, product walk, wave helpers
//   spgemm/setup.h         k_b_info, k_row_products (+ twin_probe), k_reduce_partials, k_hist, k_bin_scatter,
//                          k_publish, k_ab_compare, k_finish
//   spgemm/fused.h         k_setup_tail, k_numeric_setup                  (the helper chains as one launch, M < 1 M)
//   spgemm/symbolic.h      k_sym_small, k_sym_tb, k_sym_global            (bins 0-5)
//   spgemm/numeric.h       k_num_small, k_num_tb, k_num_global            (bins 0-4, fallback)
//   spgemm/window.h        k_sym_dense, k_sym_bits, k_num_dense           (bins 6-10)
//   spgemm/block.h         k_num_block, k_twin_groups                     (numeric bins 6-9: node blocks)
//   spgemm/heavy_tiled.h   k_num_tiled                                    (bin 5, dense tiles)
//   spgemm/heavy_ranked.h  k_num_ranked                                   (bin 5, thin rows)
//   spgemm/heavy_flat.h    k_panel_slots, k_panel_fill, k_num_flat, k_num_ranked_flat, k_sym_flat  (stateless tiles; experiments build)
//   spgemm/lean.h          k_sym_lean, k_num_lean                         (hash bins 1-4 on an instruction diet, round 4)
//
// Results: C.rpt / C.col are bit-identical to the reference by construction (distinct
// columns per row, ascending); C.val differs only by floating-point summation order
// (LDS atomics), same as the reference vs cuSPARSE (tolerance 1e-9 double / 1e-6 float,
// nsparse.cu:300-353).
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>

#include <algorithm>
#include <cstring>
#include <vector>

#include "internal.h"
#include "spgemm/common.h"
#include "spgemm/setup.h"
#include "spgemm/symbolic.h"
#include "spgemm/numeric.h"
#include "spgemm/window.h"
#include "spgemm/block.h"
#include "spgemm/fused.h"
#include "spgemm/heavy_tiled.h"
#include "spgemm/heavy_ranked.h"
#include "spgemm/heavy_flat.h"
#include "spgemm/lean.h"

namespace nsp {
namespace spgemm {

// ===================================================================================
//  host orchestration
// ===================================================================================

// (exp_env: internal.h -- switches of the measurements, read only in a -DNSPARSE_EXPERIMENTS build)

// Numeric ladder in force.  NSPARSE_NUM_HEAVY_MIN=<n> (experiments): rows with more than n non-zeros go
// to the heavy bin (cursor kernels, no sort) instead of the LDS hash bins above that size.
static const Thr &num_ladder()
{
    static Thr t = [] {
        Thr v = kNumThr;
        if (exp_env("NSPARSE_WINDOW_KERNEL", 1) == 0) v.rank_span = 0;  // needs the block kernel
        const int n = exp_env("NSPARSE_NUM_HEAVY_MIN", 0);
        if (n > 0)
            for (int q = 0; q < 4; q++)
                if (v.hash_t[q] > n) v.hash_t[q] = n;
        return v;
    }();
    return t;
}

// Symbolic ladder in force.  Bin 1 takes rows of up to 870 products with a 1024-slot table (round 3): a one-wavefront
// row of that instance is 5 KB of LDS, so a CU still holds 32 of them, where bin 2 (two wavefronts, 2048 slots) holds
// 15.  NSPARSE_SYM1_T=512 restores the 512-slot bin (rows of up to 435 products).
static const Thr &sym_ladder()
{
    static Thr t = [] {
        Thr v = kSymThr;
        if (exp_env("NSPARSE_SYM1_T", 1024) == 512) v.hash_t[0] = 435;
        return v;
    }();
    return t;
}

// (wave-cooperative probing -- every active lane inspects a slot of ONE key's probe sequence, ballot, one CAS -- is
//  the alternative BASELINE's north_star names.  Built in round 2 and measured 1.2-15x slower than one lane per key at
//  load factor <= 2/3, profiles/r02_coop_probe.jsonl; removed in round 4.)
// The flat product walk (common.h: walk_products_flat) in the big-table hash bins (numeric 3 / 4, symbolic 3 / 4).
// 2 (default): every row of those bins; 1: only rows whose longest B row is more than 8x their average (the
// round-2 criterion for parking long rows); 0: the round-2 walk (group walk + parked long rows).  R-MAT-22
// 86.5 / 86.3 / 72.8 ms for 0 / 1 / 2 -- its rows are all hubs, so their AVERAGE B row is long as well --
// webbase-1M class 2.71 / 2.46 / 2.42.
static const int g_flat = exp_env("NSPARSE_FLAT", 2);
// round 4: the lean kernels of the hash bins 1..4 (lean.h), experiments build only until they have been timed on the
// device.  NSPARSE_TB_LEAN bit 0: symbolic, bit 1: numeric; bits 2 / 3 pick the form inside them (branch-free retry
// rounds / pipelined walk) -- all four forms are template instantiations of ONE library, so the A/B costs one build
static int g_deterministic = 0;  // nsparse_set_deterministic
static const int g_tb_lean = exp_env("NSPARSE_TB_LEAN", 0);  // (off until the fixed 24-bit hash has been measured)

// Column lists (common.h: list_wanted): a heavy row goes to the listed kernel while slices x products stays within
// this (NSPARSE_LIST_WORK); beyond it the cursor kernels, which see every product once, are cheaper.
// NSPARSE_LIST: 0 no column lists; 1 (default) lists from the symbolic CURSOR kernel -- matrices wider than
// 2^20 columns, where the ranked numeric tiles are bound by the width of their bitmap -- for the list-driven tiles
// of the ranked kernel (heavy_ranked.h): R-MAT-22 numeric heavy bin 41 -> 30 ms for 5-7 ms more in the symbolic
// kernel.  (Round 3 also had lists from the one-piece bit-window kernel and a flat "listed" numeric kernel for the
// rows with one: R-MAT-18 a wash, R-MAT-22 no gain; removed in round 4, measurements in DESIGN 4.1.)
static int list_mode()
{
    static const int m = getenv("NSPARSE_LIST") ? (atoi(getenv("NSPARSE_LIST")) != 0) : 1;
    return m;
}

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

// Lanes per row for the set-up kernels when the host knows the longest row (nnz_max > 0) and no row
// would be deferred: a quarter of the average length, to the nearest power of two -- every lane then
// runs the four-gathers-in-flight loop once or twice instead of one gather and a 6-step shuffle
// reduction for one entry (k_row_products on the cant class: 52 -> 17 us with 16 lanes instead of
// 64; 27-point stencil 8 instead of 32).
static inline int pick_w_regular(long long nnz, int M, int nnz_max)
{
    // long rows possible (or unknown): half the average, at least 2 lanes when rows average more than
    // one entry -- the long rows go to the 64-lane pass anyway, and the short ones are bound by the
    // dependent loads of a row, not by lanes (webbase class: k_row_products 75 -> 45 us with 2 lanes
    // instead of 4)
    const int w_full = pick_w(nnz, M), w_half = pick_w((nnz + 1) / 2, M);
    const int w_two = w_full < 2 ? w_full : 2;
    const int w_skew = w_half > w_two ? w_half : w_two;
    if (nnz_max <= 0 || M <= 0) return w_skew;
    // no row will be deferred: a quarter of the average, to the nearest power of two
    const double q = (double)nnz / M / 4.0;
    int w = 1;
    while (w < 64 && (double)w * 1.4142 < q) w <<= 1;
    return nnz_max <= kLongFactor * w ? w : w_skew;
}


// returns the number of per-workgroup partial records; reduce = false leaves them for k_setup_tail
static int launch_row_products(const sfCSR *a, const sfCSR *b, const BInfo *binfo,
                                int *row_prod, int *row_lo, int *row_span, int *bm_words,
                                int bm_span_max, const Thr &thr, BinState *d_bs, long long *partial,
                                int *row_span_num, int *row_nz, int *row_maxb, int *long_list,
                                int *long_cnt, TwinMap tw, bool reduce, hipStream_t st)
{
    const int M = a->M;
    static const int w_env = exp_env("NSPARSE_RP_W", 0);  // lanes per row, forced
    const int w = (w_env == 1 || w_env == 2 || w_env == 4 || w_env == 8 || w_env == 16 || w_env == 32 || w_env == 64)
                      ? w_env : pick_w_regular(a->nnz, M, a->nnz_max);
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
                           kLongFactor * W, no_todo, tw);                                      \
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
                           (int *)nullptr, long_cnt, 0, (const int *)long_list, tw);
        grid += 256;
    }
    if (reduce)
        hipLaunchKernelGGL(k_reduce_partials, dim3(grid < 32 ? grid : 32), dim3(256), 0, st, partial, grid, d_bs);
    NSP_LAUNCH_CHECK();
    return grid;
}

// Phase times of the statistics (ms_setup / ms_symbolic / ms_numeric / ms_total): four event records per call,
// only when per-bin timing or profiling is switched on -- an event record is a couple of microseconds of
// host time and a packet on the stream, on a call of 335.
struct Timer {
    Context &cx;
    bool on;
    explicit Timer(Context &c) : cx(c), on(c.bin_timing || c.profiling) {}
    void mark(int i, hipStream_t st)
    {
        if (on) NSP_CHECK(hipEventRecord(cx.ev_t[i], st));
    }
    float ms(int i, int j)
    {
        float v = 0;
        if (!on) return v;
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
    bool timed;      // begin / end events per bin (nsparse_set_bin_timing, or profiling mode)
    int main_bin;    // the bin with the most rows runs on the main stream itself (no fork/join)
    bool side_bins = false;  // some other bin has rows: only then is the fork event worth recording
    void *deferred[8] = {};  // scratch of kernels still in flight: returned to the cache by collect()
    int ndeferred = 0;
    void free_later(void *p) { deferred[ndeferred++] = p; }
    int most_bin = -1;  // the bin with the most rows
    // big_a, big_b: bins whose workgroups want a whole CU each (LDS).  When one of them has a few rows
    // it takes the main stream and is launched first: on a side stream its launch waits for the fork
    // event while the million small rows of the main bin already fill every CU, and the big
    // workgroups then start only when that bin has drained (webbase class: symbolic phase 0.15 or
    // 0.23 ms from one call to the next).
    BinLauncher(Context &c, int phase, const int *hist = nullptr, int big_a = -1, int big_b = -1)
        : cx(&c), ev(c.ev_bin + phase * 2 * NB), serial(c.profiling), timed(c.profiling || c.bin_timing), main_bin(-1)
    {
        static const bool big_main = exp_env("NSPARSE_BIG_MAIN", 1) != 0;
        if (hist) {
            int best = 0;
            for (int b = 0; b < NB; b++)
                if (hist[b] > best) { best = hist[b]; most_bin = b; }
            main_bin = most_bin;
            // (a big bin with more rows than CUs keeps every CU for milliseconds anyway: R-MAT is 3 %
            //  slower with it on the main stream)
            if (big_main && big_a >= 0 && hist[big_a] > 0 && hist[big_a] <= 256) main_bin = big_a;
            else if (big_main && big_b >= 0 && hist[big_b] > 0 && hist[big_b] <= 256) main_bin = big_b;
            for (int b = 0; b < NB; b++) side_bins |= hist[b] > 0 && b != main_bin;
        } else {
            side_bins = true;
        }
    }
    // bin b runs on stream[b], the main bin on stream[0] -- which IS bin 0's own stream: when a big-LDS bin of a few
    // rows is the main one, bin 0's kernel queues behind it (webbase-1M class: the 549 K rows of k_num_small wait
    // 1.46 ms for the heavy kernel and then run alone for 0.15 ms).  NSPARSE_BIN0_SWAP=1 gives bin 0 the main bin's
    // stream instead.  Measured on that matrix, same box, three runs each: 2.40 / 2.40 / 2.41 ms as is, 2.52 / 2.46 /
    // 2.46 swapped -- the call is bound by the dependent-load chain of its longest heavy row (0.36 ms alone, 1.46 ms
    // beside the other bins: memory latency under load), and a million more small rows in flight stretch it
    // further.  So the queueing stays.
    hipStream_t stream_of(int b) const
    {
        static const bool swap0 = exp_env("NSPARSE_BIN0_SWAP", 0) == 1;
        if (serial || b == main_bin) return cx->stream[0];
        if (b == 0 && main_bin > 0 && swap0) return cx->stream[main_bin];
        return cx->stream[b];
    }
    void fork()
    {
        if (!serial && side_bins) NSP_CHECK(hipEventRecord(cx->ev_fork, cx->stream[0]));
    }
    // The begin/end events sit on the stream the bin's kernels are launched on, so their
    // difference is the duration of those kernels whether or not other bins overlap.
    hipStream_t begin(int b)
    {
        hipStream_t st = stream_of(b);
        if (st != cx->stream[0] && !used[b]) NSP_CHECK(hipStreamWaitEvent(st, cx->ev_fork, 0));
        used[b] = true;
        if (timed) NSP_CHECK(hipEventRecord(ev[2 * b], st));
        return st;
    }
    void end(int b)
    {
        if (timed) NSP_CHECK(hipEventRecord(ev[2 * b + 1], stream_of(b)));
    }
    void join()
    {
        for (int b = 0; b < NB; b++) {
            if (!used[b] || stream_of(b) == cx->stream[0]) continue;
            NSP_CHECK(hipEventRecord(cx->ev_join[b], stream_of(b)));
            NSP_CHECK(hipStreamWaitEvent(cx->stream[0], cx->ev_join[b], 0));
        }
    }
    void collect(float *out)  // call after the device is idle
    {
        for (int i = 0; i < ndeferred; i++) dev_free(deferred[i]);
        ndeferred = 0;
        for (int b = 0; b < NB; b++) {
            out[b] = 0;
            if (used[b] && timed) {
                // the flag poll proves the GPU is done, not that the runtime has retired the event
                NSP_CHECK(hipEventSynchronize(ev[2 * b + 1]));
                NSP_CHECK(hipEventElapsedTime(&out[b], ev[2 * b], ev[2 * b + 1]));
            }
        }
    }
};

// Dynamic LDS above 64 KiB has to be allowed per kernel function, once -- per DEVICE: the attribute belongs to the
// function as loaded on the device that is current (one process driving several GPUs, nsparse_dist_init_all, sets it
// on each).  One bit per device of the per-device tables; callers hold the API lock.
struct DevOnce {
    unsigned long long w[2] = {0, 0};
    bool test_and_set(int d)
    {
        const unsigned long long bit = 1ULL << (d & 63);
        const bool was = (w[(d >> 6) & 1] & bit) != 0;
        w[(d >> 6) & 1] |= bit;
        return was;
    }
};
template <typename K>
static void allow_big_lds(K kernel, DevOnce &done, int bytes_max, int device)
{
    if (done.test_and_set(device)) return;
    NSP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  bytes_max));
}

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

// Hash bins with a handful of rows (webbase class: 81 and 7 rows) are folded into the next non-empty
// hash bin: its kernel sizes the table per row, so it takes the shorter rows as they are, and a bin
// less is a stream fork / join less (~14 us of host time each).  Bins are contiguous in row_perm, so
// the receiving launch simply starts earlier.  NSPARSE_FOLD=0 switches it off.
static void fold_small_hash_bins(const int *hist_in, int *hist, int *off)
{
    static const int fold_max = exp_env("NSPARSE_FOLD", 128);
    off[0] = 0;
    for (int q = 0; q < NB; q++) {
        hist[q] = hist_in[q];
        off[q + 1] = off[q] + hist_in[q];
    }
    for (int q = 1; q < 4; q++) {  // hash bins 1..4
        if (hist[q] == 0 || hist[q] > fold_max) continue;
        int q2 = q + 1;
        while (q2 <= 4 && hist[q2] == 0) q2++;
        if (q2 > 4) break;
        off[q2] = off[q];
        hist[q2] += hist[q];
        hist[q] = 0;
    }
}

// Panel table of B for the stateless heavy-row kernels (heavy_flat.h; experiments build): built at most once per call by
// the first phase that wants it -- the symbolic phase (k_sym_flat) or the numeric one -- on that phase's heavy-bin
// stream, freed with the symbolic / numeric launcher's deferred scratch when the call has drained.  (One call at a time:
// ApiLock.)
struct PanelTable {
    int *slot_of = nullptr, *tab = nullptr;
    int np = 0;
};
static PanelTable g_panel;
constexpr int kPanelW = 12288;  // = the dense tile width (numeric_phase: kTileW)

// table geometry: panels, the shortest row that gets a table row (-1: every row), upper bound of the table rows
static void panel_table_shape(const sfCSR *b, int &np, int &min_len, long long &slots_max)
{
    np = (int)(((long long)b->N + kPanelW - 1) / kPanelW);
    // every row of B in the table while that stays small next to B; else only rows of more than min_len entries,
    // min_len from a worst-case budget of 1 GiB: at most nnz / (min_len + 1) rows are that long, and that bound
    // sizes the table (no round trip to the host; the fill kernel touches only the rows that exist).  A row
    // outside the table is walked whole by every tile of a C row and filtered by column: on R-MAT-22 at a fifth
    // of config 5's edges min_len = 16 cost a third of the products again in such re-reads (tests/emu census).
    const bool all_rows = (long long)b->M * (np + 1) * 4 <= (256LL << 20);
    const long long per_row = (long long)(np + 1) * 4;
    min_len = all_rows ? -1 : (int)std::min<long long>(64, ((long long)b->nnz * per_row + (1LL << 30) - 1) / (1LL << 30) - 1);
    if (!all_rows && min_len < 1) min_len = 1;
    slots_max = all_rows ? (long long)b->M : std::min<long long>(b->M, (long long)b->nnz / (min_len + 1) + 1);
}
// (the kernels address a table row as an int offset: 2^26 ints with every row in, 2^28 under the 1 GiB budget)
static bool panel_table_fits(const sfCSR *b)
{
    int np, min_len;
    long long slots_max;
    panel_table_shape(b, np, min_len, slots_max);
    return slots_max * (np + 1) <= 0x7fffffffLL;
}

static void build_panel_table(const sfCSR *b, BinLauncher &L, hipStream_t st)
{
    if constexpr (kExperiments) {
        if (g_panel.tab != nullptr) return;
        const int *brpt = b->d_rpt, *bcol = b->d_col;
        if (!panel_table_fits(b)) return;  // no table: the callers keep the cursor kernels
        int np, min_len;
        long long slots_max;
        panel_table_shape(b, np, min_len, slots_max);
        const bool all_rows = min_len < 0;
        // one block: [count, pad | slot_of: M | slot_row: slots_max | tab: slots_max * (np + 1) | flag: M + 1 | pos: M + 1]
        const size_t n_tab = (size_t)slots_max * (np + 1);
        const size_t n_ints = 2 + (size_t)b->M + (size_t)slots_max + n_tab + 2 + (all_rows ? 0 : 2 * ((size_t)b->M + 1));
        int *blk = (int *)dev_alloc(sizeof(int) * n_ints);
        int *d_cnt = blk, *slot_of = blk + 2, *slot_row = slot_of + b->M, *tab = slot_row + slots_max;
        if (all_rows) {
            hipLaunchKernelGGL(k_panel_slots<true>, dim3(ceil_div(b->M, 256)), dim3(256), 0, st, brpt, b->M, min_len,
                               slot_of, slot_row, d_cnt, (int)slots_max, (const int *)nullptr);
        } else {
            int *flag = tab + n_tab + 2, *pos = flag + b->M + 1;
            hipLaunchKernelGGL(k_panel_flags<0>, dim3(ceil_div(b->M + 1, 256)), dim3(256), 0, st, brpt, b->M, min_len, flag);
            void *scan_tmp = scan_exclusive(flag, pos, b->M + 1, st);
            hipLaunchKernelGGL(k_panel_slots<false>, dim3(ceil_div(b->M, 256)), dim3(256), 0, st, brpt, b->M, min_len,
                               slot_of, slot_row, d_cnt, (int)slots_max, (const int *)pos);
            L.free_later(scan_tmp);  // (rocprim's scratch: back to the cache when the call has drained)
        }
        const long long cells = slots_max * (np + 1);
        hipLaunchKernelGGL(k_panel_fill<kPanelW>, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, st, brpt, bcol,
                           (const int *)slot_row, (const int *)d_cnt, np, tab);
        NSP_LAUNCH_CHECK();
        L.free_later(blk);
        g_panel.slot_of = slot_of, g_panel.tab = tab, g_panel.np = np;
    }
}

static BinLauncher symbolic_phase(const sfCSR *a, const sfCSR *b, int *row_prod, const int *row_maxb,
                                  const int *row_lo,
                                  const int *row_span, int *row_nz, int *row_perm, const int *hist_in,
                                  int max_prod, BinState *d_bs, Context &cx, float *ms_bin,
                                  int *fail_rows, const int *bm_off, unsigned int *bm,
                                  int *row_span_num, const int *max_span, int max_alen, bool b_sorted,
                                  const unsigned char *btwin, const int4 *sdesc, int *tcol, long long *list_off)
{
    int hist[NB], off[NB + 1];
    fold_small_hash_bins(hist_in, hist, off);
    BinLauncher L(cx, 0, hist, 10, 4);
    const int *arpt = a->d_rpt, *acol = a->d_col, *brpt = b->d_rpt, *bcol = b->d_col;
    *fail_rows = 0;
    int *fail_list = nullptr;
    L.fork();
#define NSP_SYM_TB_GO(BS, TMAX)                                                                 \
    hipLaunchKernelGGL((k_sym_tb<BS, TMAX, false>), dim3(8 * ceil_div(hist[bin_], 8)), dim3(BS), 0, \
                       st, arpt, acol, brpt, bcol, row_perm, row_prod, row_maxb, row_nz, off[bin_], \
                       hist[bin_], b->nnz, d_bs, (int *)nullptr, g_flat, TMAX >= 8192 ? tcol : (int *)nullptr, list_off,   \
                       row_span, 12, 12288)
#define NSP_SYM_LEAN_GO(BS, TMAX, FORMX)                                                        \
    hipLaunchKernelGGL((k_sym_lean<BS, TMAX, (BS >= 512 ? 4 : 2), FORMX>), dim3(8 * ceil_div(hist[bin_], 8)), dim3(BS), 0, st, \
                       arpt, acol, brpt, bcol, row_perm, row_prod, row_maxb, row_nz, off[bin_], hist[bin_], b->nnz, d_bs, \
                       TMAX >= 8192 ? tcol : (int *)nullptr, list_off, row_span, 12, 12288)
#define NSP_SYM_TB(BIN, BS, TMAX)                                                              \
    if (hist[BIN] > 0 && now(BIN)) {                                                           \
        constexpr int bin_ = BIN;                                                              \
        hipStream_t st = L.begin(BIN);                                                         \
        bool lean_done = false;                                                                \
        if constexpr (kExperiments) { /* the lean family: experiments build only, until it has been timed on the device */ \
            if ((g_tb_lean & 1) && b->N <= (1 << 24)) { /* lean_slot hashes 24 bits of the column */ \
                switch ((g_tb_lean >> 2) & 3) { /* bits 2, 3: branch-free retries, pipelined walk */ \
                    case 1: NSP_SYM_LEAN_GO(BS, TMAX, 1); break;                                 \
                    case 2: NSP_SYM_LEAN_GO(BS, TMAX, 2); break;                                 \
                    case 3: NSP_SYM_LEAN_GO(BS, TMAX, 3); break;                                 \
                    default: NSP_SYM_LEAN_GO(BS, TMAX, 0);                                       \
                }                                                                              \
                lean_done = true;                                                              \
            }                                                                                  \
        }                                                                                      \
        if (!lean_done) NSP_SYM_TB_GO(BS, TMAX);                                               \
        NSP_LAUNCH_CHECK();                                                                    \
        L.end(BIN);                                                                            \
    }
#define NSP_SYM_DENSE(BIN, BS, SPAN)                                                            \
    if (hist[BIN] > 0 && now(BIN)) {                                                                       \
        hipStream_t st = L.begin(BIN);                                                         \
        static DevOnce big_ok;                                                                   \
        allow_big_lds(k_sym_dense<BS, SPAN>, big_ok, SPAN + 64, cx.device);                               \
        const int span_b = max_span[BIN] < SPAN ? max_span[BIN] : SPAN;                        \
        const size_t lds = sizeof(int) * (size_t)((span_b + 63) / 64 * 16 + 16);               \
        hipLaunchKernelGGL((k_sym_dense<BS, SPAN>), dim3(8 * ceil_div(hist[BIN], 8)), dim3(BS), lds, st, \
                           arpt, acol, brpt, bcol, row_perm, row_prod, row_maxb, row_lo, row_span, row_nz, \
                           off[BIN], hist[BIN], b->nnz, bm_off, bm, row_span_num, btwin, sdesc); \
        NSP_LAUNCH_CHECK();                                                                    \
        L.end(BIN);                                                                            \
    }
#define NSP_SYM_BITS(BIN, BS, WORDS)                                                           \
    if (hist[BIN] > 0 && now(BIN)) {                                                                       \
        hipStream_t st = L.begin(BIN);                                                         \
        hipLaunchKernelGGL((k_sym_bits<BS, WORDS>), dim3(8 * ceil_div(hist[BIN], 8)), dim3(BS), 0, st, \
                           arpt, acol, brpt, bcol, row_perm, row_prod, row_maxb, row_lo, row_span, row_nz, \
                           off[BIN], hist[BIN], b->nnz, d_bs, (int *)nullptr, list_off, -1LL, 12, 12288); \
        NSP_LAUNCH_CHECK();                                                                    \
        L.end(BIN);                                                                            \
    }
    // Host launch cost is on the critical path (~14 us per bin: stream wait, two event records,
    // the launch): the bin with the most rows -- the one on the main stream -- goes out early, the
    // others follow biggest rows first.
    // Workgroups that need (nearly) the whole LDS of a CU -- the 2^20-bit window, the 32768-key
    // table -- are issued before the flood of small workgroups: behind it they would wait for a CU
    // to drain completely, i.e. for the end of the biggest kernel (webbase class: one hub row then
    // ran alone for 128 us at the end of the phase).
    for (int pass = 0; pass < 3; pass++) {
    auto now = [&](int bin) {
        const bool big = bin == 10 || bin == 4;
        return big ? pass == 0 : (bin == L.most_bin ? pass == 1 : pass == 2);
    };
    // bin 10 with windows wider than the 2^20-bit window and sorted rows of B: cursor kernel, every
    // product seen once (k_sym_bits would walk all products once per 2^20-column piece)
    static const int sym_cursor_on = exp_env("NSPARSE_SYM_CURSOR", 1) != 0;
    static const int sym_long_len = exp_env("NSPARSE_SYM_LONG", 32);
    // NSPARSE_HEAVY_FLAT bit 2 (experiments build): the same rows through the stateless symbolic kernel (heavy_flat.h)
    static const int sym_flat = exp_env("NSPARSE_HEAVY_FLAT", 0) & 4;
    bool sym_flat_done = false;
    if constexpr (kExperiments) {
        if (sym_flat && panel_table_fits(b) && hist[10] > 0 && now(10) && sym_cursor_on && b_sorted && max_span[10] > 32768 * 32 &&
            max_alen > 0) {
            hipStream_t st = L.begin(10);
            const int rows = hist[10];
            const int groups = rows < 1024 ? rows : 1024;
            build_panel_table(b, L, st);
            hipLaunchKernelGGL((k_sym_flat<1024, 1048576, kPanelW>), dim3(groups), dim3(1024), 0, st, arpt, acol, brpt, bcol, b->nnz,
                               (const int *)g_panel.slot_of, (const int *)g_panel.tab, g_panel.np + 1, row_perm, off[10], rows, d_bs,
                               row_lo, row_span, row_nz, tcol, list_off, exp_env("NSPARSE_LIST_DRY", 0) ? -2LL : -1LL,
                               (const int *)row_prod);
            NSP_LAUNCH_CHECK();
            L.end(10);
            sym_flat_done = true;
        }
    }
    if (sym_flat_done) {
    } else if (hist[10] > 0 && now(10) && sym_cursor_on && b_sorted && max_span[10] > 32768 * 32 && max_alen > 0) {
        hipStream_t st = L.begin(10);
        const int rows = hist[10];
        const int amax = (max_alen + 1) & ~1;
        const long long stride_ints = 3LL * amax + (long long)amax * (sizeof(real) / sizeof(int));
        const int groups = rows < 1024 ? rows : 1024;
        int *slab = (int *)dev_alloc(sizeof(int) * (size_t)stride_ints * groups);
        hipLaunchKernelGGL((k_num_ranked<1024, 1048576, 8, 1024, true>), dim3(groups), dim3(1024), 0, st, arpt, acol,
                           (const real *)nullptr, brpt, bcol, (const real *)nullptr, (const int *)nullptr,
                           (int *)nullptr, (real *)nullptr, row_perm, off[10], rows, d_bs, row_lo, row_span, slab,
                           stride_ints, amax, 0, sym_long_len, -1, 0, (unsigned long long *)nullptr, row_nz, tcol, list_off,
                           exp_env("NSPARSE_LIST_DRY", 0) ? -2LL : -1LL, (const int *)row_prod);
        NSP_LAUNCH_CHECK();
        L.end(10);
        L.free_later(slab);
    } else {
        NSP_SYM_BITS(10, 1024, 32768)
    }
    NSP_SYM_BITS(9, 512, 8192)
    // workgroup sizes: the winners of the round-1 / round-2 sweeps; a -DNSPARSE_EXPERIMENTS build keeps
    // the alternatives behind NSPARSE_SYMD{6,7,8}_BS / NSPARSE_SYM{2,3}_BS
#ifdef NSPARSE_EXPERIMENTS
    static const int tune_d6 = exp_env("NSPARSE_SYMD6_BS", 128), tune_d8 = exp_env("NSPARSE_SYMD8_BS", 512);
    static const int tune_d7 = exp_env("NSPARSE_SYMD7_BS", 256);
    if (tune_d8 == 256) { NSP_SYM_DENSE(8, 256, 65536) } else if (tune_d8 == 512) { NSP_SYM_DENSE(8, 512, 65536) } else { NSP_SYM_DENSE(8, 1024, 65536) }
    if (tune_d7 == 128) { NSP_SYM_DENSE(7, 128, 16384) } else if (tune_d7 == 256) { NSP_SYM_DENSE(7, 256, 16384) } else { NSP_SYM_DENSE(7, 512, 16384) }
    if (tune_d6 == 128) { NSP_SYM_DENSE(6, 128, 4096) } else if (tune_d6 == 64) { NSP_SYM_DENSE(6, 64, 4096) } else if (tune_d6 == 512) { NSP_SYM_DENSE(6, 512, 4096) } else { NSP_SYM_DENSE(6, 256, 4096) }
    static const int tune_s3 = exp_env("NSPARSE_SYM3_BS", 512), tune_s2 = exp_env("NSPARSE_SYM2_BS", 128);
    NSP_SYM_TB(4, 1024, 32768)
    if (tune_s3 == 512) { NSP_SYM_TB(3, 512, 8192) } else if (tune_s3 == 1024) { NSP_SYM_TB(3, 1024, 8192) } else { NSP_SYM_TB(3, 256, 8192) }
    if (tune_s2 == 256) { NSP_SYM_TB(2, 256, 2048) } else if (tune_s2 == 64) { NSP_SYM_TB(2, 64, 2048) } else { NSP_SYM_TB(2, 128, 2048) }
#else
    NSP_SYM_DENSE(8, 512, 65536)
    NSP_SYM_DENSE(7, 256, 16384)
    NSP_SYM_DENSE(6, 128, 4096)
    NSP_SYM_TB(4, 1024, 32768)
    NSP_SYM_TB(3, 512, 8192)
    NSP_SYM_TB(2, 128, 2048)
#endif
    // (bin 1 with persistent wavefronts and the next rows' loads in flight -- k_sym_wave, round 3 -- was measured
    //  slower than one workgroup per row: stencil 3.07 -> 3.35 ms, webbase-1M class 2.36 -> 2.43; removed in round 4)
#ifdef NSPARSE_EXPERIMENTS
    if (sym_ladder().hash_t[0] > 435) { NSP_SYM_TB(1, 64, 1024) } else { NSP_SYM_TB(1, 64, 512) }
#else
    NSP_SYM_TB(1, 64, 1024)
#endif
    if (hist[0] > 0 && now(0)) {
        hipStream_t st = L.begin(0);
        constexpr int BS = 256, LPR = 4;
        hipLaunchKernelGGL((k_sym_small<BS, LPR, 64>), dim3(ceil_div(hist[0], BS / LPR)), dim3(BS), 0,
                           st, arpt, acol, brpt, bcol, row_perm, row_nz, off[0], hist[0]);
        NSP_LAUNCH_CHECK();
        L.end(0);
    }
    }  // pass
#undef NSP_SYM_BITS
#undef NSP_SYM_DENSE
#undef NSP_SYM_TB
#undef NSP_SYM_TB_GO
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
                                 const int *hist_in, int max_nz, BinState *d_bs, Context &cx,
                                 float *ms_bin, int write_col, const int *bm_off,
                                 const unsigned int *bm, int max_alen, bool b_sorted,
                                 const int *max_span, const unsigned char *grp, const unsigned char *btwin,
                                 const int *listed, const int *members, const int4 *desc, const int *bkey,
                                 const int *tcol, const long long *list_off, long long list_w)
{
    int hist[NB], off[NB + 1];
    fold_small_hash_bins(hist_in, hist, off);
    BinLauncher L(cx, 1, hist, kNumGlobalBin);
    const int *arpt = a->d_rpt, *acol = a->d_col, *brpt = b->d_rpt, *bcol = b->d_col;
    const real *aval = a->d_val, *bval = b->d_val;
    L.fork();
    // The heavy bin goes first: its few persistent workgroups want a whole CU each (LDS) and run
    // longest, so they should not queue behind a million small rows.  Nothing here waits on the
    // host; the cursor slab returns to the cache when the call has drained (collect()).
    constexpr int kTileW = 12288;  // LDS accumulators are double in both builds
    static const int tiled_on = exp_env("NSPARSE_TILED", 1) != 0;
    // B rows longer than this are swept by whole wavefronts (R-MAT-16 / 18 / 22: 128 -> 32 saves 9 / 8 / 4 %)
    static const int long_len = exp_env("NSPARSE_TILED_LONG", 32);
    static const int tile_sel = exp_env("NSPARSE_TILED_W", 0);
    // rows with fewer than one non-zero per ranked_dens columns of their window take the ranked
    // kernel (0: none, < 0: all)
    static const int ranked_dens = exp_env("NSPARSE_RANKED_DENS", 12);
    // dense tiles alone: at most 1024 per row, wider matrices hash globally; with the ranked kernel
    // taking the wide rows there is no limit
    const bool use_tiled = tiled_on && b_sorted && hist[kNumGlobalBin] > 0 && max_alen > 0 &&
                           (ranked_dens != 0 || (long long)b->N <= (long long)kTileW * 1024);
    // -DNSPARSE_EXPERIMENTS: NSPARSE_TB_PAD / NSPARSE_BLK_PAD = extra dynamic LDS per workgroup (fewer groups in flight per
    // CU: what bounds the kernel?), NSPARSE_TB_PROF / NSPARSE_BLK_PROF / NSPARSE_TILED_PROF = in-kernel phase timers
    static const int tb_pad = exp_env("NSPARSE_TB_PAD", 0), blk_pad = exp_env("NSPARSE_BLK_PAD", 0);
    unsigned long long *tb_prof = nullptr, *blk_prof = nullptr;
#ifdef NSPARSE_EXPERIMENTS
    if (exp_env("NSPARSE_TB_PROF", 0)) {
        tb_prof = (unsigned long long *)dev_alloc(8 * NB * sizeof(unsigned long long));
        NSP_CHECK(hipMemsetAsync(tb_prof, 0, 8 * NB * sizeof(unsigned long long), cx.stream[0]));
        NSP_CHECK(hipStreamSynchronize(cx.stream[0]));
    }
    if (exp_env("NSPARSE_BLK_PROF", 0)) {
        blk_prof = (unsigned long long *)dev_alloc(8 * sizeof(unsigned long long) * (size_t)(a->M + 8));
        NSP_CHECK(hipMemsetAsync(blk_prof, 0, 8 * sizeof(unsigned long long) * (size_t)(a->M + 8), cx.stream[0]));
        NSP_CHECK(hipStreamSynchronize(cx.stream[0]));
    }
#endif
    constexpr int kBlkU = 2;  // tasks in flight per lane in the node-block kernel (of kBlkCols columns each)
    // launch order: the heavy bin, then the bin with the most rows (main stream), then the rest
    // biggest rows first (see symbolic_phase)
    for (int pass = 0; pass < 3; pass++) {
    auto now = [&](int bin) {
        const bool big = bin == kNumGlobalBin;  // persistent workgroups that own a CU's LDS
        return big ? pass == 0 : (bin == L.most_bin ? pass == 1 : pass == 2);
    };
    if (use_tiled && now(kNumGlobalBin)) {
        hipStream_t st = L.begin(kNumGlobalBin);
        const int rows = hist[kNumGlobalBin];
        const int amax = (max_alen + 1) & ~1;  // even: the value slice stays 8-byte aligned
        const long long stride_ints = 3LL * amax + (long long)amax * (sizeof(real) / sizeof(int));
        const int groups = rows < 1024 ? rows : 1024;
        int *slab = (int *)dev_alloc(sizeof(int) * (size_t)stride_ints * groups);
        static const int tiled_prof = exp_env("NSPARSE_TILED_PROF", 0);
        unsigned long long *d_prof = nullptr;
        if (tiled_prof) {
            d_prof = (unsigned long long *)dev_alloc(32 * sizeof(unsigned long long));
            NSP_CHECK(hipMemsetAsync(d_prof, 0, 32 * sizeof(unsigned long long), st));
        }
#define NSP_TILED(BSX, WX)                                                                     \
    hipLaunchKernelGGL((k_num_tiled<BSX, WX>), dim3(groups), dim3(BSX), 0, st, arpt, acol, aval, brpt, \
                       bcol, bval, c->d_rpt, c->d_col, c->d_val, row_perm, off[kNumGlobalBin], rows,   \
                       d_bs, row_lo, row_span, slab, stride_ints, amax, write_col, long_len, d_prof, ranked_dens, \
                       list_off, list_w, row_prod)
        // NSPARSE_HEAVY_FLAT=1 (experiments build, until it has been timed on the device): the dense tiles as
        // stateless flat walks over a panel table of B (heavy_flat.h) instead of the cursor kernel
        static const int heavy_flat = exp_env("NSPARSE_HEAVY_FLAT", 0);  // bit 0: dense tiles, bit 1: list-driven ranked tiles, bit 2: symbolic
        bool flat_done = false;
        int *pt_slot_of = nullptr, *pt_tab = nullptr;
        int pt_np = 0;
        if constexpr (kExperiments) {
            if (heavy_flat & 3) {
                static_assert(kPanelW == kTileW, "a dense tile is one panel of the table");
                build_panel_table(b, L, st);  // (the symbolic phase may have built it already)
                pt_slot_of = g_panel.slot_of, pt_tab = g_panel.tab, pt_np = g_panel.np;
                if ((heavy_flat & 1) && pt_tab != nullptr && ranked_dens >= 0 && tile_sel == 0) {
                    hipLaunchKernelGGL((k_num_flat<1024, kTileW>), dim3(groups), dim3(1024), 0, st, arpt, acol, aval, brpt, bcol,
                                       bval, b->nnz, (const int *)pt_slot_of, (const int *)pt_tab, pt_np + 1, c->d_rpt, c->d_col, c->d_val,
                                       row_perm, off[kNumGlobalBin], rows, d_bs, row_lo, row_span, write_col, ranked_dens,
                                       list_off, list_w, row_prod, slab, stride_ints, amax);
                    flat_done = true;
                }
            }
        }
        if (ranked_dens >= 0 && !flat_done) {
#ifdef NSPARSE_EXPERIMENTS
            if (tile_sel == 1) { NSP_TILED(1024, kTileW / 2); }
            else if (tile_sel == 2) { NSP_TILED(512, kTileW / 2); }
            else if (tile_sel == 3) { NSP_TILED(512, kTileW / 4); }
            else
#endif
            { NSP_TILED(1024, kTileW); }
        }
#undef NSP_TILED
        NSP_LAUNCH_CHECK();
        // thin rows: bitmap-ranked accumulator (same stream: both kernels want the whole LDS of a CU)
        if (ranked_dens != 0) {
            constexpr int kRankCap = 10240;
            // Matrices wider than 2^20 columns take tiles of 2^19 columns with a smaller value array:
            // their thin rows are bound by the number of tiles (R-MAT-22: -5 %), while on
            // narrower matrices the smaller array costs cuts (R-MAT-18: +4 %).  NSPARSE_RANKED_SEL=0/1 forces.
            static const int ranked_env = exp_env("NSPARSE_RANKED_SEL", -1);
            const int ranked_sel = ranked_env >= 0 ? ranked_env : (b->N > (1 << 20) ? 1 : 0);
#define NSP_RANKED(WX, CAPX, LCAPX)                                                             \
    hipLaunchKernelGGL((k_num_ranked<1024, WX, CAPX, LCAPX>), dim3(groups), dim3(1024), 0, st, arpt, acol, aval,  \
                       brpt, bcol, bval, c->d_rpt, c->d_col, c->d_val, row_perm, off[kNumGlobalBin], rows, d_bs,   \
                       row_lo, row_span, slab, stride_ints, amax, write_col, long_len, ranked_dens,                 \
                       tile_sel == 0 ? kTileW : (tile_sel == 3 ? kTileW / 4 : kTileW / 2), d_prof, (int *)nullptr,  \
                       const_cast<int *>(tcol), const_cast<long long *>(list_off), list_w, row_prod, ranked_flat ? 1 : 0)
            // NSPARSE_HEAVY_FLAT bit 1: the rows that have a column list through stateless tiles (heavy_flat.h); the
            // cursor kernel then takes only the rows without one
            const bool ranked_flat = kExperiments && (heavy_flat & 2) && pt_tab != nullptr && tcol != nullptr;
            if constexpr (kExperiments) {
                if (ranked_flat) {
#define NSP_RANKED_FLAT(WX, CAPX)                                                               \
    hipLaunchKernelGGL((k_num_ranked_flat<1024, WX, CAPX, kTileW>), dim3(groups), dim3(1024), 0, st, arpt, acol, aval, brpt, \
                       bcol, bval, b->nnz, (const int *)pt_slot_of, (const int *)pt_tab, pt_np + 1, c->d_rpt, c->d_col,    \
                       c->d_val, row_perm, off[kNumGlobalBin], rows, d_bs, row_lo, row_span, write_col, ranked_dens,       \
                       tile_sel == 0 ? kTileW : (tile_sel == 3 ? kTileW / 4 : kTileW / 2), tcol, list_off, list_w, row_prod, \
                       slab, stride_ints, amax)
                    if (ranked_sel == 1) { NSP_RANKED_FLAT(524288, 5120); }
                    else { NSP_RANKED_FLAT(262144, kRankCap); }
#undef NSP_RANKED_FLAT
                    NSP_LAUNCH_CHECK();
                }
            }
            if (ranked_sel == 1) { NSP_RANKED(524288, 6144, 512); }
            else { NSP_RANKED(262144, kRankCap, 1024); }
#undef NSP_RANKED
            NSP_LAUNCH_CHECK();
        }
        L.end(kNumGlobalBin);
        if (d_prof) {
            NSP_CHECK(hipStreamSynchronize(st));
            unsigned long long h[32];
            NSP_CHECK(hipMemcpy(h, d_prof, sizeof(h), hipMemcpyDeviceToHost));
            const double us = 0.01 / groups;  // 100 MHz ticks summed over the workgroups
            fprintf(stderr, "[tiled] groups %d rows %llu tiles %llu | per-group us: setup %.0f queue %.0f regs %.0f overflow %.0f emit %.0f+%.0f+%.0f+%.0f | trips %llu (long %llu)\n",
                    groups, h[6], h[5], h[0] * us, h[1] * us, h[2] * us, h[3] * us, h[7] * us, h[8] * us, h[9] * us, h[4] * us, h[10], h[11]);
            fprintf(stderr, "[ranked] rows %llu tiles %llu | per-group us: setup %.0f pass1 %.0f (first tile %.0f) scan %.0f pass2 %.0f (first %.0f) emit %.0f | cuts %llu | sum nlong %llu alen %llu list-overflow rows %llu\n",
                    h[22], h[21], h[16] * us, h[17] * us, h[23] * us, h[18] * us, h[19] * us, h[24] * us, h[20] * us, h[25], h[26], h[27], h[28]);
            dev_free(d_prof);
        }
        L.free_later(slab);
    }
#define NSP_NUM_TB_GO(BS, TMAX, PMAX)                                                           \
    hipLaunchKernelGGL((k_num_tb<BS, TMAX, PMAX>), dim3(8 * ceil_div(hist[bin_], 8)), dim3(BS), tb_pad, st, arpt, \
                       acol, aval, brpt, bcol, bval, c->d_rpt, c->d_col, c->d_val, row_perm,    \
                       row_prod, row_maxb, off[bin_], hist[bin_], b->nnz,                       \
                       write_col | (g_flat ? 0 : 4) | (g_flat == 2 ? 8 : 0),                    \
                       tb_prof ? tb_prof + 8 * bin_ : nullptr)
#define NSP_NUM_LEAN_GO(BS, TMAX, FORMX)                                                        \
    hipLaunchKernelGGL((k_num_lean<BS, TMAX, (BS >= 512 ? 4 : 2), FORMX>), dim3(8 * ceil_div(hist[bin_], 8)), dim3(BS), 0, st, \
                       arpt, acol, aval, brpt, bcol, bval, c->d_rpt, c->d_col, c->d_val, row_perm, row_prod, row_maxb, \
                       off[bin_], hist[bin_], b->nnz, write_col)
#define NSP_NUM_TB(BIN, BS, TMAX, PMAX)                                                        \
    if (hist[BIN] > 0 && now(BIN)) {                                                           \
        constexpr int bin_ = BIN;                                                              \
        hipStream_t st = L.begin(BIN);                                                         \
        bool lean_done = false;                                                                \
        if constexpr (kExperiments) {                                                          \
            if ((g_tb_lean & 2) && !tb_prof && b->N <= (1 << 24)) {                              \
                switch ((g_tb_lean >> 2) & 3) {                                                \
                    case 1: NSP_NUM_LEAN_GO(BS, TMAX, 1); break;                                 \
                    case 2: NSP_NUM_LEAN_GO(BS, TMAX, 2); break;                                 \
                    case 3: NSP_NUM_LEAN_GO(BS, TMAX, 3); break;                                 \
                    default: NSP_NUM_LEAN_GO(BS, TMAX, 0);                                       \
                }                                                                              \
                lean_done = true;                                                              \
            }                                                                                  \
        }                                                                                      \
        if (!lean_done) NSP_NUM_TB_GO(BS, TMAX, PMAX);                                         \
        NSP_LAUNCH_CHECK();                                                                    \
        L.end(BIN);                                                                            \
    }
#define NSP_NUM_DENSE_GO(BS, SPAN, MODEX)                                                       \
    {                                                                                          \
        /* the first kernel keeps at most two bitmap words per lane: widest windows need 512 threads */ \
        constexpr int BSO = (SPAN / (BS / 64) + 2047) / 2048 <= 2 ? BS : 512;                  \
        static DevOnce big_ok;                                                                   \
        allow_big_lds(k_num_dense<BSO, SPAN, MODEX>, big_ok, (int)sizeof(acc_t) * (SPAN + 64), cx.device); \
        hipLaunchKernelGGL((k_num_dense<BSO, SPAN, MODEX>), dim3(8 * ceil_div(hist[bin_], 8)), dim3(BSO), \
                           lds, st, arpt, acol, aval, brpt, bcol, bval, c->d_rpt, c->d_col,      \
                           c->d_val, row_perm, row_prod, row_maxb, row_lo, row_span, off[bin_],  \
                           hist[bin_], b->nnz, bm_off, bm);                                    \
    }
#define NSP_NUM_BLOCK_GO2(BS, SPAN, MODEX, KEYEDX)                                               \
    {                                                                                          \
        static DevOnce big_ok;                                                                   \
        allow_big_lds(k_num_block<BS, SPAN, MODEX, kBlkU, KEYEDX>, big_ok, (int)sizeof(acc_t) * (SPAN + 64), cx.device); \
        /* followers of a group head are not listed (k_bin_scatter): listed[bin] heads */        \
        const int heads = grp ? listed[bin_] : hist[bin_];                                     \
        /* fold6_: the rows of bin 6 ride along (their stretch of the list first) */             \
        const int heads6 = fold6_ ? (grp ? listed[6] : hist[6]) : 0;                            \
        hipLaunchKernelGGL((k_num_block<BS, SPAN, MODEX, kBlkU, KEYEDX>), dim3(8 * ceil_div(heads + heads6, 8)), dim3(BS), \
                           lds_blk, st, arpt, acol, aval, brpt, bcol, bval, c->d_rpt, c->d_col,  \
                           c->d_val, row_perm, row_maxb, row_lo, row_span, (fold6_ && !fold_rev) ? off[6] : off[bin_], \
                           heads + heads6, b->nnz, bm_off, bm, grp, btwin, blk_prof ? blk_prof + 8ull * off[bin_] : nullptr, members, desc, bkey, \
                           fold6_ ? (fold_rev ? heads : heads6) : 0x7fffffff, fold_rev ? off[6] : off[bin_]); \
    }
// keyed runs (twin rows of B that are not neighbours): the default 128-thread, full-call form only
#define NSP_NUM_BLOCK_GO(BS, SPAN, MODEX)                                                       \
    {                                                                                          \
        if (bkey != nullptr && BS == 128 && MODEX == 1) NSP_NUM_BLOCK_GO2(128, SPAN, 1, true)  \
        else NSP_NUM_BLOCK_GO2(BS, SPAN, MODEX, false)                                         \
    }
#define NSP_NUM_DENSE(BIN, BS, SPAN)                                                            \
    if (hist[BIN] > 0 && now(BIN) && !(BIN == 6 && fold6)) {                                   \
        constexpr int bin_ = BIN;                                                              \
        const bool fold6_ = BIN == 7 && fold6;                                                 \
        hipStream_t st = L.begin(BIN);                                                         \
        const int span_b = max_span[BIN] < SPAN ? max_span[BIN] : SPAN;                        \
        const int stride_b = (span_b + 63) / 64 * 64 + 8;                                      \
        const size_t lds = sizeof(acc_t) * (size_t)stride_b;                                   \
        /* node-block kernel (block.h): compact accumulators, nnz of the row each, up to kBlkRows rows */ \
        const int nz1 = ((max_nz < span_b ? max_nz : span_b) + 7) / 8 * 8 + 8;                 \
        int blk_elems = nz1;  /* any group satisfies rows * own nnz <= min(budget, 3 * longest row) */ \
        if (grp) {                                                                             \
            const int want = kBlkRows * nz1 < kBlkAccElems ? kBlkRows * nz1 : kBlkAccElems;    \
            blk_elems = want > nz1 ? want : nz1;                                               \
        }                                                                                      \
        const size_t lds_blk = sizeof(acc_t) * (size_t)blk_elems + (size_t)blk_pad;            \
        /* product builds: the node-block kernel runs with 128 threads and the first kernel never does (tune_nd*   \
           below), so each (bin, workgroup size) pair instantiates ONE of the two kernels -- the discarded branch of \
           an `if constexpr` is not odr-used (round 5: twelve kernel instantiations nobody could launch) */           \
        if (lean_on && (grp || blk_all)) {                                                     \
            if constexpr ((BS) == 128 || kExperiments) {                                       \
                if (write_col & 1) NSP_NUM_BLOCK_GO(BS, SPAN, 1) else NSP_NUM_BLOCK_GO(BS, SPAN, 2) \
            } else {  /* the invariant blk <=> 128 threads broke: report, never leave the bin's rows unwritten */ \
                set_error(-21, "window kernel not instantiated for this workgroup size", __FILE__, __LINE__); \
            }                                                                                  \
        } else {                                                                               \
            if constexpr ((BS) != 128 || kExperiments) {                                       \
                if (write_col & 1) NSP_NUM_DENSE_GO(BS, SPAN, 1) else NSP_NUM_DENSE_GO(BS, SPAN, 2) \
            } else {                                                                           \
                set_error(-21, "window kernel not instantiated for this workgroup size", __FILE__, __LINE__); \
            }                                                                                  \
        }                                                                                      \
        NSP_LAUNCH_CHECK();                                                                    \
        L.end(BIN);                                                                            \
    }
    // node-block numeric window kernel (block.h) for matrices with twin rows (grp != nullptr); rows of
    // matrices without that structure are one-row groups with one-entry runs, which the first kernel
    // (window.h: four entries per lane and step) walks in fewer instructions: cant-class irregular
    // stand-in 0.42 ms against 0.63.  NSPARSE_WINDOW_KERNEL=0 (experiments build): always the first kernel; =2: always the block one.
    static const bool lean_on = exp_env("NSPARSE_WINDOW_KERNEL", 1) != 0;
    static const bool blk_all = exp_env("NSPARSE_WINDOW_KERNEL", 1) == 2;
    // workgroup sizes of the window bins: the node-block kernel is bound by the latency of its dependent
    // loads, i.e. by the groups in flight per CU, and does best with 128 threads per group (cant class:
    // 64 / 128 / 256 / 512 threads -> 0.240 / 0.197 / 0.216 / 0.41 ms); the first kernel keeps 256 / 256 / 512
    const bool blk = lean_on && (grp || blk_all);
    static const int env_nd6 = exp_env("NSPARSE_NUMD6_BS", 0), env_nd7 = exp_env("NSPARSE_NUMD7_BS", 0),
                     env_nd8 = exp_env("NSPARSE_NUMD8_BS", 0);
    const int tune_nd6 = env_nd6 ? env_nd6 : (blk ? 128 : 256);
    const int tune_nd7 = env_nd7 ? env_nd7 : (blk ? 128 : 256);
    const int tune_nd8 = env_nd8 ? env_nd8 : (blk ? 128 : 512);
    // Two window bins of a finite-element matrix in ONE launch (NSPARSE_FOLD_WIN=1; off by default): when the
    // node-block kernel serves both bin 6 (windows up to 1536 columns) and bin 7 (up to 4096), the 4096 instance can
    // take the rows of bin 6 as well (the kernel sizes everything by the row's own window): one fork / join and one
    // ramp-up less.  Measured on the irregular cant-class stand-in (42 K + 20 K rows), same box, three runs each:
    // 0.377-0.384 ms with two launches side by side, 0.393-0.402 folded.  Two kernels that overlap hide each
    // other's tails better than one.
    static const bool fold_win = exp_env("NSPARSE_FOLD_WIN", 0) >= 1;
    static const bool fold_rev = exp_env("NSPARSE_FOLD_WIN", 0) == 2;  // bin 7's rows first
    const bool fold6 = fold_win && blk && tune_nd7 == 128 && tune_nd6 == 128 && hist[6] > 0 && hist[7] > 0;
    // ranked-window rows (numeric bin 9): always the node-block kernel, windows up to 65536 columns
    if (hist[kRankBin] > 0 && now(kRankBin)) {
        constexpr int bin_ = kRankBin;
        hipStream_t st = L.begin(kRankBin);
        const int span_b = max_span[kRankBin] < 65536 ? max_span[kRankBin] : 65536;
        const int nzcap = num_ladder().rank_max_nz;
        const int nz1 = ((max_nz < nzcap ? max_nz : nzcap) + 7) / 8 * 8 + 8;
        int blk_elems = nz1;
        if (grp) {
            const int want = kBlkRows * nz1 < kBlkAccElems ? kBlkRows * nz1 : kBlkAccElems;
            blk_elems = want > nz1 ? want : nz1;
        }
        (void)span_b;
        const size_t lds_blk = sizeof(acc_t) * (size_t)blk_elems;
        const int heads = grp ? listed[bin_] : hist[bin_];
#define NSP_RANKWIN(MODEX)                                                                      \
    hipLaunchKernelGGL((k_num_block<128, 65536, MODEX, kBlkU>), dim3(8 * ceil_div(heads, 8)), dim3(128), lds_blk, st, \
                       arpt, acol, aval, brpt, bcol, bval, c->d_rpt, c->d_col, c->d_val, row_perm, row_maxb, \
                       row_lo, row_span, off[bin_], heads, b->nnz, bm_off, bm, grp, btwin, blk_prof ? blk_prof + 8ull * off[bin_] : nullptr, members, desc, (const int *)nullptr)
        if (write_col & 1) NSP_RANKWIN(1); else NSP_RANKWIN(2);
#undef NSP_RANKWIN
        NSP_LAUNCH_CHECK();
        L.end(kRankBin);
    }
    // (the node-block kernel runs 128 threads per group, the first window kernel 256 / 256 / 512)
    if (tune_nd8 == 128) { NSP_NUM_DENSE(8, 128, 12288) }
#ifdef NSPARSE_EXPERIMENTS
    else if (tune_nd8 == 256) { NSP_NUM_DENSE(8, 256, 12288) }
#endif
    else { NSP_NUM_DENSE(8, 512, 12288) }
    if (tune_nd7 == 128) { NSP_NUM_DENSE(7, 128, 4096) } else { NSP_NUM_DENSE(7, 256, 4096) }
    if (tune_nd6 == 128) { NSP_NUM_DENSE(6, 128, 1536) }
#ifdef NSPARSE_EXPERIMENTS
    else if (tune_nd6 == 64) { NSP_NUM_DENSE(6, 64, 1536) } else if (tune_nd6 == 512) { NSP_NUM_DENSE(6, 512, 1536) }
#endif
    else { NSP_NUM_DENSE(6, 256, 1536) }
    NSP_NUM_TB(4, 1024, 8192, 8192)
    NSP_NUM_TB(3, 512, 4096, 4096)
#ifdef NSPARSE_EXPERIMENTS
    static const int tune_n2 = exp_env("NSPARSE_NUM2_BS", 256), tune_n1 = exp_env("NSPARSE_NUM1_BS", 64);
    if (tune_n2 == 128) { NSP_NUM_TB(2, 128, 1024, 1024) } else if (tune_n2 == 512) { NSP_NUM_TB(2, 512, 1024, 1024) } else { NSP_NUM_TB(2, 256, 1024, 1024) }
    if (tune_n1 == 128) { NSP_NUM_TB(1, 128, 256, 256) } else { NSP_NUM_TB(1, 64, 256, 256) }
#else
    NSP_NUM_TB(2, 256, 1024, 1024)
    // (bin 1 with persistent wavefronts and the next rows' loads in flight -- k_num_wave, round 3 -- was measured
    //  slower: 27-point stencil numeric 1.87 -> 2.58 ms; removed in round 4)
    NSP_NUM_TB(1, 64, 256, 256)
#endif
    if (hist[0] > 0 && now(0)) {
        hipStream_t st = L.begin(0);
        constexpr int BS = 256, LPR = 4;
        hipLaunchKernelGGL((k_num_small<BS, LPR, 32>), dim3(ceil_div(hist[0], BS / LPR)), dim3(BS), 0,
                           st, arpt, acol, aval, brpt, bcol, bval, c->d_rpt, c->d_col, c->d_val,
                           row_perm, off[0], hist[0], write_col);
        NSP_LAUNCH_CHECK();
        L.end(0);
    }
    }  // pass
#undef NSP_NUM_DENSE
#undef NSP_NUM_DENSE_GO
#undef NSP_NUM_BLOCK_GO
#undef NSP_NUM_TB
#undef NSP_NUM_TB_GO
    // rows beyond the LDS tables without the tile kernels (unsorted B, or switched off): global
    // table + segmented sort; synchronises on the host (scratch freed here), hence last
    if (!use_tiled && hist[kNumGlobalBin] > 0) {
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
#ifdef NSPARSE_EXPERIMENTS
    if (tb_prof) {
        NSP_CHECK(hipDeviceSynchronize());
        unsigned long long h[8 * NB];
        NSP_CHECK(hipMemcpy(h, tb_prof, sizeof(h), hipMemcpyDeviceToHost));
        for (int q = 1; q <= 4; q++) {
            const unsigned long long *r = h + 8 * q;
            if (!r[5]) continue;
            const double us = 0.01 / (double)r[5];
            fprintf(stderr, "[tb] numeric bin %d rows %llu | us per row: record+clear %.2f walk %.2f compact %.2f sort %.2f read-out %.2f\n",
                    q, r[5], r[0] * us, r[1] * us, r[2] * us, r[3] * us, r[4] * us);
        }
        dev_free(tb_prof);
    }
    if (blk_prof) {
        NSP_CHECK(hipDeviceSynchronize());
        std::vector<unsigned long long> hb(8 * (size_t)(a->M + 8));
        NSP_CHECK(hipMemcpy(hb.data(), blk_prof, hb.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long h[8] = {};
        // [5], [6]: start / end of every group on the 100 MHz clock -> groups resident over the kernel's life
        unsigned long long t_lo = ~0ull, t_hi = 0, life = 0;
        for (size_t i = 0; i < hb.size(); i += 8) {
            if (hb[i + 6] == 0) continue;
            for (int q = 0; q < 5; q++) h[q] += hb[i + q];
            h[6]++;
            h[7] += hb[i + 7];
            t_lo = hb[i + 5] < t_lo ? hb[i + 5] : t_lo;
            t_hi = hb[i + 6] > t_hi ? hb[i + 6] : t_hi;
            life += hb[i + 6] - hb[i + 5];
        }
        const double g = h[6] ? (double)h[6] : 1.0;
        fprintf(stderr, "[blk] groups %llu rows %llu | cycles per group: meta %.0f clear+park-loads %.0f runs %.0f walk %.0f emit %.0f\n",
                h[6], h[7], h[0] / g, h[1] / g, h[2] / g, h[3] / g, h[4] / g);
        if (h[6] && t_hi > t_lo) {
            const double span = (double)(t_hi - t_lo);
            int dec[10] = {};
            for (size_t i = 0; i < hb.size(); i += 8) {
                if (hb[i + 6] == 0) continue;
                for (int q = 0; q < 10; q++) {
                    const double t = (double)t_lo + span * (q + 0.5) / 10.0;
                    dec[q] += (double)hb[i + 5] <= t && t < (double)hb[i + 6];
                }
            }
            fprintf(stderr, "[blk] first start to last end %.1f us (all window bins), mean life %.2f us, mean resident %.0f groups = %.2f per CU | resident at 5%%..95%%:",
                    span * 0.01, life / g * 0.01, life / span, life / span / cx.num_cus);
            for (int q = 0; q < 10; q++) fprintf(stderr, " %d", dec[q]);
            fprintf(stderr, "\n");
        }
        dev_free(blk_prof);
    }
#endif
    return L;
}

// nnz_max is a HINT here (it selects lane counts and whether long rows are deferred); upstream only
// ever prints it.  A caller-built sfCSR may leave it unset: anything outside (0, N] means unknown.
static inline sfCSR with_checked_hint(const sfCSR *m)
{
    sfCSR r = *m;
    if (!(r.nnz_max > 0 && r.nnz_max <= r.N)) r.nnz_max = 0;
    return r;
}

// How many 1024-thread workgroups are resident together on this device as this process sees it (CU masks,
// partitions, whatever else holds CUs right now): the fused tails put a grid barrier into an ordinary launch and
// are used only for grids that passed this census.  hipDeviceAttributeMultiprocessorCount knows nothing of
// HSA_CU_MASK / ROC_GLOBAL_CU_MASK.  One to a few launches of ~10 us, once per context.
static int census_coresident(Context &cx, hipStream_t st)
{
    int *d = cx.d_scratch + 8000;  // two ints behind the workgroup records of the fused tails
    int grid = cx.num_cus < kFusedMaxBlocks ? cx.num_cus : kFusedMaxBlocks;
    static DevOnce big_ok;       
    allow_big_lds(k_census, big_ok, kCensusLds, cx.device);  // one census workgroup per CU (fused.h)
    for (; grid >= 8; grid = grid * 3 / 4) {
        NSP_CHECK(hipMemsetAsync(d, 0, 2 * sizeof(int), st));
        hipLaunchKernelGGL(k_census, dim3(grid), dim3(1024), kCensusLds, st, d, 20000 /* 0.2 ms of 100 MHz ticks */);
        NSP_LAUNCH_CHECK();
        NSP_CHECK(hipMemcpyAsync(cx.h_pinned + 130, d, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
        NSP_CHECK(hipStreamSynchronize(st));
        if (cx.h_pinned[130] == grid && cx.h_pinned[131] == 0) return grid;
    }
    return 0;
}

// returns true when the call has to be repeated with the kernel chains (a grid barrier of the fused tails
// timed out: nothing of the call survives, the inputs are untouched)
static bool run_once(sfCSR *a_in, sfCSR *b_in, sfCSR *c, bool numeric_only)
{
    Context &cx = ctx();
    g_panel = PanelTable{};  // (whatever the last call built went back to the block cache with its launcher)
    sfCSR a_chk = with_checked_hint(a_in), b_chk = with_checked_hint(b_in);
    const sfCSR *a = &a_chk, *b = &b_chk;
    if (a->M <= 0 || a->nnz <= 0 || b->nnz <= 0 || b->M <= 0) {
        // nothing to multiply: C has a->M empty rows (zero-size grids are not launchable)
        memset(&g_stats.s, 0, sizeof(g_stats.s));
        g_stats.s.sym_bin_size[0] = g_stats.s.num_bin_size[0] = a->M > 0 ? a->M : 0;  // every row: nothing to do
        if (!numeric_only) {
            const int M0 = a->M > 0 ? a->M : 0;
            c->M = M0;
            c->N = b->N;
            c->nnz = 0;
            c->nnz_max = 0;
            c->d_rpt = (int *)dev_alloc(sizeof(int) * (size_t)(M0 + 1));
            c->d_col = (int *)dev_alloc(sizeof(int));
            c->d_val = (real *)dev_alloc(sizeof(real));
            NSP_CHECK(hipMemsetAsync(c->d_rpt, 0, sizeof(int) * (size_t)(M0 + 1), cx.stream[0]));
            NSP_CHECK(hipStreamSynchronize(cx.stream[0]));
        }
        return false;
    }
    Timer tm(cx);
    TraceRange phase_range("setup");
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
    int *tcol = nullptr;            // column lists of the bit-window rows (symbolic -> numeric, listed.h)
    long long *list_off = nullptr;  // per row: where its list starts in tcol, -1: none
    unsigned int *bm = nullptr;
    unsigned char *grp = nullptr;
    int4 *blk_desc = nullptr;  // row records of the node-block kernel, in list order (k_numeric_setup)
    int4 *sym_desc = nullptr;  // row records of k_sym_dense, in list order (k_setup_tail)
    BinLauncher sym_used(cx, 0);
    const int K = b->M;
    // The per-row arrays, the records of B, the partials and the twin map are ONE block of the cache, carved
    // up here: a lookup under the cache lock per array (seventeen of them) was host time at the very start
    // of the call, before the first kernel is even queued.
    static const bool twins_on = !(getenv("NSPARSE_TWINS") && atoi(getenv("NSPARSE_TWINS")) == 0);
    static const bool lean_on = exp_env("NSPARSE_WINDOW_KERNEL", 1) != 0;
    const bool find_twins = !numeric_only && twins_on && M > 1;
    const bool want_btwin = twins_on && lean_on && K > 1;
    unsigned int tsize = 1024;
    while (find_twins && tsize < 2u * (unsigned int)M) tsize <<= 1;
    const long long twin_fill_words = find_twins ? (long long)tsize + ((long long)M * (1 + kGroupMembers) + 1) / 2 : 0;
    // matrices of kTwinSampleMin rows and more: k_b_info samples the rows of A, k_row_products probes the pattern map
    // only if the sample holds a pattern twice (setup.h: TwinSample; NSPARSE_TWIN_SAMPLE=0: always probe)
    static const bool sample_on = exp_env("NSPARSE_TWIN_SAMPLE", 1) != 0;
    unsigned int sample_slots = 0;
    if (find_twins && sample_on && M >= kTwinSampleMin) {
        sample_slots = 1024;
        while (sample_slots < 2u * 64u * (unsigned int)((M + 1023) >> 10)) sample_slots <<= 1;  // load <= 1/2
    }
    // (with the cache switched off -- "reference-compatible" timing -- every array is its own hipMalloc as
    //  before: the runtime serves small blocks from pools, one block of megabytes is mapped for real and
    //  made that timing 1.17 -> 1.62 ms)
    const bool pooled = dev_cache_enabled();
    constexpr int kCarveMax = 20;
    size_t carve = 0, c_off[kCarveMax], c_sz[kCarveMax];
    int ncarve = 0;
    auto reserve = [&](size_t bytes) {
        c_off[ncarve] = carve;
        c_sz[ncarve] = bytes;
        carve += (bytes + 255) & ~(size_t)255;
        return ncarve++;
    };
    const size_t M1 = (size_t)(M > 0 ? M : 1);
    const int o_prod = reserve(sizeof(int) * (M1 + 1)), o_nz = reserve(sizeof(int) * (M1 + 1)),
              o_perm = reserve(sizeof(int) * M1), o_lo = reserve(sizeof(int) * M1),
              o_span = reserve(sizeof(int) * M1), o_maxb = reserve(sizeof(int) * M1),
              o_binfo = reserve(sizeof(BInfo) * (size_t)(K > 0 ? K : 1)), o_long = reserve(sizeof(int) * kLongCap),
              o_part = reserve(sizeof(long long) * kPartialStride * kSetupMaxGrid),
              o_bmw = reserve(sizeof(int) * (M1 + 1)), o_bmo = reserve(sizeof(int) * (M1 + 1)),
              o_spn = reserve(sizeof(int) * M1), o_btwin = reserve(want_btwin ? (size_t)K : 0),
              o_table = reserve(sizeof(unsigned long long) * (size_t)twin_fill_words),
              o_twof = reserve(find_twins ? sizeof(int) * M1 : 0), o_twin = reserve(find_twins ? M1 : 0),
              o_stab = reserve(sizeof(unsigned long long) * (size_t)sample_slots);
    char *block_base = pooled ? (char *)dev_alloc(carve) : nullptr;
    char *c_ptr[kCarveMax];
    for (int q = 0; q < ncarve; q++)
        c_ptr[q] = pooled ? block_base + c_off[q] : (c_sz[q] ? (char *)dev_alloc(c_sz[q]) : nullptr);
    int *row_prod = (int *)(c_ptr[o_prod]);
    int *row_nz = (int *)(c_ptr[o_nz]);
    int *row_perm = (int *)(c_ptr[o_perm]);
    int *row_lo = (int *)(c_ptr[o_lo]);
    int *row_span = (int *)(c_ptr[o_span]);
    int *row_maxb = (int *)(c_ptr[o_maxb]);
    BInfo *binfo = (BInfo *)(c_ptr[o_binfo]);
    int *long_list = (int *)(c_ptr[o_long]);  // reused: B rows first, then A rows
    int *long_cnt = cx.d_scratch + 240;                         // [0] B pass, [1] A pass
    if (g_dense_enabled < 0) {
        const char *e = getenv("NSPARSE_DENSE");
        g_dense_enabled = !(e && e[0] == '0');
    }
    Thr sym_thr = sym_ladder(), num_thr = num_ladder();
    if (!g_dense_enabled) sym_thr.dense_ratio = num_thr.dense_ratio = sym_thr.bits_ratio = num_thr.rank_span = 0;
    NSP_CHECK(hipStreamSynchronize(0));  // inputs queued on the null stream by the caller
    // one fill for both counter blocks and the four words at long_cnt (the two list counters and the
    // two words of k_col_range); the ints in between belong to calls that reset them themselves
    static_assert(2 * sizeof(BinState) <= 240 * sizeof(int), "counter blocks end before long_cnt");
    // (+ the four words of the fused tails).  A call that ran to its end has left them zeroed (k_finish).
    if (!cx.counters_clean) NSP_CHECK(hipMemsetAsync(cx.d_scratch, 0, 248 * sizeof(int), s0));
    cx.counters_clean = false;

    // rows of B with the column pattern of the row before them (k_b_info): runs of the numeric
    // window kernel (block.h).  NSPARSE_TWINS=0 switches the whole twin machinery off.
    unsigned char *btwin = want_btwin ? (unsigned char *)(c_ptr[o_btwin]) : nullptr;
    // rows with the column pattern of another row are not run through the symbolic phase: they take that
    // row's result (twin_probe / k_twin_copy).  NSPARSE_TWINS=0 switches the detection off.
    unsigned char *twin = nullptr;
    int *twin_of = nullptr, *fcnt = nullptr, *members = nullptr;
    unsigned long long *ttable = nullptr;
    TwinMap tw = {nullptr, 0u, 0, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (find_twins) {
        // table, sign-up counters and members side by side, one fill (all ones = free / -1 / none;
        // k_b_info fills them on its way, in 64-bit words)
        ttable = (unsigned long long *)(c_ptr[o_table]);
        fcnt = (int *)(ttable + tsize);
        members = fcnt + M;
        twin_of = (int *)(c_ptr[o_twof]);
        twin = (unsigned char *)(c_ptr[o_twin]);
        tw = TwinMap{ttable, tsize - 1, a->nnz, twin_of, twin, fcnt, members,
                     sample_slots ? (const int *)&d_sym->twin_sample : (const int *)nullptr};
    }
    TwinSample tsamp = {nullptr, nullptr, 0, nullptr, 0u, 0u};
    if (sample_slots)
        tsamp = TwinSample{a->d_rpt, a->d_col, M, (unsigned long long *)(c_ptr[o_stab]), sample_slots - 1,
                           1u + (unsigned int)(cx.seq % 65535)};
    // ---- setup: column window of every B row, products + window per C row, symbolic bins ----
    const bool same_shape = find_twins && lean_on && M == K && a->N == b->N && a->nnz == b->nnz;
    {
        const int wb = pick_w_regular(b->nnz, K, b->nnz_max);
        int gb = ceil_div((long long)K * wb, 256);
        int *blist = (b->nnz_max > 0 && b->nnz_max <= kLongFactor * wb) ? nullptr : long_list;
        // A with fewer rows than B: a row block of a partitioned product (B replicated).  Then only
        // the rows of B its columns reach get a record -- for a banded matrix the block's own
        // stretch, so the set-up cost does not grow with the number of ranks.
        unsigned int *range = nullptr, *range_part = nullptr;
        if (M < K && a->nnz > 0) {
            range = reinterpret_cast<unsigned int *>(long_cnt + 2);
            const int nparts = 1024;
            range_part = (unsigned int *)dev_alloc(sizeof(unsigned int) * 2 * nparts);
            hipLaunchKernelGGL(k_col_range, dim3(nparts), dim3(256), 0, s0, a->d_col, a->nnz, range_part);
            hipLaunchKernelGGL(k_col_range_fold, dim3(1), dim3(256), 0, s0, range_part, nparts, range);
            // the stretch is not known here: a grid for all of B would mostly be workgroups that
            // find nothing to do (25 us of them at 8 ranks), so a bounded grid strides over it
            const int gcap = ceil_div((long long)M * wb, 256) + 1024;
            gb = gb < gcap ? gb : gcap;
        }
#define NSP_BI(W)                                                                              \
    case W:                                                                                    \
        hipLaunchKernelGGL(k_b_info<W>, dim3(gb), dim3(256), 0, s0, b->d_rpt, b->d_col, K, binfo, d_sym, \
                           blist, long_cnt, kLongFactor * W, (const int *)nullptr, range, btwin, ttable, twin_fill_words, tsamp); \
        break;
        switch (wb) {
            NSP_BI(1) NSP_BI(2) NSP_BI(4) NSP_BI(8) NSP_BI(16) NSP_BI(32) NSP_BI(64)
        }
#undef NSP_BI
        if (blist)
            hipLaunchKernelGGL(k_b_info<64>, dim3(256), dim3(256), 0, s0, b->d_rpt, b->d_col, K, binfo, d_sym,
                               (int *)nullptr, long_cnt, 0, (const int *)long_list, (const unsigned int *)nullptr, btwin,
                               (unsigned long long *)nullptr, 0LL);
        if (range_part) dev_free(range_part);  // stream-ordered reuse, see scan_exclusive
    }
    long long *partial = (long long *)(c_ptr[o_part]);
    // column bitmaps handed from the symbolic to the numeric dense kernels
    int *bm_words = (int *)(c_ptr[o_bmw]);
    int *bm_off = (int *)(c_ptr[o_bmo]);
    int *row_span_num = (int *)(c_ptr[o_spn]);
    const bool use_bm = !numeric_only && num_thr.dense_ratio > 0;
    // matrices of up to 1 M rows: the helper chains behind the big kernels are one launch each (fused.h)
    static const bool fused_on = !(getenv("NSPARSE_FUSED") && atoi(getenv("NSPARSE_FUSED")) == 0);
    // rows per thread of the fused tails: one up to 256 K rows, four up to 1 M (NSPARSE_FUSED_BIG=0: chains beyond 256 K)
    static const bool fused_big = exp_env("NSPARSE_FUSED_BIG", 1) != 0;
    const int frows = (M + 1 <= kFusedMaxBlocks * 1024 || !fused_big) ? 1 : 4;
    const int fgrid = ceil_div(M + 1, 1024 * frows);
    // (one 1024-thread workgroup per CU at most: the grid barrier needs all of them resident, also on a
    //  partitioned or CU-masked device)
    // NSPARSE_FUSED_FORCE=1 (tests): skip the census, so that a CU-masked device exercises the time-out path
    static const bool fused_force = exp_env("NSPARSE_FUSED_FORCE", 0) == 1;
    if (fused_on && !numeric_only && cx.coresident < 0 && fgrid <= kFusedMaxBlocks)
        cx.coresident = fused_force ? kFusedMaxBlocks : census_coresident(cx, s0);
    const bool fuse = fused_on && cx.fused_ok && !numeric_only && fgrid <= kFusedMaxBlocks && fgrid <= cx.coresident;
    bool retry = false, c_rpt_ours = false;
    char *btw_block = nullptr;     // pattern map of the rows of B (k_b_twins), general B only
    int *b_twin_of = nullptr;
    bool b_twins_pending = false;
    const int nparts = launch_row_products(a, b, binfo, row_prod, row_lo, row_span, bm_words,
                        use_bm ? (num_thr.rank_span > num_thr.dense_span[2] ? num_thr.rank_span : num_thr.dense_span[2]) : 0, sym_thr, d_sym, partial, row_span_num, row_nz, row_maxb, long_list, long_cnt + 1, tw, !fuse, s0);
    void *bm_scan_tmp = nullptr;
    const int grid_m = ceil_div(M, 1024);
    // (row records: not with the cache off -- two more megabyte-sized hipMalloc / hipFree pairs cost more
    //  than the round trips they save)
    if (pooled && fuse && frows == 1 && use_bm && exp_env("NSPARSE_BLK_DESC", 1) != 0)
        sym_desc = (int4 *)dev_alloc(sizeof(int4) * 3 * (size_t)M);
    if (fuse) {
        const int seq = ++cx.seq;
        const FusedSync fs = {cx.d_scratch + 244, cx.d_scratch + 512, cx.d_mapped, cx.d_mapped + 120, seq, cx.d_mapped + 123};
#define NSP_SETUP_TAIL(RX)                                                                        \
        hipLaunchKernelGGL(k_setup_tail<RX>, dim3(fgrid), dim3(1024), 0, s0, (const long long *)partial, nparts, d_sym, \
                           use_bm ? (const int *)bm_words : (const int *)nullptr, bm_off, (const int *)row_prod,  \
                           (const int *)row_span, M, sym_thr, row_perm, (const unsigned char *)twin, fs, sym_desc, \
                           (const int *)a->d_rpt, (const int *)row_lo, (const int *)row_maxb)
        if (frows == 1) NSP_SETUP_TAIL(1); else NSP_SETUP_TAIL(4);
#undef NSP_SETUP_TAIL
        NSP_LAUNCH_CHECK();
        tm.mark(1, s0);
        // two copies of one matrix (C = A * A as the reference's sample calls it)?  Compared while the host is
        // busy with the flag; the answer (d_num->ab_differ) travels with the second publish.
        if (same_shape && !(a->d_rpt == b->d_rpt && a->d_col == b->d_col)) {
            int gc = ceil_div(a->nnz, 4 * 256);
            gc = gc < 1 ? 1 : (gc > 2048 ? 2048 : gc);
            hipLaunchKernelGGL(k_ab_compare, dim3(gc), dim3(256), 0, s0, (const int *)a->d_rpt, (const int *)a->d_col,
                               (const int *)b->d_rpt, (const int *)b->d_col, M, a->nnz, d_num, tw.sample_flag);
            NSP_LAUNCH_CHECK();
        }
        wait_published(120, seq, s0);
        retry = __atomic_load_n(cx.h_mapped + 123, __ATOMIC_ACQUIRE) == seq;
    } else {
    if (use_bm) bm_scan_tmp = scan_exclusive(bm_words, bm_off, M + 1, s0);
    if (!numeric_only) {
        if (M >= (1 << 18))
            hipLaunchKernelGGL(k_bin_scatter<4>, dim3(ceil_div(M, 4096)), dim3(1024), 0, s0, row_prod, row_span,
                               (const int *)nullptr, M, sym_thr, d_sym, row_perm, (const unsigned char *)twin, 0xff);
        else
            hipLaunchKernelGGL(k_bin_scatter<1>, dim3(grid_m), dim3(1024), 0, s0, row_prod, row_span,
                               (const int *)nullptr, M, sym_thr, d_sym, row_perm, (const unsigned char *)twin, 0xff);
        NSP_LAUNCH_CHECK();
    }
    {
        const int seq = ++cx.seq;
        hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, s0, d_sym, cx.d_mapped, (int)(sizeof(BinState) / 4),
                           (const int *)nullptr, cx.d_mapped + 120, seq);
        NSP_LAUNCH_CHECK();
        tm.mark(1, s0);  // issued before the host waits: everything between the flag and the first
                         // symbolic launch is GPU idle time
        wait_published(120, seq, s0);
    }
    }
    if (!retry) {
    S.n_prod = h_sym->total;
    S.max_prod_row = h_sym->maxv;
    for (int q = 0; q < NB; q++) S.sym_bin_size[q] = h_sym->hist[q];

    // ---- symbolic: nnz of every row of C, then C.rpt ----------------------------------
    phase_range.next("symbolic");
    if (!numeric_only) {
        c->M = M;
        c->N = b->N;
        // bitmaps only when they fit int offsets comfortably; otherwise the dense window is
        // used by the symbolic phase alone and the numeric phase hashes
        if (use_bm && h_sym->bm_total > 0 && h_sym->bm_total < (1LL << 30))
            bm = (unsigned int *)dev_alloc(sizeof(unsigned int) * (size_t)h_sym->bm_total);
        // A * B with B != A on a matrix whose rows share column patterns (a row block of a partitioned C = A * A):
        // pattern leaders of the rows of B for the keyed runs of the node-block kernel, on a side stream beside
        // the symbolic phase (setup.h: k_b_twins)
        {
            long long binned0 = 0;
            for (int q = 0; q < NB; q++) binned0 += h_sym->hist[q];
            static const bool keyed_b_on = exp_env("NSPARSE_KEYED", 1) != 0 && exp_env("NSPARSE_KEYED_B", 1) != 0;
            if (keyed_b_on && find_twins && lean_on && fuse && !same_shape && K > 1 && (M - binned0) * 8 >= M) {
                unsigned int tsb = 1024;
                while (tsb < 2u * (unsigned int)K) tsb <<= 1;
                const size_t words64 = (size_t)tsb + ((size_t)K * (1 + kGroupMembers) + 1) / 2;  // table | fcnt | members
                const size_t bytes = sizeof(unsigned long long) * words64 + sizeof(int) * (size_t)K + (size_t)K;
                btw_block = (char *)dev_alloc(bytes);
                unsigned long long *tb = (unsigned long long *)btw_block;
                int *fc = (int *)(tb + tsb);
                b_twin_of = (int *)(btw_block + sizeof(unsigned long long) * words64);
                const TwinMap tmb = {tb, tsb - 1, b->nnz, b_twin_of, (unsigned char *)(b_twin_of + K), fc, fc + K, nullptr};
                hipStream_t sb = cx.stream[kMaxBins - 1];  // (no bin of either ladder runs there)
                NSP_CHECK(hipEventRecord(cx.ev_fork, s0));
                NSP_CHECK(hipStreamWaitEvent(sb, cx.ev_fork, 0));
                NSP_CHECK(hipMemsetAsync(btw_block, 0xff, bytes, sb));
                const unsigned int *rg = (M < K && a->nnz > 0) ? reinterpret_cast<const unsigned int *>(long_cnt + 2)
                                                               : (const unsigned int *)nullptr;
                const int wb = pick_w_regular(b->nnz, K, b->nnz_max);
                int gb = ceil_div((long long)(M < K ? M + 4096 : K) * wb, 256);
                gb = gb < 1 ? 1 : (gb > 16384 ? 16384 : gb);
#define NSP_BT(W)                                                                              \
    case W:                                                                                    \
        hipLaunchKernelGGL(k_b_twins<W>, dim3(gb), dim3(256), 0, sb, b->d_rpt, b->d_col, K, rg, tmb); \
        break;
                switch (wb) {
                    NSP_BT(1) NSP_BT(2) NSP_BT(4) NSP_BT(8) NSP_BT(16) NSP_BT(32) NSP_BT(64)
                }
#undef NSP_BT
                NSP_LAUNCH_CHECK();
                NSP_CHECK(hipEventRecord(cx.ev_join[kMaxBins - 1], sb));
                b_twins_pending = true;
            }
        }
        // column lists of the bit-window rows for the numeric listed kernel (listed.h): a slab with room for
        // min(products, window) entries per such row, when that fits comfortably (NSPARSE_LIST=0: off)
        const bool list_on = list_mode() > 0;
        const bool cursor_sym = h_sym->hist[kBitsBin0 + 1] > 0 && h_sym->max_span[kBitsBin0 + 1] > (1 << 20);  // symbolic_phase's rule
        if (list_on && h_sym->list_total > 0 && h_sym->b_unsorted == 0 &&
            cursor_sym) {
            size_t free_b = 0, total_b = 0;
            const size_t want = sizeof(int) * (size_t)h_sym->list_total + sizeof(long long) * (size_t)M;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && want <= free_b / 3 + (pooled ? (64u << 20) : 0u) &&
                want <= ((size_t)64 << 30)) {
                tcol = (int *)dev_alloc(sizeof(int) * (size_t)h_sym->list_total);
                list_off = (long long *)dev_alloc(sizeof(long long) * (size_t)M);
                NSP_CHECK(hipMemsetAsync(list_off, 0xff, sizeof(long long) * (size_t)M, s0));
            }
        }
        BinLauncher LS = symbolic_phase(a, b, row_prod, row_maxb, row_lo, row_span, row_nz, row_perm, h_sym->hist,
                                        h_sym->maxv, d_sym, cx, S.ms_sym_bin, &S.sym_fail_rows,
                                        bm_off, bm, row_span_num, h_sym->max_span, (int)h_sym->max_alen,
                                        h_sym->b_unsorted == 0, btwin, bm ? (const int4 *)sym_desc : (const int4 *)nullptr,
                                        tcol, list_off);
        sym_used = LS;
        {
            long long binned = 0;
            for (int q = 0; q < NB; q++) binned += h_sym->hist[q];
            S.twin_rows = twin ? (int)(M - binned) : 0;
            // groups of twin rows for the numeric window kernel: worth their LDS only when a good share
            // of the rows has a twin (a finite-element matrix), not for a few chance repeats
            const bool want_grp = lean_on && bm && (long long)S.twin_rows * 8 >= M;
            if (want_grp) grp = (unsigned char *)dev_alloc((size_t)M);
            if (!fuse) {
                if (S.twin_rows > 0) {
                    hipLaunchKernelGGL(k_twin_copy, dim3(ceil_div(M, 256)), dim3(256), 0, s0, (const int *)twin_of, M,
                                       row_nz, row_span_num, bm ? bm_off : (int *)nullptr);
                    NSP_LAUNCH_CHECK();
                }
                if (want_grp) {
                    hipLaunchKernelGGL(k_twin_groups, dim3(ceil_div(M, 256)), dim3(256), 0, s0, (const int *)twin_of,
                                       (const int *)members, (const int *)row_span_num,
                                       (const int *)row_nz, (const int *)row_prod, num_thr, M, grp);
                    NSP_LAUNCH_CHECK();
                }
            }
        }
        c->d_rpt = (int *)dev_alloc(sizeof(int) * (size_t)(M + 1));
        c_rpt_ours = true;
        if (!fuse) scan_tmp = scan_exclusive(row_nz, c->d_rpt, M + 1, s0);
    } else {
        // structure given: row_nz[i] = rpt[i+1] - rpt[i] is recovered inside the kernels
        // from C.rpt; for binning we need it explicitly.
        hipLaunchKernelGGL(k_row_len, dim3(ceil_div(M, 256)), dim3(256), 0, s0, c->d_rpt, row_nz, M);
        NSP_LAUNCH_CHECK();
    }
    tm.mark(2, s0);

    // ---- numeric binning ------------------------------------------------------------
    phase_range.next("numeric");
    // numeric window: full call -> rows whose bitmap was written; re-run -> every eligible row
    const int *num_span = numeric_only ? row_span : row_span_num;
    if (!numeric_only && bm == nullptr) num_thr.dense_ratio = num_thr.rank_span = 0;
    if (pooled && fuse && frows == 1 && bm && lean_on && exp_env("NSPARSE_BLK_DESC", 1) != 0)
        blk_desc = (int4 *)dev_alloc(sizeof(int4) * 3 * (size_t)M);
    if (fuse) {
        // twins' results, groups, C.rpt, histogram, permutation and the publish in one launch (fused.h)
        const int seq = ++cx.seq;
        const FusedSync fs = {cx.d_scratch + 246, cx.d_scratch + 512 + kFusedMaxBlocks * kFusedRec,
                              reinterpret_cast<int *>(h_num_dev), cx.d_mapped + 121, seq, cx.d_mapped + 123};
#define NSP_NUM_SETUP(RX)                                                                         \
        hipLaunchKernelGGL(k_numeric_setup<RX>, dim3(fgrid), dim3(1024), 0, s0,                                    \
                           S.twin_rows > 0 ? (const int *)twin_of : (const int *)nullptr, (const int *)members, row_nz, \
                           row_span_num, bm ? bm_off : (int *)nullptr, (const int *)row_prod, M, num_thr, d_num,     \
                           c->d_rpt, row_perm, grp, fs, blk_desc, (const int *)a->d_rpt, (const int *)row_lo,       \
                           (const int *)row_maxb)
        if (frows == 1) NSP_NUM_SETUP(1); else NSP_NUM_SETUP(4);
#undef NSP_NUM_SETUP
        NSP_LAUNCH_CHECK();
        wait_published(121, seq, s0);
        retry = __atomic_load_n(cx.h_mapped + 123, __ATOMIC_ACQUIRE) == seq;
    } else {
    hipLaunchKernelGGL(k_hist, dim3(grid_m < 128 ? grid_m : 128), dim3(1024), 0, s0, row_nz, num_span,
                       (const int *)row_prod, M, num_thr, d_num);
    if (M >= (1 << 18))
        hipLaunchKernelGGL(k_bin_scatter<4>, dim3(ceil_div(M, 4096)), dim3(1024), 0, s0, row_nz, num_span,
                           (const int *)row_prod, M, num_thr, d_num, row_perm, (const unsigned char *)grp, 3);
    else
        hipLaunchKernelGGL(k_bin_scatter<1>, dim3(grid_m), dim3(1024), 0, s0, row_nz, num_span,
                           (const int *)row_prod, M, num_thr, d_num, row_perm, (const unsigned char *)grp, 3);
    NSP_LAUNCH_CHECK();
    {
        const int seq = ++cx.seq;
        hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, s0, d_num, reinterpret_cast<int *>(h_num_dev),
                           (int)(sizeof(BinState) / 4), c->d_rpt + M, cx.d_mapped + 121, seq);
        NSP_LAUNCH_CHECK();
        wait_published(121, seq, s0);
    }
    }
    }  // !retry (first barrier)
    if (retry) {
        // A grid barrier timed out (CUs taken away since the census): wait for the stragglers of that launch,
        // drop everything this call has built -- the inputs are untouched -- and let the caller repeat it with
        // the kernel chains.  From now on this context does not fuse.
        NSP_CHECK(hipStreamSynchronize(s0));
        cx.fused_ok = false;
        sym_used.collect(S.ms_sym_bin);
        if (c_rpt_ours) {
            dev_free(c->d_rpt);
            c->d_rpt = nullptr;
        }
        too_big = false;
        cx.counters_clean = false;  // the next attempt starts from zeroed counter blocks
    } else {
    for (int q = 0; q < NB; q++) S.num_bin_size[q] = h_num->hist[q];
    S.max_nnz_row = h_num->maxv;
    if (!numeric_only) {
        // upstream keeps nnz(C) and C.rpt in int (nsparse.h:62-75) and wraps silently; refuse
        too_big = h_num->total > 0x7fffffffLL;
    }
    if (!numeric_only && !too_big) {
        c->nnz = h_num->nnz;
        c->nnz_max = h_num->maxv;  // longest row of C: the hint a chained product or sf_csr2amb reads
        c->d_col = (int *)dev_alloc(sizeof(int) * (size_t)(c->nnz > 0 ? c->nnz : 1));
        c->d_val = (real *)dev_alloc(sizeof(real) * (size_t)(c->nnz > 0 ? c->nnz : 1));
    }
    S.nnz_c = too_big ? h_num->total : c->nnz;

    // C = A * A on a matrix whose twin rows are mostly NOT neighbours (k_numeric_setup counted them): the
    // node-block kernel builds its runs of B rows from the pattern leaders instead of from neighbouring entries
    static const bool keyed_on = exp_env("NSPARSE_KEYED", 1) != 0;
    const int *bkey = nullptr;
    if (keyed_on && fuse && grp && twin_of && (long long)h_num->far_twins * 4 > (long long)S.twin_rows) {
        if (same_shape && h_num->ab_differ == 0) bkey = twin_of;  // B is A: A's own pattern leaders
        else if (b_twin_of) bkey = b_twin_of;                     // general B: its own (k_b_twins)
    }
    if (b_twins_pending) NSP_CHECK(hipStreamWaitEvent(s0, cx.ev_join[kMaxBins - 1], 0));
    // ---- numeric --------------------------------------------------------------------
    // (a numeric-only re-run has the list of every row: C.col itself)
    if (!too_big) {
    BinLauncher LN = numeric_phase(a, b, c, row_prod, row_maxb, row_lo, row_span, row_perm, h_num->hist,
                                   h_num->maxv, d_num, cx, S.ms_num_bin,
                                   numeric_only ? 0 : (g_sorted ? 1 : 3), bm_off, bm,
                                   (int)h_sym->max_alen, h_sym->b_unsorted == 0, h_num->max_span, grp, btwin, h_num->cursor, members, blk_desc, bkey,
                                   numeric_only ? (list_mode() > 0 ? (const int *)c->d_col : (const int *)nullptr) : (const int *)tcol, list_off,
                                   0LL);
    if (g_deterministic && c->nnz > 0) {
        // (behind the join of the bins on the main stream: it overwrites what they wrote)
        hipLaunchKernelGGL(k_num_deterministic, dim3(ceil_div((long long)M * 64, 256)), dim3(256), 0, s0, a->d_rpt, a->d_col,
                           a->d_val, b->d_rpt, b->d_col, b->d_val, (const int *)c->d_rpt, (const int *)c->d_col, c->d_val, M,
                           h_sym->b_unsorted == 0 ? 1 : 0);
        NSP_LAUNCH_CHECK();
    }
    tm.mark(3, s0);
    {   // synchronous on return, like upstream (:1287): poll a flag raised behind the last kernel
        const int seq = ++cx.seq;
        hipLaunchKernelGGL(k_finish, dim3(1), dim3(256), 0, s0, cx.d_scratch, cx.d_mapped + 122, seq);
        NSP_LAUNCH_CHECK();
        wait_published(122, seq, s0);
        cx.counters_clean = true;
    }
    LN.collect(S.ms_num_bin);
    sym_used.collect(S.ms_sym_bin);
    S.ms_setup = tm.ms(0, 1);
    S.ms_symbolic = tm.ms(1, 2);
    S.ms_numeric = tm.ms(2, 3);
    S.ms_total = tm.ms(0, 3);
    } else {
        NSP_CHECK(hipStreamSynchronize(s0));
        sym_used.collect(S.ms_sym_bin);  // returns the scratch the symbolic kernels deferred
        dev_free(c->d_rpt);
        c->d_rpt = nullptr;
        c->d_col = nullptr;
        c->d_val = nullptr;
        c->nnz = 0;
    }
    }  // !retry

    dev_free(scan_tmp);
    if (btw_block) {
        if (retry || too_big) NSP_CHECK(hipStreamSynchronize(cx.stream[kMaxBins - 1]));  // (else the call drained behind the join)
        dev_free(btw_block);
    }
    dev_free(tcol);
    dev_free(list_off);
    dev_free(bm);
    dev_free(bm_scan_tmp);
    if (grp) dev_free(grp);
    if (blk_desc) dev_free(blk_desc);
    if (sym_desc) dev_free(sym_desc);
    if (pooled) {
        dev_free(block_base);
    } else {
        for (int q = 0; q < ncarve; q++)
            if (c_ptr[q]) dev_free(c_ptr[q]);
    }
    if (too_big) {
        char msg[160];
        snprintf(msg, sizeof(msg), "nnz(C) = %lld does not fit the int row pointers of sfCSR", (long long)S.nnz_c);
        set_error(-40, msg, __FILE__, __LINE__);
    }
    return retry;
}

static void run(sfCSR *a_in, sfCSR *b_in, sfCSR *c, bool numeric_only)
{
    ApiLock api_lock;
    CallScope call_scope;
    TraceRange range(numeric_only ? "nsparse:spgemm_numeric" : "nsparse:spgemm");
    clear_error();
    if (run_once(a_in, b_in, c, numeric_only)) {
        g_stats.fused_fallbacks++;
        (void)run_once(a_in, b_in, c, numeric_only);  // fused_ok is false now: this one cannot ask again
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

void nsparse_get_spgemm_stats(nsparse_spgemm_stats *out)
{
    nsp::ApiLock lk;
    *out = nsp::spgemm::g_stats.s;
}

int nsparse_set_deterministic(int on)
{
    nsp::ApiLock lk;
    const int old = nsp::spgemm::g_deterministic;
    nsp::spgemm::g_deterministic = on ? 1 : 0;
    return old;
}

int nsparse_fused_state(int *coresident, int *fallbacks)
{
    nsp::ApiLock lk;
    if (!nsp::ctx_ready()) {  // a query creates nothing: no context on this device yet (or no such device)
        if (coresident) *coresident = -1;
        if (fallbacks) *fallbacks = nsp::spgemm::g_stats.fused_fallbacks;
        return -1;
    }
    nsp::Context &cx = nsp::ctx();
    if (coresident) *coresident = cx.coresident;
    if (fallbacks) *fallbacks = nsp::spgemm::g_stats.fused_fallbacks;
    return cx.fused_ok ? 1 : 0;
}

int nsparse_spgemm_set_sorted(int on)
{
    const int old = nsp::spgemm::g_sorted;
    nsp::spgemm::g_sorted = on ? 1 : 0;
    return old;
}

void nsparse_get_spgemm_bins(int *sym, int *num)
{
    // 18 ints each: tiny, hash_t[4], dense_span[3], dense_ratio, bits_span[2], bits_ratio,
    // bits_min, bits_wide_min, bits_wide_span, rank_span, rank_ratio, rank_max_nz (ratios / rank_span are 0 when NSPARSE_DENSE=0)
    const nsp::spgemm::Thr *t[2] = {&nsp::spgemm::sym_ladder(), &nsp::spgemm::num_ladder()};
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
        out[p][15] = nsp::spgemm::g_dense_enabled ? t[p]->rank_span : 0;
        out[p][16] = t[p]->rank_ratio;
        out[p][17] = t[p]->rank_max_nz;
    }
}

void get_spgemm_flop(sfCSR *a, sfCSR *b, int M, long long int *flop)
{
    nsp::ApiLock lk;
    nsp::clear_error();
    *flop = 0;
    if (M <= 0) return;
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
