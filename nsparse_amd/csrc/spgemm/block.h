// spgemm/block.h -- numeric window kernel that works on NODE BLOCKS (bins 6-8, default path).
// Part of the spgemm_hash.hip translation unit (kernels are launched from its host code).
//
// What measurements of the first numeric window kernel (window.h: k_num_dense) showed on the
// cant-class brick (ablations): without its LDS atomics the kernel takes the SAME time;
// without the loads of B it drops from 0.46 to 0.33 ms; without both, still 0.32 ms -- 0.18 ms of pure
// instruction stream of the product walk and 0.13 ms of per-row latency (clear, A-entry parking,
// emission).  It is bound by instructions per product and by bytes through the 64 B/clk L1, not by
// ds_add_f64 (tools/lds_atomic: 5-20 lane-operations per clock per CU; the kernel needed 1.3).
//
// So this kernel cuts the instructions and the bytes per product, by the structure finite-element
// matrices have: the degrees of freedom of one mesh node are rows with ONE column pattern ("twin
// rows", found by pattern for the symbolic phase -- neighbours or not).
//   * up to 3 twin rows of A are one workgroup: the A entries are parked once, every element of B is
//     loaded once and used for all of them (3 windows in LDS);
//   * A entries whose rows of B are twins (the 3 dof of the node the entry points to: neighbouring
//     entries, or -- KEYED, C = A * A on a scattered numbering -- entries grouped by pattern leader)
//     are one RUN: the lane that holds column p of the run loads the column id once and the values of
//     the (up to 3) rows, forms  sum_d a[r][d] * b[d][p]  in registers and issues ONE ds_add_f64 per
//     C row -- a 3 x 3 node block costs 4 loads, 9 multiply-adds and 3 atomics for 9 products,
//     where the generic walk needed 9 x (load share + 6 slot instructions + multiply + atomic);
//   * one lane per COLUMN of the B row (a group of G lanes reads G consecutive entries), so an atomic
//     instruction sees consecutive columns: the window is indexed plainly, a partial chunk is one
//     exec mask.
// Matrices without twin rows run the same code with runs and groups of one (the lean scalar walk).
// Structure and values are the reference's (kernel_spgemm_hash_d.cu:829-927 accumulates the same
// products with shared-memory atomics); only the order of the floating-point additions differs, as it
// already does between two runs of the reference.
#pragma once
#include "common.h"
#include "window.h"

namespace nsp {
namespace spgemm {

constexpr int kBlkRows = 3;        // twin rows of A per workgroup
constexpr int kBlkRun = 3;         // twin rows of B per run
constexpr int kBlkCols = 2;        // columns of a chunk per lane
constexpr int kBlkAccElems = 4608;  // LDS budget of the accumulator rows of one workgroup (36 KiB)

// grp[r]: bits 0-1 = position of row r inside its group (0: head), bits 2-3 = rows in the group (heads).
// A group = a pattern leader and the lowest / highest twin that signed up with it (twin_probe:
// members[]), as many of them as accumulator rows of this pattern's nnz fit the LDS budget; later twins of the same leader and rows
// outside the numeric window bins are groups of one.  Every row decides for itself from its leader's
// numbers, so heads and followers agree without talking.
__global__ __launch_bounds__(256) void k_twin_groups(const int *__restrict__ twin_of,
                                                     const int *__restrict__ members,
                                                     const int *__restrict__ row_span_num,
                                                     const int *__restrict__ row_nz,
                                                     const int *__restrict__ row_prod, Thr num_thr, int M,
                                                     unsigned char *__restrict__ grp)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= M) return;
    const int l = twin_of[r];
    const int lead = l >= 0 ? l : r;
    const int span = row_span_num[lead];
    const int nzs = ((row_nz[lead] + 7) >> 3) << 3;
    // only rows the numeric binning will put into a window bin (the same rule, bin_of): the followers
    // of a group are left out of the bin lists, which only the node-block kernel understands
    const bool windowed = span > 0 && nzs > 0 && bin_of(row_nz[lead], span, num_thr, row_prod[lead]) >= kDenseBin0;
    int cap = windowed ? kBlkAccElems / nzs : 1;
    cap = cap < 1 ? 1 : (cap > kBlkRows ? kBlkRows : cap);
    const int m0 = members[kGroupMembers * lead], m1 = members[kGroupMembers * lead + 1];
    const int nf = m0 < 0 ? 0 : (m1 != m0 ? 2 : 1);  // the lowest and the highest follower that signed up
    const int gsize = 1 + nf < cap ? 1 + nf : cap;  // rows in the leader's group
    int code = 1 << 2;                               // a group of one
    if (l < 0) {
        code = gsize << 2;
    } else {
        const int pos = r == m0 ? 1 : (r == m1 ? 2 : 0);  // which member am I?  (other twins: none)
        if (pos > 0 && pos < gsize) code = pos;
    }
    grp[r] = (unsigned char)code;
}

template <int BS, int SPAN_MAX, int MODE, int U, bool KEYED = false>
// (six wavefronts per SIMD for the window bins; the ranked-window instance -- SPAN_MAX 65536, two bitmap words more per
// lane in registers -- reaches four, which is what is asked of it)
__global__ __launch_bounds__(BS) __attribute__((amdgpu_waves_per_eu(BS <= 256 ? (SPAN_MAX > 12288 ? 4 : 6) : 2))) void k_num_block(const int *__restrict__ arpt, const int *__restrict__ acol,
                                                  const real *__restrict__ aval,
                                                  const int *__restrict__ brpt, const int *__restrict__ bcol,
                                                  const real *__restrict__ bval,
                                                  const int *__restrict__ crpt, int *__restrict__ ccol,
                                                  real *__restrict__ cval,
                                                  const int *__restrict__ row_perm,
                                                  const int *__restrict__ row_maxb,
                                                  const int *__restrict__ row_lo,
                                                  const int *__restrict__ row_span, int bin_off,
                                                  int bin_size, int bnnz,
                                                  const int *__restrict__ bm_off,
                                                  const unsigned int *__restrict__ bm,
                                                  const unsigned char *__restrict__ grp,
                                                  const unsigned char *__restrict__ btwin,
                                                  unsigned long long *__restrict__ prof,
                                                  const int *__restrict__ members,
                                                  const int4 *__restrict__ desc,
                                                  const int *__restrict__ bkey, int seg1 = 0x7fffffff,
                                                  int bin_off2 = 0)
{
    // seg1 / bin_off2: the launch covers TWO stretches of the row list -- the first seg1 rows at bin_off, the rest
    // at bin_off2 (two neighbouring window bins folded into one launch: spgemm_hash.hip, NSPARSE_FOLD_WIN).
    // KEYED (C = A * A with twin rows that are NOT neighbours): bkey[c] = pattern leader of row c of B or
    // -1; the entries of an A row whose rows of B share a pattern form a run wherever they sit in the row
    // (grouped by counting among the parked entries), instead of only when they are neighbours.
    // desc != nullptr: k_numeric_setup left a 48-byte record per listed row, IN LIST ORDER (fused.h:
    // BlkDesc): the row words come back in one round trip instead of list -> row words -> members' words.
    // MODE 1: full call (structure = the column bitmap k_sym_dense wrote); MODE 2: numeric-only re-run
    // (structure = C.col, the bitmap is rebuilt from it).
    // prof != nullptr (NSPARSE_BLK_PROF=1): thread 0 of every group head adds the shader-clock cycles of
    // its phases to its slots prof[8 * block + 0..4] (meta, park loads, run building, walk, emission;
    // [5], [6] = start and end of the group on the 100 MHz clock, [7] = rows)
    unsigned long long t_prev = (kExperiments && prof) ? __builtin_readcyclecounter() : 0;
    auto stamp = [&](int phase) {
        if (kExperiments && prof && threadIdx.x == 0) {
            const unsigned long long t = __builtin_readcyclecounter();
            prof[8ull * blockIdx.x + phase] += t - t_prev;
            t_prev = t;
        }
    };
    constexpr int NW = BS / 64;
    constexpr int PARK = BS < 96 ? BS : (BS <= 128 ? 96 : 256);  // A entries parked per batch (a 27-point node stencil has 81; LDS per group bounds the groups in flight)
    constexpr int NWORDS = SPAN_MAX / 32 + 2;
    // The accumulators are COMPACT: the value of column c lives at rank(c) = number of columns of the
    // row below c, read off the bitmap (prefix of the word + popcount inside it).  A row of a
    // finite-element matrix fills a third of its window (375 of 1215 columns on the cant class), and
    // the LDS per workgroup is what bounds the rows in flight per CU -- the kernel is bound by the
    // latency of its dependent loads, i.e. by how many rows are in flight.  Values leave in rank order,
    // which is the output order: the emission is a straight copy.
    acc_t *acc = reinterpret_cast<acc_t *>(nsp_dyn_lds);  // RA rows of nzs accumulators
    __shared__ unsigned int s_bits[NWORDS];
    __shared__ int s_pre[NWORDS];
    // A batch of parked entries is walked in passes of at most RUNCAP runs, a pass in stretches of at most
    // TCAP tasks (one pass and one stretch on the cant class: 27 runs, 81 tasks).
    // s_task: one record per (run, chunk of G entries of its rows of B) -- x, y, z = first entry of the
    // chunk in the (up to 3) rows of B, w = entries in the chunk | rows << 8 | run << 16.  Written ONCE by
    // the lane that leads the run, so that the lanes of a group do not each decode extents, chunk
    // counters and bounds again, and a run only has the chunks its own length needs.
    // s_a: the 3 x 3 block of A values of a run, [row of B in the run][row of the group] -- 9 consecutive
    // words written by the entries themselves once they know (run, place in the run), ZERO where the run has
    // fewer rows of B: the walk reads them back to back without conditions, and neither the values of a
    // neighbouring run nor an Inf / NaN of it can leak into this one (the reference yields NaN / Inf only
    // in the columns the offending entry touches).
    constexpr int RUNCAP = PARK * 5 / 12, TCAP = KEYED ? PARK * 4 / 3 : PARK;
    constexpr int NPASS = (PARK + RUNCAP - 1) / RUNCAP;
    __shared__ int4 s_task[TCAP];
    __shared__ real s_a[RUNCAP * kBlkRun * kBlkRows];
    __shared__ int s_wtask[NW], s_tstart[NPASS + 1];
    __shared__ int s_wcnt[NW];
    const int slot0 = xcd_row_slot(bin_size);
    if (slot0 < 0) return;
    const int slot = slot0 < seg1 ? bin_off + slot0 : bin_off2 + (slot0 - seg1);  // position in the row list
    int rid, RA, lo, span, maxb, bmo = 0, alen;
    int off[kBlkRows], a_beg[kBlkRows];
    if (desc) {
        const int4 d0 = desc[3 * slot], d1 = desc[3 * slot + 1], d2 = desc[3 * slot + 2];
        rid = d0.x, lo = d0.y, span = d0.z, maxb = d0.w;
        bmo = d1.x, a_beg[0] = d1.y, alen = d1.z, RA = d1.w;
        off[1] = RA > 1 ? crpt[d2.x] : 0;
        off[2] = RA > 2 ? crpt[d2.y] : 0;
        a_beg[1] = RA > 1 ? arpt[d2.x] : 0;
        a_beg[2] = RA > 2 ? arpt[d2.y] : 0;
    } else {
        rid = row_perm[slot];
        const int gcode = grp ? (int)grp[rid] : (1 << 2);
        if (gcode & 3) return;  // a follower: its group head computes this row
        RA = gcode >> 2;
        lo = row_lo[rid];
        span = row_span[rid];
        // the rows of the group: the head (a pattern leader) and the twins recorded as its members -- any
        // rows of the matrix, not necessarily neighbours
#pragma unroll
        for (int r = 0; r < kBlkRows; r++) {
            const int rr = (r == 0 || r >= RA) ? rid : members[kGroupMembers * rid + r - 1];
            off[r] = (r > 0 && r < RA) ? crpt[rr] : 0;
            a_beg[r] = r < RA ? arpt[rr] : 0;
        }
        alen = arpt[rid + 1] - a_beg[0];
        maxb = row_maxb[rid];
        if (MODE == 1) bmo = bm_off[rid];
    }
    if (kExperiments && prof && threadIdx.x == 0) {
        prof[8ull * blockIdx.x + 5] = wall_clock64();  // 100 MHz: when this group started
        prof[8ull * blockIdx.x + 7] = (unsigned long long)RA;
    }
    struct __attribute__((aligned(4))) I2 {
        int b, e;
    };
    // The first batch of A entries is requested HERE, as soon as the row's place in A is known, and the
    // extents of their rows of B right behind them: the row words, the bitmap and the accumulators are set up
    // while those two round trips are under way (they used to start after all of that: three round trips more
    // on the critical path of a group).
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    struct Parked {
        int c, kb, ke, kl;
        unsigned char twb;
        real av[kBlkRows];
    };
    auto park_a = [&](int a0, Parked &pk) {  // the entries themselves
        pk.c = -2;
#pragma unroll
        for (int r = 0; r < kBlkRows; r++) pk.av[r] = (real)0;
        const int j = a0 + (int)threadIdx.x;
        if (j < alen && (int)threadIdx.x < PARK) {
            pk.c = __builtin_nontemporal_load(acol + a_beg[0] + j);
#pragma unroll
            for (int r = 0; r < kBlkRows; r++)
                if (r < RA) pk.av[r] = __builtin_nontemporal_load(aval + a_beg[r] + j);
        }
    };
    auto park_b = [&](Parked &pk) {  // what hangs on their columns (values as loaded: nothing here waits)
        pk.kb = 0, pk.ke = 0, pk.kl = -1, pk.twb = 0;
        if (pk.c >= 0) {
            const I2 rr = *reinterpret_cast<const I2 *>(brpt + pk.c);
            pk.kb = rr.b, pk.ke = rr.e;
            if (btwin != nullptr) pk.twb = btwin[pk.c];
            if constexpr (KEYED) pk.kl = bkey[pk.c];
        }
    };
    Parked pk0;
    park_a(0, pk0);
    const int nw = (span + 31) >> 5;
    unsigned int bw0 = 0;  // the first bitmap word of this thread: requested between the two
    if (MODE == 1 && (int)threadIdx.x < nw) bw0 = bm[bmo + threadIdx.x];
    park_b(pk0);
    const I2 cr = *reinterpret_cast<const I2 *>(crpt + rid);
    off[0] = cr.b;
    const int nz = cr.e - cr.b;
    const int nzs = ((nz + 7) >> 3) << 3;  // the number k_twin_groups sized the group with
    if (MODE == 1) {
        const unsigned int *bits = bm + bmo;
        if ((int)threadIdx.x < nw) s_bits[threadIdx.x] = bw0;
        for (int i = threadIdx.x + BS; i < nw; i += BS) s_bits[i] = bits[i];
    } else {
        for (int i = threadIdx.x; i < nw; i += BS) s_bits[i] = 0;
        __syncthreads();
        for (int p = threadIdx.x; p < nz; p += BS) {
            const int idx = ccol[off[0] + p] - lo;
            atomicOr(&s_bits[idx >> 5], 1u << (idx & 31));
        }
    }
    for (int i = threadIdx.x; i < RA * nzs; i += BS) acc[i] = 0;
    // A lane takes K columns of a chunk (gl, gl + G, ...): one task record and one 3 x 3 block of A values read
    // from LDS serve K columns -- the LDS unit is the busiest part of a CU under this kernel (counters: 69 %
    // of the cycles with one column per lane, the block alone 36 of 72 cycles per task and wavefront).
    constexpr int K = kBlkCols;
    int G = 4;
    while (G < 64 / K && G * (3 * K) < maxb) G <<= 1;  // three chunks cover the longest row of B this C row meets
    const int lg = 31 - __clz(G);
    const int GW = G * K;  // entries per chunk
    const int gid = (int)threadIdx.x >> lg, gl = (int)threadIdx.x & (G - 1);
    const int NG = BS >> lg;
    if (kExperiments && prof) { __syncthreads(); stamp(0); }

    // One iteration = one stretch of one pass of one batch of parked entries.  Batches with several passes or
    // stretches (more than RUNCAP runs / TCAP tasks among PARK entries: rare) park their entries AGAIN for each
    // of them, so that nothing but three uniform counters lives across the walk (registers: 6 waves per SIMD).
    int a0 = 0, pass = 0, base = 0;
    bool first_iter = true;
    while (a0 < alen) {
        // ---- park up to PARK entries of A and cut them into runs ---------------------------------
        const int j = a0 + (int)threadIdx.x;
        const bool valid = j < alen && (int)threadIdx.x < PARK;
        Parked pk = pk0;
        if (!first_iter) {
            park_a(a0, pk);
            park_b(pk);
        }
        const int c = pk.c, kb = pk.kb, ke = pk.ke;
        const bool tw = pk.twb != 0;
        real av[kBlkRows];
#pragma unroll
        for (int r = 0; r < kBlkRows; r++) av[r] = pk.av[r];
        int d, nB, my_run, nruns = 0;
        bool leader;
        int lead_lane = lane;  // KEYED: the lane whose entry opened my run,
        int f1 = lane, f2 = lane;  // ... and (for that lane) the lanes of its run mates
        if constexpr (!KEYED) {
            const int cprev = __shfl_up(c, 1);
            // entry j continues the run of entry j - 1 when its row of B is the twin of that one's (same
            // columns, hence same length) and directly follows it; runs never cross a wavefront
            const bool head = !(lane > 0 && tw && c == cprev + 1);
            const unsigned long long hm = __ballot(head || !valid);
            const unsigned long long vm = __ballot(valid);
            const int nvalid = __popcll(vm);  // valid lanes are a prefix
            const unsigned long long below = hm & ((2ull << lane) - 1ull);
            const int h = 63 - __clzll((long long)below);  // lane 0 is always a head
            const int pos = lane - h;
            d = pos % kBlkRun;
            leader = valid && d == 0;
            const unsigned long long above = lane < 63 ? hm >> (lane + 1) : 0ull;
            int end = above ? lane + 1 + (__ffsll((long long)above) - 1) : 64;
            end = end < nvalid ? end : nvalid;
            nB = end - lane;
            nB = nB > kBlkRun ? kBlkRun : nB;
        } else {
            // keyed runs: the entries of this WAVEFRONT whose rows of B share a pattern leader, three at a
            // time in lane order.  One ballot per distinct key (a third of the entries on a 3-dof mesh)
            // instead of every entry comparing itself with every parked one; runs never cross a wavefront.
            const int key = valid ? (pk.kl >= 0 ? pk.kl : c) : -3;
            unsigned long long mine = 0ull;
            unsigned long long todo = __ballot(valid);
            while (todo) {
                const int src = __ffsll((long long)todo) - 1;
                const int k = __builtin_amdgcn_readlane(key, src);
                const unsigned long long m = __ballot(valid && key == k);
                if (valid && key == k) mine = m;
                todo &= ~m;
            }
            const unsigned long long lower = mine & ((1ull << lane) - 1ull);
            const int before = __popcll(lower);
            const int last1 = lower ? 63 - __clzll((long long)lower) : 0;
            const unsigned long long lower2 = lower & ~(1ull << last1);
            const int last2 = lower2 ? 63 - __clzll((long long)lower2) : 0;
            d = before % kBlkRun;
            leader = valid && d == 0;
            nB = __popcll(mine) - before;
            nB = nB > kBlkRun ? kBlkRun : nB;
            lead_lane = d == 1 ? last1 : (d == 2 ? last2 : lane);
            const unsigned long long upper = lane < 63 ? mine & ~((2ull << lane) - 1ull) : 0ull;
            const unsigned long long upper2 = upper & (upper - 1ull);
            f1 = upper ? __ffsll((long long)upper) - 1 : lane;
            f2 = upper2 ? __ffsll((long long)upper2) - 1 : lane;
        }
        const unsigned long long lm = __ballot(leader);
        // the leader of a run collects the extents of its run mates and counts the chunks of the run
        int kb1, kb2;
        if constexpr (!KEYED) {
            kb1 = __shfl_down(kb, 1);
            kb2 = __shfl_down(kb, 2);
        } else {
            kb1 = __shfl(kb, f1);
            kb2 = __shfl(kb, f2);
        }
        const int blen = ke - kb;
        const int nchunk = leader ? (blen + GW - 1) / GW : 0;
        const int tincl = wave_incl_scan(nchunk);
        if (lane == 63) s_wtask[wv] = tincl;
        if (lane == 0) s_wcnt[wv] = __popcll(lm);
        __syncthreads();  // also: bitmap in LDS, accumulators cleared; the previous batch's walk is over
        stamp(1);
        if (first_iter && wv == NW - 1) {
            // exclusive prefix of the word popcounts: the last wavefront (it parks the fewest entries)
            int carry = 0;
            for (int b0 = 0; b0 < nw; b0 += 64) {
                const int v = b0 + lane < nw ? __popc(s_bits[b0 + lane]) : 0;
                const int inc = wave_incl_scan(v);
                if (b0 + lane < nw) s_pre[b0 + lane] = carry + inc - v;
                carry += __shfl(inc, 63);
            }
        }
        int wbase = 0, tbase = 0, ttotal = 0;
#pragma unroll
        for (int u = 0; u < NW; u++) {
            wbase += u < wv ? s_wcnt[u] : 0;
            nruns += s_wcnt[u];
            tbase += u < wv ? s_wtask[u] : 0;
            ttotal += s_wtask[u];
        }
        my_run = wbase + __popcll(lm & ((2ull << lane) - 1ull)) - 1;  // of the last leader at or below this lane
        if constexpr (KEYED) my_run = __shfl(my_run, lead_lane);       // of the entry that opened my run
        const int tpre = tbase + tincl - nchunk;  // leaders: the first task of my run among the batch's
        // where the tasks of every pass start (read after the barrier that follows the first staging)
        if (leader && my_run % RUNCAP == 0) s_tstart[my_run / RUNCAP] = tpre;
        if (threadIdx.x == 0) s_tstart[(nruns + RUNCAP - 1) / RUNCAP] = ttotal;

        // ---- stage the A values of the runs [pass RUNCAP, (pass + 1) RUNCAP) and the task records of the
        // stretch [base, base + TCAP): the tasks of the batch are numbered through -----------------------
        {
            const int rel = my_run - pass * RUNCAP;
            if (valid && rel >= 0 && rel < RUNCAP) {
#pragma unroll
                for (int r = 0; r < kBlkRows; r++) s_a[(rel * kBlkRun + d) * kBlkRows + r] = av[r];
                if (leader) {
                    for (int q = nB; q < kBlkRun; q++)
#pragma unroll
                        for (int r = 0; r < kBlkRows; r++) s_a[(rel * kBlkRun + q) * kBlkRows + r] = (real)0;
                    const int meta = (nB << 8) | (rel << 16);
                    int q = base - tpre > 0 ? base - tpre : 0;
                    const int q1 = base + TCAP - tpre < nchunk ? base + TCAP - tpre : nchunk;
                    for (; q < q1; q++) {
                        const int st = q * GW;
                        const int cnt = blen - st < GW ? blen - st : GW;
                        s_task[tpre + q - base] =
                            make_int4(kb + st, (nB > 1 ? kb1 : kb) + st, (nB > 2 ? kb2 : kb) + st, cnt | meta);
                    }
                }
            }
        }
        __syncthreads();
        stamp(2);
        const int tA = s_tstart[pass], tB = s_tstart[pass + 1];  // the tasks of this pass
        {
            // ---- walk: group q takes a contiguous stretch of the task list, U tasks in flight (their
            // loads issued before the first is added) ---------------------------------------------
            const int lo_t = (tA > base ? tA : base) - base;
            const int hi_t = (tB < base + TCAP ? tB : base + TCAP) - base;
            const int ntask = hi_t - lo_t;
            const int per = (ntask + NG - 1) / NG;
            const int t0 = lo_t + gid * per;
            const int t1 = t0 + per < hi_t ? t0 + per : hi_t;
            for (int tb = t0; tb < t1; tb += U) {
                int col[U][K], ru[U];
                real v[kBlkRun][U][K];
                bool ok[U][K];
#pragma unroll
                for (int i = 0; i < U; i++) {
                    const bool live = tb + i < t1;
                    const int4 e = s_task[live ? tb + i : t0];
                    const int nb = (e.w >> 8) & 3;
                    const int cnt = live ? e.w & 0xff : 0;
                    ru[i] = e.w >> 16;
#pragma unroll
                    for (int k = 0; k < K; k++) {
                        const int pk = gl + k * G;
                        ok[i][k] = pk < cnt;
                        const unsigned idx = ok[i][k] ? (unsigned)pk : 0u;  // masked lanes re-read the first entry of the chunk
                        col[i][k] = bcol[(unsigned)e.x + idx];
                        v[0][i][k] = bval[(unsigned)e.x + idx];
                        v[1][i][k] = nb > 1 ? bval[(unsigned)e.y + idx] : (real)0;
                        v[2][i][k] = nb > 2 ? bval[(unsigned)e.z + idx] : (real)0;
                    }
                }
#pragma unroll
                for (int i = 0; i < U; i++) {
                    if (ok[i][0]) {  // (the columns of a lane are filled in order)
                        int rank[K];
#pragma unroll
                        for (int k = 0; k < K; k++) {
                            const int idx = ok[i][k] ? col[i][k] - lo : 0;
                            rank[k] = s_pre[idx >> 5] + __popc(s_bits[idx >> 5] & ((1u << (idx & 31)) - 1u));
                        }
                        const real *ab = s_a + ru[i] * (kBlkRun * kBlkRows);
#pragma unroll
                        for (int r = 0; r < kBlkRows; r++) {
                            if (r < RA) {
                                const real a0 = ab[r], a1 = ab[kBlkRows + r], a2 = ab[2 * kBlkRows + r];
#pragma unroll
                                for (int k = 0; k < K; k++) {
                                    if (ok[i][k]) {
                                        acc_t sum = (acc_t)(a0 * v[0][i][k]);
                                        sum += (acc_t)(a1 * v[1][i][k]);  // both factors are zero for the rows the run lacks
                                        sum += (acc_t)(a2 * v[2][i][k]);
                                        unsafeAtomicAdd(acc + r * nzs + rank[k], sum);
                                    }
                                }
                            }
                        }
                    }
                }
            }
        }
        stamp(3);
        if (base + TCAP < tB) {
            base += TCAP;  // the next stretch of this pass
        } else if ((pass + 1) * RUNCAP < nruns) {
            pass++;  // the next pass starts where this one ended
            base = tB / TCAP * TCAP;
        } else {
            a0 += PARK;
            pass = 0;
            base = 0;
        }
        first_iter = false;
    }
    __syncthreads();

    // ---- emission: values are in output order already; columns are read off the bitmap ------------
    for (int r = 0; r < RA; r++)
        for (int k = threadIdx.x; k < nz; k += BS) cval[off[r] + k] = (real)acc[r * nzs + k];
    if (MODE == 1) {
        for (int idx = threadIdx.x; idx < span; idx += BS) {
            const unsigned int wbits = s_bits[idx >> 5];
            if ((wbits >> (idx & 31)) & 1u) {
                const int rank = s_pre[idx >> 5] + __popc(wbits & ((1u << (idx & 31)) - 1u));
                for (int r = 0; r < RA; r++) ccol[off[r] + rank] = lo + idx;
            }
        }
    }
    if (kExperiments && prof) {
        __syncthreads();
        stamp(4);
        if (threadIdx.x == 0) prof[8ull * blockIdx.x + 6] = wall_clock64();  // ... and when it was done
    }
}


}  // namespace spgemm
}  // namespace nsp
