// spgemm/fused.h -- the tails of the set-up and of the symbolic phase as ONE launch each, for matrices of
// up to 1 M rows (one row per thread up to 256 K rows, four beyond).
// Part of the spgemm_hash.hip translation unit.
//
// Between the big kernels of a call sit chains of 1024-rows-per-workgroup helpers that each run for a
// microsecond and cost four to five (launch, drain, the dependency on the one before):
//     set-up:    k_reduce_partials -> scan (2 launches) -> k_bin_scatter -> k_publish          24 us
//     symbolic:  k_twin_copy -> k_twin_groups -> scan (2) -> k_hist -> k_bin_scatter -> k_publish 32 us
// of a 360 us product on the cant class.  Here every chain is one kernel of ceil((M + 1) / 1024)
// workgroups (at most 256: one per CU, all resident) with one grid barrier in the middle -- what needs
// every row's contribution (histogram, scan carries) is on one side, what consumes it on the other -- and
// a wavefront of workgroup 0 publishes the counters to the host right behind the barrier.  Same results as
// the chains they replace (which stay for larger M and for numeric-only calls); NSPARSE_FUSED=0 forces
// the chains.
//
// Cross-workgroup data inside a launch is read with agent-scope atomic loads (never through const
// __restrict__ pointers: the scalar cache is not part of the memory model), produced before a release
// fence + atomic arrive, consumed after the acquire side of the barrier.
#pragma once
#include "block.h"
#include "common.h"
#include "setup.h"

namespace nsp {
namespace spgemm {

constexpr int kFusedMaxBlocks = 256;

struct FusedSync {
    int *arrive;    // [0] grid barrier counter, [1] "a workgroup gave up" (both zeroed by the per-call fill)
    int *blk;       // per workgroup: [0] scan carry, [1 + q] rows it lists in bin q   (kFusedRec ints)
    int *pub_dst;   // mapped host copy of the counter block
    int *pub_flag;  // sequence flag the host polls
    int seq;
    int *fail_flag; // mapped host word: = seq when the barrier timed out (the host then repeats the call unfused)
};
constexpr int kFusedRec = NB + 1;

__device__ __forceinline__ int ld_agent(const int *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Every workgroup of the grid has arrived.  The launch is an ordinary one, so that all workgroups are resident
// at once is something the HOST has made sure of (Context::coresident: a census at context creation with one
// workgroup per CU, under whatever CU mask or partition the process runs in) -- but CUs can still be taken by
// another tenant between the census and the call.  So the wait is bounded (50 ms of the 100 MHz wall clock) and its
// OUTCOME IS ONE AGREED WORD, arrive[1]: 0 open, 1 passed, 2 failed.  Only the last workgroup to arrive can move
// it from open to passed, only a workgroup whose wait has run out from open to failed, both with a
// compare-and-swap, and every workgroup -- also one that only starts later -- acts on what it then READS there.
// (Round 3 let every workgroup decide for itself: a last arrival inside the window between another workgroup's
// final look at the counter and its "gave up" store could run the tail while that workgroup skipped its rows.)
// On failure nothing behind the barrier is touched and the host repeats the call with the kernel chains (round 2
// trapped here, which poisons the process's HIP context).  Returns false when the barrier failed.
__device__ __forceinline__ bool grid_barrier(const FusedSync &fs, int *s_ok)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        int *state = fs.arrive + 1;
        const int target = (int)gridDim.x;
        const int mine = __hip_atomic_fetch_add(fs.arrive, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1;
        int st;
        if (mine == target) {
            int open = 0;
            __hip_atomic_compare_exchange_strong(state, &open, 1, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            st = __hip_atomic_load(state, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const unsigned long long t0 = wall_clock64();
            while ((st = __hip_atomic_load(state, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) == 0) {
                __builtin_amdgcn_s_sleep(2);
                if (wall_clock64() - t0 > 5000000ull) {  // 50 ms
                    int open = 0;
                    __hip_atomic_compare_exchange_strong(state, &open, 2, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                    st = __hip_atomic_load(state, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        if (st != 1) {  // (every failing workgroup says so: the stores are idempotent)
            __hip_atomic_store(fs.fail_flag, fs.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __threadfence_system();
            __hip_atomic_store(fs.pub_flag, fs.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        *s_ok = st == 1 ? 1 : 0;
    }
    __syncthreads();
    return *s_ok != 0;
}

// Census for the grid barrier: `grid` workgroups of 1024 threads; every one waits (bounded: ~0.2 ms) until all
// have arrived and reports whether it saw that happen.  out[0] counts arrivals, out[1] the workgroups that
// timed out: 0 means `grid` such workgroups are resident together on this device as this process sees it.
// Launched with kCensusLds bytes of dynamic LDS -- more than half a CU's -- so that the census counts CUs (one
// workgroup each) and not how many of ITS featherweight workgroups fit a CU: the fused kernels are register-heavy
// 1024-thread workgroups of which a CU may hold just one.
constexpr int kCensusLds = 96 * 1024;
__global__ __launch_bounds__(1024) void k_census(int *out, int limit_ticks)
{
#ifndef NSP_EMU
    extern __shared__ int s_census_pad[];
#else
    int *s_census_pad = reinterpret_cast<int *>(::emu::t_dyn_lds);  // tests/emu
#endif
    if (threadIdx.x == 1023 && limit_ticks < 0) s_census_pad[0] = 0;  // (keeps the allocation alive)
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(out, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = wall_clock64();
        bool ok = true;
        while (__hip_atomic_load(out, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (int)gridDim.x) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > (unsigned long long)limit_ticks) {
                ok = false;
                break;
            }
        }
        if (!ok) __hip_atomic_fetch_add(out + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
}

// exclusive scan of one int per thread over a 1024-thread workgroup; total = sum of all
__device__ __forceinline__ int block_scan_1024(int v, int *s_w /* 16 */, int &total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const int w = s_w[q];
        all += w;
        before += q < wv ? w : 0;
    }
    total = all;
    __syncthreads();
    return before + inc - v;
}

// rank of the thread's row inside (workgroup, bin): ballot + popcount inside the wave, one LDS atomic per
// (wave, bin present in the wave)
__device__ __forceinline__ int rank_in_bin(int bin, int *s_cnt)
{
    const int lane = threadIdx.x & 63;
    int r = 0;
    unsigned long long todo = __ballot(bin >= 0);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int bb = __shfl(bin, leader);
        const unsigned long long same = __ballot(bin == bb);
        int base = 0;
        if (lane == leader) base = atomicAdd(&s_cnt[bb], __popcll(same));
        base = __shfl(base, leader);
        if (bin == bb) r = base + __popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    return r;
}

// Second half, shared by both kernels.  Everything the host wants (histogram, listed rows per bin, maxima,
// totals) was complete at the barrier, so one wavefront of workgroup 0 publishes at once while the others
// place their rows: a workgroup's rows follow those of the workgroups before it in every bin (sums of the
// records written before the barrier -- no atomics, and the lists come out in ascending row order).
// Leaves s_pref[0] = scan carry of the workgroups before this one and s_base[q] = where this workgroup's rows of
// bin q start in the permutation; the caller then places its rows (fused_place).
__device__ __forceinline__ void fused_tail(int *s_base, int *s_pref /* kFusedRec */, int *s_h /* NB */, BinState *bs,
                                           const FusedSync &fs, bool set_nnz)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    static_assert(kFusedRec + 2 <= 16, "one wavefront per quantity");
    if (wv < kFusedRec) {
        int acc = 0;
        for (int t = lane; t < (int)blockIdx.x; t += 64) acc += ld_agent(fs.blk + t * kFusedRec + wv);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) s_pref[wv] = acc;
    } else if (wv == kFusedRec) {
        if (lane < NB) s_h[lane] = ld_agent(&bs->hist[lane]);
    } else if (wv == kFusedRec + 1 && blockIdx.x == 0) {
        constexpr int words = (int)(sizeof(BinState) / 4);
        static_assert(words <= 64, "one word per lane");
        const int *s = reinterpret_cast<const int *>(bs);
        if (lane < words) fs.pub_dst[lane] = ld_agent(s + lane);
        // program order of this wavefront, said aloud (scheduling barriers: no instruction; the sync points of
        // tests/emu): lane 0's word lands after the copy of the same word, the flag after every lane's word
        __builtin_amdgcn_wave_barrier();
        if (set_nnz && lane == 0)
            reinterpret_cast<BinState *>(fs.pub_dst)->nnz = ld_agent(reinterpret_cast<const int *>(&bs->total));
        __threadfence_system();
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) __hip_atomic_store(fs.pub_flag, fs.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    if (threadIdx.x < NB) {
        int off = 0;
        for (int q = 0; q < (int)threadIdx.x; q++) off += s_h[q];
        s_base[threadIdx.x] = off + s_pref[1 + threadIdx.x];
    }
    __syncthreads();
}

// row i of this thread: its scan value and its place in the permutation; returns the list position (-1: not listed)
__device__ __forceinline__ int fused_place(int i, int M, int excl, int *out_scan, int bin, bool listed, int rank,
                                            const int *s_base, const int *s_pref, int *perm)
{
    if (out_scan && i <= M) out_scan[i] = s_pref[0] + excl;
    if (!listed) return -1;
    perm[s_base[bin] + rank] = i;
    return s_base[bin] + rank;
}

// ---- set-up tail: fold the per-workgroup partials of k_row_products, offsets of the column bitmaps,
//      symbolic row permutation, publish ---------------------------------------------------------------
// R rows per thread: workgroup b owns rows [b * R * 1024, (b + 1) * R * 1024) in R slices of 1024 (R = 1 up to 256 K
// rows, R = 4 up to 1 M: the grid stays at one workgroup per CU at most; the row records -- desc -- only with R = 1)
template <int R>
__global__ __launch_bounds__(1024) void k_setup_tail(const long long *__restrict__ partial, int nparts,
                                                     BinState *bs, const int *__restrict__ bm_words,
                                                     int *__restrict__ bm_off,
                                                     const int *__restrict__ row_prod,
                                                     const int *__restrict__ row_span, int M, Thr thr,
                                                     int *__restrict__ perm,
                                                     const unsigned char *__restrict__ skip, FusedSync fs,
                                                     int4 *__restrict__ desc, const int *__restrict__ arpt,
                                                     const int *__restrict__ row_lo,
                                                     const int *__restrict__ row_maxb)
{
    // desc != nullptr: a record per row listed in a window bin, in list order, for k_sym_dense (window.h):
    // {row, lo, span, longest B row | first A entry, end of the A row, products, bitmap offset | bitmap words}
    __shared__ unsigned long long s_acc[kPartialStride];
    __shared__ int s_max, s_alen, s_ok;
    __shared__ int s_cnt[NB], s_base[NB], s_span[NB], s_w[16], s_pref[kFusedRec], s_h[NB];
    if (threadIdx.x < kPartialStride) s_acc[threadIdx.x] = 0;
    if (threadIdx.x < NB) s_cnt[threadIdx.x] = s_span[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_max = s_alen = 0;
    __syncthreads();
    // (1) this workgroup's slice of the partials (k_reduce_partials)
    {
        const int per = (nparts + (int)gridDim.x - 1) / (int)gridDim.x;
        const int b0 = blockIdx.x * per, b1 = b0 + per < nparts ? b0 + per : nparts;
        const int f = threadIdx.x & 31;
        if (f < NB + 5 && b0 < b1) {
            long long acc = 0;
            for (int b = b0 + ((int)threadIdx.x >> 5); b < b1; b += 32) {
                const long long v = partial[(long long)b * kPartialStride + f];
                acc = (f == NB || f == NB + 3) ? (v > acc ? v : acc) : acc + v;
            }
            if (f == NB) atomicMax(&s_max, (int)acc);
            else if (f == NB + 3) atomicMax(&s_alen, (int)acc);
            else if (acc) atomicAdd(&s_acc[f], (unsigned long long)acc);
        }
    }
    // (2) bitmap words of this workgroup's rows, scanned; (3) bin and rank of every listed row
    int v[R], excl[R], bin[R], rank[R];
    bool listed[R];
    int4 d0 = make_int4(0, 0, 0, 0), d1 = d0;
    int carry = 0;
#pragma unroll
    for (int k = 0; k < R; k++) {
        const int i = ((int)blockIdx.x * R + k) * 1024 + (int)threadIdx.x;
        v[k] = (bm_words && i <= M) ? bm_words[i] : 0;
        int total = 0;
        excl[k] = carry + block_scan_1024(v[k], s_w, total);
        carry += total;
        listed[k] = i < M && !(skip && skip[i]);
        bin[k] = -1;
        if (listed[k]) {
            const int ni = row_prod[i], sp = row_span[i];
            bin[k] = bin_of(ni, sp, thr, ni);
            if (bin[k] >= kDenseBin0) {
                atomicMax(&s_span[bin[k]], sp);
                if (R == 1 && desc) {
                    d0 = make_int4(i, row_lo[i], sp, row_maxb[i]);
                    d1 = make_int4(arpt[i], arpt[i + 1], ni, 0);
                }
            }
        }
        rank[k] = rank_in_bin(bin[k], s_cnt);  // (slices in order: a later slice ranks behind the earlier ones)
    }
    __syncthreads();
    if (threadIdx.x < NB) {
        if (s_acc[threadIdx.x]) atomicAdd(&bs->hist[threadIdx.x], (int)s_acc[threadIdx.x]);
        if (s_cnt[threadIdx.x]) atomicAdd(&bs->cursor[threadIdx.x], s_cnt[threadIdx.x]);
        if (s_span[threadIdx.x]) atomicMax(&bs->max_span[threadIdx.x], s_span[threadIdx.x]);
        __hip_atomic_store(fs.blk + blockIdx.x * kFusedRec + 1 + threadIdx.x, s_cnt[threadIdx.x], __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0) {
        if (s_max) atomicMax(&bs->maxv, s_max);
        if (s_acc[NB + 1]) atomicAdd((unsigned long long *)&bs->total, s_acc[NB + 1]);
        if (s_acc[NB + 2]) atomicAdd((unsigned long long *)&bs->bm_total, s_acc[NB + 2]);
        if (s_acc[NB + 4]) atomicAdd((unsigned long long *)&bs->list_total, s_acc[NB + 4]);
        if (s_alen) atomicMax((unsigned long long *)&bs->max_alen, (unsigned long long)s_alen);
        __hip_atomic_store(fs.blk + blockIdx.x * kFusedRec, carry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!grid_barrier(fs, &s_ok)) return;
    fused_tail(s_base, s_pref, s_h, bs, fs, false);
#pragma unroll
    for (int k = 0; k < R; k++) {
        const int i = ((int)blockIdx.x * R + k) * 1024 + (int)threadIdx.x;
        const int pos = fused_place(i, M, excl[k], bm_words ? bm_off : nullptr, bin[k], listed[k], rank[k], s_base, s_pref, perm);
        if (R == 1 && desc && pos >= 0 && bin[k] >= kDenseBin0) {
            d1.w = s_pref[0] + excl[k];  // this row's bitmap offset (what the scan wrote to bm_off[i])
            desc[3 * pos] = d0;
            desc[3 * pos + 1] = d1;
            desc[3 * pos + 2] = make_int4(v[k], 0, 0, 0);
        }
    }
}

// ---- symbolic tail: twins take their leader's result, node-block groups, C.rpt, numeric histogram and
//      row permutation, publish (k_twin_copy, k_twin_groups, scan, k_hist, k_bin_scatter, k_publish) -----
template <int R>
__global__ __launch_bounds__(1024) void k_numeric_setup(const int *__restrict__ twin_of,
                                                        const int *__restrict__ members, int *row_nz,
                                                        int *row_span_num, int *bm_off,
                                                        const int *__restrict__ row_prod, int M, Thr thr,
                                                        BinState *bs, int *__restrict__ crpt,
                                                        int *__restrict__ perm, unsigned char *__restrict__ grp,
                                                        FusedSync fs, int4 *__restrict__ desc,
                                                        const int *__restrict__ arpt,
                                                        const int *__restrict__ row_lo,
                                                        const int *__restrict__ row_maxb)
{
    // desc != nullptr (R = 1 only): a record per row listed in a window bin, in list order, for the node-block kernel
    // (block.h): {row, lo, span, longest B row | bitmap offset, first A entry, A entries, rows in the group |
    // member rows}
    __shared__ int s_hist[NB], s_cnt[NB], s_base[NB], s_span[NB], s_w[16], s_pref[kFusedRec], s_h[NB];
    __shared__ int s_max, s_far, s_ok;
    __shared__ unsigned long long s_sum;
    if (threadIdx.x < NB) s_hist[threadIdx.x] = s_cnt[threadIdx.x] = s_span[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        s_max = 0;
        s_far = 0;
        s_sum = 0;
    }
    __syncthreads();
    int excl[R], bin[R], rank[R];
    bool listed[R];
    int4 d0 = make_int4(0, 0, 0, 0), d1 = d0, d2 = d0;
    int carry = 0;
#pragma unroll
    for (int k = 0; k < R; k++) {
    const int i = ((int)blockIdx.x * R + k) * 1024 + (int)threadIdx.x;
    int nz = 0, hbin = -1;
    bin[k] = -1;
    listed[k] = false;
    if (i < M) {
        const int l = twin_of ? twin_of[i] : -1;
        const int lead = l >= 0 ? l : i;
        // a leader is no twin: nobody writes the entries read here
        nz = row_nz[lead];
        const int sp = row_span_num[lead];
        if (l >= 0) {
            row_nz[i] = nz;
            row_span_num[i] = sp;
            if (bm_off) bm_off[i] = bm_off[lead];
            if (l - i > 2 || i - l > 2) atomicAdd(&s_far, 1);  // (a matrix with neighbouring twins has none)
        }
        const int work = row_prod[i];
        hbin = bin_of(nz, sp, thr, work);
        int code = 1 << 2;
        int m0 = -1, m1 = -1;
        if (grp) {  // k_twin_groups
            const int nzs = ((nz + 7) >> 3) << 3;
            const bool windowed = sp > 0 && nzs > 0 && hbin >= kDenseBin0;
            int cap = windowed ? kBlkAccElems / nzs : 1;
            cap = cap < 1 ? 1 : (cap > kBlkRows ? kBlkRows : cap);
            m0 = members[kGroupMembers * lead], m1 = members[kGroupMembers * lead + 1];
            const int nf = m0 < 0 ? 0 : (m1 != m0 ? 2 : 1);
            const int gsize = 1 + nf < cap ? 1 + nf : cap;
            if (l < 0) {
                code = gsize << 2;
            } else {
                const int pos = i == m0 ? 1 : (i == m1 ? 2 : 0);
                if (pos > 0 && pos < gsize) code = pos;
            }
            grp[i] = (unsigned char)code;
        }
        listed[k] = (code & 3) == 0;
        if (listed[k]) {
            bin[k] = hbin;
            if (hbin >= kDenseBin0) {
                atomicMax(&s_span[hbin], sp);
                if (R == 1 && desc) {
                    const int ra = code >> 2, ab = arpt[i];
                    d0 = make_int4(i, row_lo[i], sp, row_maxb[i]);
                    d1 = make_int4(bm_off ? bm_off[i] : 0, ab, arpt[i + 1] - ab, ra);
                    d2 = make_int4(m0, m1, 0, 0);  // (their first A entries: read by the kernel with its C.rpt words)
                }
            }
        }
    }
    // histogram of every row (k_hist), longest row, 64-bit total
    {
        unsigned long long todo = __ballot(hbin >= 0);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int b = __shfl(hbin, leader);
            const unsigned long long same = __ballot(hbin == b);
            if ((threadIdx.x & 63) == leader) atomicAdd(&s_hist[b], __popcll(same));
            todo &= ~same;
        }
        int v = nz;
        unsigned long long sum = (unsigned long long)nz;
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
    }
    int total = 0;
    excl[k] = carry + block_scan_1024(nz, s_w, total);
    carry += total;
    rank[k] = rank_in_bin(bin[k], s_cnt);
    }  // slices
    __syncthreads();
    if (threadIdx.x < NB) {
        if (s_hist[threadIdx.x]) atomicAdd(&bs->hist[threadIdx.x], s_hist[threadIdx.x]);
        if (s_cnt[threadIdx.x]) atomicAdd(&bs->cursor[threadIdx.x], s_cnt[threadIdx.x]);
        if (s_span[threadIdx.x]) atomicMax(&bs->max_span[threadIdx.x], s_span[threadIdx.x]);
        __hip_atomic_store(fs.blk + blockIdx.x * kFusedRec + 1 + threadIdx.x, s_cnt[threadIdx.x], __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0) {
        if (s_max) atomicMax(&bs->maxv, s_max);
        if (s_sum) atomicAdd((unsigned long long *)&bs->total, s_sum);
        if (s_far) atomicAdd(&bs->far_twins, s_far);
        __hip_atomic_store(fs.blk + blockIdx.x * kFusedRec, carry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!grid_barrier(fs, &s_ok)) return;
    fused_tail(s_base, s_pref, s_h, bs, fs, true);
#pragma unroll
    for (int k = 0; k < R; k++) {
        const int i = ((int)blockIdx.x * R + k) * 1024 + (int)threadIdx.x;
        const int pos = fused_place(i, M, excl[k], crpt, bin[k], listed[k], rank[k], s_base, s_pref, perm);
        if (R == 1 && desc && pos >= 0 && bin[k] >= kDenseBin0) {
            desc[3 * pos] = d0;
            desc[3 * pos + 1] = d1;
            desc[3 * pos + 2] = d2;
        }
    }
}

}  // namespace spgemm
}  // namespace nsp
