// spgemm/heavy_flat.h -- heavy numeric rows, STATELESS tiles over a panel table of B (round 6).
// Part of the spgemm_hash.hip translation unit (kernels are launched from its host code).
//
// The cursor kernels (heavy_tiled.h, heavy_ranked.h) keep, per entry of the A row, a position in its row of B that
// moves from tile to tile: tile t + 1 cannot start before tile t has written its cursors back, every lane-serial
// cursor is a chain of dependent loads (state -> column -> compare -> refill), look-ahead that is not consumed is
// fetched again (tests/emu census, profiles/r06_emu_fetch_counts.txt: 1.17-1.43 x the minimum bytes per product in
// the dense tiles, 1.35-2.4 x in the ranked ones), and a hub row of A (more than 4096 entries) walks its state
// arrays in global memory one entry after the other.
//
// Here the position is not carried, it is LOOKED UP: once per call a panel table of B is built,
//     tab[s][p] = index of the first entry of the B row in slot s whose column is >= p * G      (p = 0 .. np),
// so that the part of B row c inside column panel p is  [tab[slot(c)][p], tab[slot(c)][p + 1])  -- one 8-byte gather,
// the same shape as the (B.rpt[c], B.rpt[c + 1]) gather of every other kernel.  A tile of a C row is then an ordinary
// FLAT product walk (common.h: walk_products_flat) whose extents come from the table: three dependent round trips per
// batch of 1024 entries of A whatever the lengths, vector loads of consecutive entries, no state to write back,
// nothing that ties tile t + 1 to tile t, and every entry of B inside the table is fetched exactly once per C row.
// Rows of B of at most `min_len` entries may be left out of the table (slot -1: matrices with millions of short
// rows, where a pointer row per B row would dwarf B): they are walked whole by every tile and filtered by column --
// 4 % of the products of the heavy rows of R-MAT-22.
//
// Replaces the tile loop of calculate_value_col_bin_each_gl (cuda-c/src/kernel/kernel_spgemm_hash_d.cu:929-1027: one
// global-memory hash table per heavy row, then a sort) like k_num_tiled does; output identical to it.
#pragma once
#include "common.h"

namespace nsp {
namespace spgemm {

// slot_of[r] = table slot of B row r, -1 for rows outside the table; slot_row[s] = r; *count = slots in use.
// ALL: every row has a slot (its own number).  Otherwise the rows longer than min_len, numbered by an exclusive scan of
// their flags (k_panel_flags -> rocprim::exclusive_scan -> k_panel_slots<false>): ascending, deterministic, and no
// atomics -- one returning atomic per long row on ONE counter would be 10^5 same-address device-scope atomics, which this
// part serialises at ~20 ns each (HISTORY.md 4.1: why no kernel here counts that way).
// (templates, like every kernel of this header: nothing of it is instantiated in a build that does not launch it)
template <int DUMMY>
__global__ __launch_bounds__(256) void k_panel_flags(const int *__restrict__ brpt, int m, int min_len, int *__restrict__ flag)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r <= m) flag[r] = r < m && brpt[r + 1] - brpt[r] > min_len ? 1 : 0;  // (flag[m] = 0: pos[m] is the count)
}

template <bool ALL>
__global__ __launch_bounds__(256) void k_panel_slots(const int *__restrict__ brpt, int m, int min_len,
                                                     int *__restrict__ slot_of, int *__restrict__ slot_row,
                                                     int *__restrict__ count, int slots_max, const int *__restrict__ pos)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r == 0) *count = ALL ? m : pos[m];
    if (r >= m) return;
    int s = -1;
    if (ALL) {
        s = r;
        slot_row[r] = r;
    } else if (brpt[r + 1] - brpt[r] > min_len) {
        s = pos[r];
        if (s < slots_max) slot_row[s] = r;
        else s = -1;  // (cannot happen: slots_max is an upper bound of the rows longer than min_len)
    }
    slot_of[r] = s;
}

// One thread per (slot, panel boundary): lower bound of p * G in the sorted row.
template <int G>
__global__ __launch_bounds__(256) void k_panel_fill(const int *__restrict__ brpt, const int *__restrict__ bcol,
                                                    const int *__restrict__ slot_row, const int *__restrict__ count,
                                                    int np, int *__restrict__ tab)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int stride = np + 1;
    const long long s = i / stride;
    const int p = (int)(i - s * stride);
    if (s >= *count) return;
    const int r = slot_row[s];
    int lo = brpt[r], hi = brpt[r + 1];
    const long long target = (long long)p * G;
    while (lo < hi) {
        const int mid = lo + ((hi - lo) >> 1);
        if ((long long)bcol[mid] < target) lo = mid + 1;
        else hi = mid;
    }
    tab[i] = lo;
}

// ---- where the extents of an A entry are found --------------------------------------------------------------------------
// The entries of a heavy row of A do not change from tile to tile, only the panel does.  `where` >= 0: the entry's B row
// has a table row starting at tab + where (slot * stride: the table is at most 2^28 ints); `where` < 0: a short row
// outside the table, ~where is the B row and its extent is the whole row (filtered by column later); kNoEntry: this
// thread holds no entry of the batch.  The first kFlatEpt batches of BS entries keep (where, A value) in registers across
// the tile loop, so that a tile starts with ONE gather -- the extents -- instead of the chain A.col -> slot -> extents;
// further batches (hub rows of A: more than kFlatEpt * BS entries) keep theirs in the workgroup's slice of the heavy
// bin's slab (the one the cursor kernels use for their state), written once per row and read back coalesced by the
// thread that wrote it -- without it every tile repeats a random 4-byte gather of slot_of per entry.  (k_sym_flat has no
// slab and walks the chain for those batches.)
constexpr int kNoEntry = (int)0x80000000;
constexpr int kFlatEpt = 2;
__device__ __forceinline__ int flat_where(const int *__restrict__ acol, const int *__restrict__ slot_of, int tstride, int j)
{
    const int c = acol[j];
    const int s = slot_of[c];
    return s >= 0 ? s * tstride : ~c;
}
__device__ __forceinline__ int2 flat_extent(const int *__restrict__ brpt, const int *__restrict__ tab, int where, int p_a, int p_b)
{
    if (where == kNoEntry) return make_int2(0, 0);
    if (where >= 0) return make_int2(tab[where + p_a], tab[where + p_b]);
    const int c = ~where;
    return make_int2(brpt[c], brpt[c + 1]);
}

// ===================================================================================
//  heavy numeric rows: dense column tiles = panels of the table, one flat walk each
// ===================================================================================
// Same rows, same queue, same LDS window and the same ordered emission as k_num_tiled<BS, W>; G = W.
template <int BS, int W>
__global__ __launch_bounds__(BS) void k_num_flat(const int *__restrict__ arpt, const int *__restrict__ acol,
                                                 const real *__restrict__ aval,
                                                 const int *__restrict__ brpt, const int *__restrict__ bcol,
                                                 const real *__restrict__ bval, int bnnz,
                                                 const int *__restrict__ slot_of, const int *__restrict__ tab, int tstride,
                                                 const int *__restrict__ crpt, int *__restrict__ ccol,
                                                 real *__restrict__ cval,
                                                 const int *__restrict__ row_perm, int bin_off, int count,
                                                 BinState *bs, const int *__restrict__ row_lo,
                                                 const int *__restrict__ row_span, int write_col, int dens,
                                                 const long long *__restrict__ list_off, long long list_work,
                                                 const int *__restrict__ row_prod,
                                                 int *__restrict__ slab, long long stride_ints, int amax)
{
    constexpr int NW = BS / 64;
    constexpr int V = VW, U = 4;
    constexpr int R = W / NW;  // columns of a tile emitted by one wavefront
    constexpr int IT = R / 64;
    static_assert(W % (NW * 64) == 0, "tile width must split evenly over the wavefronts");
    __shared__ __attribute__((aligned(16))) acc_t dense[W];
    __shared__ __attribute__((aligned(16))) unsigned int flag4[W / 4];
    __shared__ int2 s_ext[BS];
    __shared__ real s_av[BS];
    __shared__ FlatScratch<BS> fs;
    __shared__ int s_row;
    __shared__ int s_wcnt[NW];
    unsigned char *flag = reinterpret_cast<unsigned char *>(flag4);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // The window is clean on entry to every tile: cleared here once, the emission resets what it finds occupied.
    for (int i = threadIdx.x; i < W; i += BS) dense[i] = 0;
    for (int i = threadIdx.x; i < W / 4; i += BS) flag4[i] = 0;
    while (true) {
        __syncthreads();
        if (threadIdx.x == 0) s_row = atomicAdd(&bs->queue_head, 1);
        __syncthreads();
        const int q = s_row;
        if (q >= count) break;
        const int rid = row_perm[bin_off + q];
        const int lo = row_lo[rid], span = row_span[rid];
        // the rows k_num_tiled leaves to k_num_ranked / k_num_listed are left to them here as well
        if (dens > 0 && ((long long)(crpt[rid + 1] - crpt[rid]) * dens < span || span > 32 * W)) continue;
        if (list_work > 0 && list_wanted(crpt[rid + 1] - crpt[rid], row_prod[rid], list_work) &&
            (list_off == nullptr || list_off[rid] >= 0))
            continue;
        const int a_beg = arpt[rid], a_end = arpt[rid + 1];
        const int p_lo = lo / W, p_hi = (int)(((long long)lo + span - 1) / W);
        int pos = crpt[rid];
        int e_where[kFlatEpt];
        real e_av[kFlatEpt];
#pragma unroll
        for (int u = 0; u < kFlatEpt; u++) {
            const int j = a_beg + u * BS + (int)threadIdx.x;
            e_where[u] = kNoEntry;
            e_av[u] = 0;
            if (j < a_end) {
                e_where[u] = flat_where(acol, slot_of, tstride, j);
                e_av[u] = aval[j];
            }
        }
        // hub rows: the entries beyond the register batches park theirs in the slab (entry e <-> thread e % BS, so every
        // thread reads back what it wrote itself)
        int *st_where = slab + (long long)blockIdx.x * stride_ints;
        real *st_av = reinterpret_cast<real *>(st_where + amax);
        for (int j = a_beg + kFlatEpt * BS + (int)threadIdx.x; j < a_end; j += BS) {
            st_where[j - a_beg] = flat_where(acol, slot_of, tstride, j);
            st_av[j - a_beg] = aval[j];
        }
        for (int p = p_lo; p <= p_hi; p++) {
            const int c0 = p * W;
            if (threadIdx.x == 0) NSP_COUNT(FC_FLAT, 3, 1);
            // ---- one flat walk of the products of this tile ----------------------------------------------------
            int kb = 0;
            for (int b0 = a_beg; b0 < a_end; b0 += BS, kb++) {
                const int nb = a_end - b0 < BS ? a_end - b0 : BS;
                int where = kb == 0 ? e_where[0] : e_where[kFlatEpt - 1];
                real av = kb == 0 ? e_av[0] : e_av[kFlatEpt - 1];
                static_assert(kFlatEpt == 2, "two batches in registers");
                if (kb >= kFlatEpt) {  // (uniform) beyond the register batches: from the slab, coalesced
                    where = kNoEntry;
                    av = 0;
                    if ((int)threadIdx.x < nb) {
                        where = st_where[b0 - a_beg + (int)threadIdx.x];
                        av = st_av[b0 - a_beg + (int)threadIdx.x];
                    }
                }
                const int2 e = flat_extent(brpt, tab, where, p, p + 1);  // (short rows: whole, filtered below)
                const int nch = (e.y - e.x + V - 1) / V;
                s_ext[threadIdx.x] = e;
                s_av[threadIdx.x] = av;
                const int incl = wave_incl_scan(nch);
                if (lane == 63) fs.wsum[w] = incl;
                __syncthreads();
                int base = 0, total = 0;
#pragma unroll
                for (int u = 0; u < NW; u++) {
                    const int c = fs.wsum[u];
                    base += u < w ? c : 0;
                    total += c;
                }
                fs.pref[threadIdx.x] = base + incl - nch;
                __syncthreads();
                for (int ch0 = threadIdx.x; ch0 < total; ch0 += BS * U) {
                    IVecT<V> pk[U];
                    RVecT<V> pv[U];
                    int pn[U];
                    real sc[U];
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        const int ch = ch0 + u * BS;
                        pn[u] = 0;
                        sc[u] = 0;
                        if (ch < total) {
                            int i = 0;  // the last entry whose first chunk is not beyond ch
#pragma unroll
                            for (int step = BS / 2; step >= 1; step >>= 1) {
                                const int j = i + step;
                                if (j < nb && fs.pref[j] <= ch) i = j;
                            }
                            const int2 x = s_ext[i];
                            sc[u] = s_av[i];
                            pn[u] = fetch_chunk<true, V>(bcol, bval, x.x + (ch - fs.pref[i]) * V, x.y, bnnz, pk[u], pv[u]);
                            if (pn[u] > 0) {
                                NSP_COUNT(FC_FLAT, 0, V);  // the vector load brings V entries, whatever pn says
                                NSP_COUNT(FC_FLAT, 1, V);
                                NSP_COUNT(FC_FLAT_EXTENT, 0, pn[u]);  // ... of which inside the extent
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < U; u++) {
#pragma unroll
                        for (int i = 0; i < V; i++) {
                            const unsigned int idx = (unsigned int)(pk[u].v[i] - c0);
                            if (i < pn[u] && idx < (unsigned int)W) {
                                NSP_COUNT(FC_FLAT, 2, 1);
                                flag[idx] = 1;
                                unsafeAtomicAdd(dense + idx, (acc_t)(sc[u] * pv[u].v[i]));
                            }
                        }
                    }
                }
                __syncthreads();  // the next batch overwrites the parked entries
            }
            lds_barrier();
            // ---- ordered emission: wavefront w owns columns [w*R, (w+1)*R) of the tile (as in k_num_tiled) ---------
            const int r0 = w * R;
            unsigned long long msk[IT];
            int cnt = 0;
#pragma unroll
            for (int j = 0; j < IT; j++) {
                msk[j] = __ballot(flag[r0 + j * 64 + lane] != 0);
                cnt += __popcll(msk[j]);
            }
            if (lane == 0) s_wcnt[w] = cnt;
            lds_barrier();
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
                    const int o = wpos + __popcll(m & ((1ull << lane) - 1ull));
                    if (write_col & 1) ccol[o] = c0 + idx;
                    cval[o] = (real)dense[idx];
                    dense[idx] = 0;  // leave the window clean for the next tile
                    flag[idx] = 0;
                }
                wpos += __popcll(m);
            }
            pos += total;
            lds_barrier();
        }
    }
}

// ===================================================================================
//  heavy numeric rows, sparse flavour: list-driven bitmap-ranked tiles, one flat walk each
// ===================================================================================
// The rows k_num_ranked takes (thin over a wide window, or wider than 32 dense tiles) THAT HAVE A COLUMN LIST -- written by
// the symbolic cursor kernel on matrices wider than 2^20 columns (config 5), or C.col itself in a numeric-only re-run.
// Tile and accumulator are k_num_ranked's list-driven ones: a bitmap over W columns set from the next <= CAP entries of
// the list, the prefix of a bitmap word = list position of its first entry, values added at prefix + rank, columns
// leaving as a copy of the list.  What differs:
//   * the tile is cut at PANEL boundaries of the table (it starts at the panel of the next listed column and ends at the
//     last panel boundary not beyond the CAP-th next entry / the bitmap window), so the extent of a B row inside the tile
//     is exact: tab[slot][first panel], tab[slot][last panel + 1];
//   * only when ONE panel holds more than CAP listed columns is the tile cut inside it; the walk then filters by column
//     and the next tile fetches the rest of that panel's entries again;
//   * the products are walked flat (as in k_num_flat): no cursors, no sweep slots, no state.
// Rows without a list stay with k_num_ranked (its `skip_listed` argument makes it leave the listed ones alone).  Own row
// queue (BinState::queue_head3).
template <int BS, int W, int CAP, int G>
__global__ __launch_bounds__(BS) void k_num_ranked_flat(const int *__restrict__ arpt, const int *__restrict__ acol,
                                                        const real *__restrict__ aval,
                                                        const int *__restrict__ brpt, const int *__restrict__ bcol,
                                                        const real *__restrict__ bval, int bnnz,
                                                        const int *__restrict__ slot_of, const int *__restrict__ tab, int tstride,
                                                        const int *__restrict__ crpt, int *__restrict__ ccol,
                                                        real *__restrict__ cval,
                                                        const int *__restrict__ row_perm, int bin_off, int count,
                                                        BinState *bs, const int *__restrict__ row_lo,
                                                        const int *__restrict__ row_span, int write_col, int dens, int tiled_w,
                                                        const int *__restrict__ tcol, const long long *__restrict__ list_off,
                                                        long long list_work, const int *__restrict__ row_prod,
                                                        int *__restrict__ slab, long long stride_ints, int amax)
{
    constexpr int NW = BS / 64;
    constexpr int V = VW, U = 4;
    constexpr int NWORD = W / 32;
    constexpr int WPT = NWORD / BS;       // bitmap words per thread
    constexpr int WG = (W / G) * G;       // whole panels under the bitmap
    constexpr int INF = 0x7fffffff;
    static_assert(NWORD == WPT * BS, "bitmap words split evenly over the threads");
    static_assert(CAP <= 65535, "ranks are kept in 16 bits");
    static_assert(WG >= G, "the bitmap covers at least one panel");
    __shared__ __attribute__((aligned(16))) unsigned int bits[NWORD];
    __shared__ __attribute__((aligned(16))) unsigned short pref[NWORD];
    __shared__ __attribute__((aligned(16))) acc_t vals[CAP];
    __shared__ int2 s_ext[BS];
    __shared__ real s_av[BS];
    __shared__ FlatScratch<BS> fs;
    __shared__ int s_row, s_ntile;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < NWORD; i += BS) bits[i] = 0;
    for (int i = threadIdx.x; i < CAP; i += BS) vals[i] = 0;
    while (true) {
        __syncthreads();
        if (threadIdx.x == 0) s_row = atomicAdd(&bs->queue_head3, 1);
        __syncthreads();
        const int q = s_row;
        if (q >= count) break;
        const int rid = row_perm[bin_off + q];
        const int span = row_span[rid];
        int pos = crpt[rid];
        const int row_nnz = crpt[rid + 1] - pos;
        // the same split as k_num_ranked: thick rows of at most 32 dense tiles belong to the dense-tile kernel
        if (dens > 0 && (long long)row_nnz * dens >= span && span <= 32 * tiled_w) continue;
        if (list_work > 0 && list_wanted(row_nnz, row_prod[rid], list_work) && (list_off == nullptr || list_off[rid] >= 0))
            continue;
        const int *__restrict__ rlist = nullptr;
        if (tcol != nullptr) {  // (list_off == nullptr: a numeric-only re-run, the list is C.col itself)
            if (list_off == nullptr) rlist = tcol + pos;
            else if (list_off[rid] >= 0) rlist = tcol + list_off[rid];
        }
        if (rlist == nullptr) continue;  // k_num_ranked's row
        const int a_beg = arpt[rid], a_end = arpt[rid + 1];
        int e_where[kFlatEpt];
        real e_av[kFlatEpt];
#pragma unroll
        for (int u = 0; u < kFlatEpt; u++) {
            const int j = a_beg + u * BS + (int)threadIdx.x;
            e_where[u] = kNoEntry;
            e_av[u] = 0;
            if (j < a_end) {
                e_where[u] = flat_where(acol, slot_of, tstride, j);
                e_av[u] = aval[j];
            }
        }
        // hub rows: the entries beyond the register batches park theirs in the slab (entry e <-> thread e % BS, so every
        // thread reads back what it wrote itself)
        int *st_where = slab + (long long)blockIdx.x * stride_ints;
        real *st_av = reinterpret_cast<real *>(st_where + amax);
        for (int j = a_beg + kFlatEpt * BS + (int)threadIdx.x; j < a_end; j += BS) {
            st_where[j - a_beg] = flat_where(acol, slot_of, tstride, j);
            st_av[j - a_beg] = aval[j];
        }
        int k0 = 0;
        while (k0 < row_nnz) {
            if (threadIdx.x == 0) NSP_COUNT(FC_RANKED_FLAT, 3, 1);
            const int cnt = row_nnz - k0 < CAP ? row_nnz - k0 : CAP;
            const int first = rlist[k0];
            const int base = first / G * G;
            const long long limit_ll = (long long)base + WG;
            const int limit = limit_ll > INF ? INF : (int)limit_ll;
            const int nxt = k0 + cnt < row_nnz ? rlist[k0 + cnt] : INF;
            int hi = limit;
            if (nxt < limit) {
                hi = nxt / G * G;             // the last panel boundary not beyond the first entry that does not fit
                if (hi <= base) hi = nxt;     // more than CAP listed columns inside one panel: cut inside it
            }
            if (threadIdx.x == 0) s_ntile = cnt;  // unless an entry at or beyond `hi` says otherwise (below)
            lds_barrier();
            for (int k = threadIdx.x; k < cnt; k += BS) {
                const int c = rlist[k0 + k];
                const int cp = k > 0 ? rlist[k0 + k - 1] : -1;
                if (c < hi) {
                    const unsigned int idx = (unsigned int)(c - base);
                    atomicOr(&bits[idx >> 5], 1u << (idx & 31));
                    if (k == 0 || ((unsigned int)(cp - base) >> 5) != (idx >> 5)) pref[idx >> 5] = (unsigned short)k;
                } else if (k == 0 || cp < hi) {
                    s_ntile = k;  // the first entry beyond the tile ends it (exactly one thread sees it)
                }
            }
            lds_barrier();
            const int ntile = s_ntile;
            const int p_a = base / G, p_b = (int)(((long long)hi + G - 1) / G);
            // ---- one flat walk of the products inside [first, hi) --------------------------------------------------
            const int p_bc = p_b < tstride ? p_b : tstride - 1;
            int kb = 0;
            for (int b0 = a_beg; b0 < a_end; b0 += BS, kb++) {
                const int nb = a_end - b0 < BS ? a_end - b0 : BS;
                int where = kb == 0 ? e_where[0] : e_where[kFlatEpt - 1];
                real av = kb == 0 ? e_av[0] : e_av[kFlatEpt - 1];
                if (kb >= kFlatEpt) {  // (uniform) beyond the register batches: from the slab, coalesced
                    where = kNoEntry;
                    av = 0;
                    if ((int)threadIdx.x < nb) {
                        where = st_where[b0 - a_beg + (int)threadIdx.x];
                        av = st_av[b0 - a_beg + (int)threadIdx.x];
                    }
                }
                const int2 e = flat_extent(brpt, tab, where, p_a, p_bc);  // (short rows: whole, filtered below)
                const int nch = (e.y - e.x + V - 1) / V;
                s_ext[threadIdx.x] = e;
                s_av[threadIdx.x] = av;
                const int incl = wave_incl_scan(nch);
                if (lane == 63) fs.wsum[w] = incl;
                __syncthreads();
                int sbase = 0, total = 0;
#pragma unroll
                for (int u = 0; u < NW; u++) {
                    const int c = fs.wsum[u];
                    sbase += u < w ? c : 0;
                    total += c;
                }
                fs.pref[threadIdx.x] = sbase + incl - nch;
                __syncthreads();
                for (int ch0 = threadIdx.x; ch0 < total; ch0 += BS * U) {
                    IVecT<V> pk[U];
                    RVecT<V> pv[U];
                    int pn[U];
                    real sc[U];
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        const int ch = ch0 + u * BS;
                        pn[u] = 0;
                        sc[u] = 0;
                        if (ch < total) {
                            int i = 0;
#pragma unroll
                            for (int step = BS / 2; step >= 1; step >>= 1) {
                                const int j = i + step;
                                if (j < nb && fs.pref[j] <= ch) i = j;
                            }
                            const int2 x = s_ext[i];
                            sc[u] = s_av[i];
                            pn[u] = fetch_chunk<true, V>(bcol, bval, x.x + (ch - fs.pref[i]) * V, x.y, bnnz, pk[u], pv[u]);
                            if (pn[u] > 0) {
                                NSP_COUNT(FC_RANKED_FLAT, 0, V);
                                NSP_COUNT(FC_RANKED_FLAT, 1, V);
                                NSP_COUNT(FC_FLAT_EXTENT, 1, pn[u]);  // ... of which inside the extent (the rest: lanes of the same vector load)
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < U; u++) {
#pragma unroll
                        for (int i = 0; i < V; i++) {
                            const int col = pk[u].v[i];
                            if (i < pn[u] && col >= first && col < hi) {
                                NSP_COUNT(FC_RANKED_FLAT, 2, 1);
                                const unsigned int idx = (unsigned int)(col - base);
                                const unsigned int below = bits[idx >> 5] & ((1u << (idx & 31)) - 1u);
                                unsafeAtomicAdd(vals + (int)pref[idx >> 5] + __popc(below), (acc_t)(sc[u] * pv[u].v[i]));
                            }
                        }
                    }
                }
                __syncthreads();  // the next batch overwrites the parked entries
            }
            lds_barrier();
            // ---- emission: the columns are the list, the values are in list order ------------------------------------
            if (write_col & 1)
                for (int r = threadIdx.x; r < ntile; r += BS) ccol[pos + r] = rlist[k0 + r];
            for (int r = threadIdx.x; r < ntile; r += BS) {
                cval[pos + r] = (real)vals[r];
                vals[r] = 0;
            }
#pragma unroll
            for (int j = 0; j < WPT; j++) bits[threadIdx.x + j * BS] = 0;
            pos += ntile;
            k0 += ntile;
            lds_barrier();
        }
    }
}

// ===================================================================================
//  symbolic twin: bit windows wider than 2^20 columns, one flat walk per tile
// ===================================================================================
// The rows the SYM mode of k_num_ranked takes (symbolic bin 10 on matrices wider than 2^20 columns with sorted B): count
// the distinct columns of the row and, when asked, write them out as the row's sorted list (common.h: bits_to_list) for
// the list-driven numeric tiles.  Same tile (a bitmap over W columns), same listing rule and list allocation; what
// differs is the walk: a tile is the WG = (W / G) * G columns of whole panels starting at the panel of the row's first
// column, the extent of a B row inside it is tab[slot][first panel] .. tab[slot][last panel + 1], and the column ids are
// read flat, four at a time, once -- the cursor kernel loads 1.6-1.9 column ids per product (tests/emu census).
// Own row queue (queue_head2 of the SYMBOLIC counter block, like the kernel it replaces).
template <int BS, int W, int G>
__global__ __launch_bounds__(BS) void k_sym_flat(const int *__restrict__ arpt, const int *__restrict__ acol,
                                                 const int *__restrict__ brpt, const int *__restrict__ bcol, int bnnz,
                                                 const int *__restrict__ slot_of, const int *__restrict__ tab, int tstride,
                                                 const int *__restrict__ row_perm, int bin_off, int count, BinState *bs,
                                                 const int *__restrict__ row_lo, const int *__restrict__ row_span,
                                                 int *__restrict__ row_nz_out, int *__restrict__ tcol,
                                                 long long *__restrict__ list_off, long long list_work,
                                                 const int *__restrict__ row_prod)
{
    constexpr int NW = BS / 64;
    constexpr int V = VW, U = 4;
    constexpr int NWORD = W / 32;
    constexpr int WG = (W / G) * G;
    static_assert(WG >= G, "the bitmap covers at least one panel");
    __shared__ __attribute__((aligned(16))) unsigned int bits[NWORD];
    __shared__ int2 s_ext[BS];
    __shared__ FlatScratch<BS> fs;
    __shared__ int s_wsum[NW];
    __shared__ int s_row;
    __shared__ long long s_off;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < NWORD; i += BS) bits[i] = 0;
    while (true) {
        __syncthreads();
        if (threadIdx.x == 0) s_row = atomicAdd(&bs->queue_head2, 1);
        __syncthreads();
        const int q = s_row;
        if (q >= count) break;
        const int rid = row_perm[bin_off + q];
        const int lo = row_lo[rid], span = row_span[rid];
        bool listing = false;
        if (tcol != nullptr) {
            const int np = row_prod[rid];
            const int lcap = np < span ? np : span;
            // list_work < 0: a list for every row that can be heavy in the numeric phase; > 0: only the rows the listed
            // kernel would take; -2: diagnostics, no stores (the rule of k_num_ranked<SYM>)
            listing = list_work < 0 ? np > kListMinNnz : list_wanted(lcap, np, list_work);
            if (listing && threadIdx.x == 0) {
                s_off = (long long)atomicAdd(&bs->list_cursor, (unsigned long long)lcap);
                list_off[rid] = list_work == -2 ? -1 : s_off;
            }
        }
        __syncthreads();  // s_off
        const int a_beg = arpt[rid], a_end = arpt[rid + 1];
        const long long row_end = (long long)lo + span;
        int sym_cnt = 0;
        int e_where[kFlatEpt];
#pragma unroll
        for (int u = 0; u < kFlatEpt; u++) {
            const int j = a_beg + u * BS + (int)threadIdx.x;
            e_where[u] = j < a_end ? flat_where(acol, slot_of, tstride, j) : kNoEntry;
        }
        for (long long t = (long long)(lo / G) * G; t < row_end; t += WG) {
            const int t0 = (int)t;
            const int p_a = t0 / G, p_b = p_a + W / G;
            if (threadIdx.x == 0) NSP_COUNT(FC_SYM_FLAT, 3, 1);
            const int p_bc = p_b < tstride ? p_b : tstride - 1;
            int kb = 0;
            for (int b0 = a_beg; b0 < a_end; b0 += BS, kb++) {
                const int nb = a_end - b0 < BS ? a_end - b0 : BS;
                int where = kb == 0 ? e_where[0] : e_where[kFlatEpt - 1];
                if (kb >= kFlatEpt)  // (uniform) beyond the register batches
                    where = (int)threadIdx.x < nb ? flat_where(acol, slot_of, tstride, b0 + (int)threadIdx.x) : kNoEntry;
                const int2 e = flat_extent(brpt, tab, where, p_a, p_bc);  // (short rows: whole, filtered below)
                const int nch = (e.y - e.x + V - 1) / V;
                s_ext[threadIdx.x] = e;
                const int incl = wave_incl_scan(nch);
                if (lane == 63) fs.wsum[w] = incl;
                __syncthreads();
                int sbase = 0, total = 0;
#pragma unroll
                for (int u = 0; u < NW; u++) {
                    const int c = fs.wsum[u];
                    sbase += u < w ? c : 0;
                    total += c;
                }
                fs.pref[threadIdx.x] = sbase + incl - nch;
                __syncthreads();
                for (int ch0 = threadIdx.x; ch0 < total; ch0 += BS * U) {
                    IVecT<V> pk[U];
                    RVecT<1> pv;
                    int pn[U];
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        const int ch = ch0 + u * BS;
                        pn[u] = 0;
                        if (ch < total) {
                            int i = 0;
#pragma unroll
                            for (int step = BS / 2; step >= 1; step >>= 1) {
                                const int j = i + step;
                                if (j < nb && fs.pref[j] <= ch) i = j;
                            }
                            const int2 x = s_ext[i];
                            pn[u] = fetch_chunk<false, V>(bcol, (const real *)nullptr, x.x + (ch - fs.pref[i]) * V, x.y, bnnz, pk[u], pv);
                            if (pn[u] > 0) {
                                NSP_COUNT(FC_SYM_FLAT, 0, V);
                                NSP_COUNT(FC_FLAT_EXTENT, 2, pn[u]);
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < U; u++) {
#pragma unroll
                        for (int i = 0; i < V; i++) {
                            const unsigned int idx = (unsigned int)(pk[u].v[i] - t0);
                            if (i < pn[u] && idx < (unsigned int)WG) {
                                NSP_COUNT(FC_SYM_FLAT, 2, 1);
                                atomicOr(&bits[idx >> 5], 1u << (idx & 31));
                            }
                        }
                    }
                }
                __syncthreads();  // the next batch overwrites the parked entries
            }
            lds_barrier();
            sym_cnt += bits_to_list<BS, true>(bits, NWORD, t0, listing ? tcol + s_off + sym_cnt : (int *)nullptr, s_wsum,
                                              list_work == -2);
        }
        if (threadIdx.x == 0) row_nz_out[rid] = sym_cnt;  // nnz of the row = bits seen over all tiles (uniform)
    }
}

}  // namespace spgemm
}  // namespace nsp
