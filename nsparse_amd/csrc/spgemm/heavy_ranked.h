// spgemm/heavy_ranked.h -- heavy numeric rows, bitmap-ranked accumulator.
// Part of the spgemm_hash.hip translation unit (kernels are launched from its host code).
#pragma once
#include "common.h"

namespace nsp {
namespace spgemm {

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
//
// SYM = true is the SYMBOLIC twin for rows whose window is wider than the 2^20-bit window of
// k_sym_bits: the same cursors, one walk per tile that only sets bits (and commits), a popcount,
// no ranks, no values -- W = 2^20 columns per tile.  k_sym_bits covers such a window in pieces and
// walks ALL products for every piece (R-MAT-22: 4 pieces); with cursors every product is seen once.
// Needs sorted rows of B like the numeric kernels; writes row_nz_out[rid].
template <int BS, int W, int CAP, int LCAP, bool SYM = false>
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
                                                   unsigned long long *prof, int *__restrict__ row_nz_out = nullptr,
                                                   int *__restrict__ tcol = nullptr,
                                                   long long *__restrict__ list_off = nullptr,
                                                   long long list_work = 0, const int *__restrict__ row_prod = nullptr,
                                                   int skip_listed = 0)
{
    // skip_listed (experiments build): rows that have a column list belong to k_num_ranked_flat (heavy_flat.h)
    // SYM with tcol != nullptr: the columns of every tile are also written out as the row's sorted list
    // (common.h: bits_to_list) when common.h: list_wanted says so.  Numeric with list_work > 0: rows that
    // list_wanted picks and that have a list (list_off == nullptr: every such row, the list is C.col itself)
    // belong to k_num_listed.
    // prof (NSPARSE_TILED_PROF=1), 100 MHz ticks of thread 0: 0 set-up, 1 pass 1, 2 scan, 3 pass 2,
    // 4 emission; 5 tiles, 6 rows
    unsigned long long tk = (kExperiments && prof) ? wall_clock64() : 0;
    unsigned long long t_acc[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto tick = [&](int phase) {
        if (kExperiments && prof) {
            const unsigned long long now = wall_clock64();
            t_acc[phase] += now - tk;
            tk = now;
        }
    };
    constexpr int NW = BS / 64;
    constexpr int EPT = 4, VMAX = 32, LA = 4;
    constexpr int INF = 0x7fffffff;
    constexpr int NWORD = W / 32;
    constexpr int WPT = NWORD / BS;  // bitmap words per thread
    static_assert(NWORD == WPT * BS && WPT % 2 == 0, "an even number of bitmap words per thread");
    static_assert(CAP <= 65535, "ranks are kept in 16 bits");
    __shared__ __attribute__((aligned(16))) unsigned int bits[NWORD];
    __shared__ __attribute__((aligned(16))) unsigned short pref[SYM ? 8 : NWORD];
    __shared__ __attribute__((aligned(16))) acc_t vals[SYM ? 8 : CAP];
    __shared__ int4 l_meta[LCAP];
    __shared__ real l_av[SYM ? 8 : LCAP];  // (the symbolic twin carries no values)
    __shared__ int s_row, s_nlong, s_cut, s_ntile;
    __shared__ int s_wsum[NW];
    __shared__ long long s_off;
    int *st_cur = slab + (long long)blockIdx.x * stride_ints;
    int *st_end = st_cur + amax;
    int *st_next = st_end + amax;
    real *st_av = reinterpret_cast<real *>(st_next + amax);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < NWORD; i += BS) bits[i] = 0;
    for (int i = threadIdx.x; i < (SYM ? 8 : CAP); i += BS) vals[i] = 0;
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
        int pos = SYM ? 0 : crpt[rid];
        const int pos0 = pos;
        const int row_nnz = SYM ? 0 : crpt[rid + 1] - pos;
        // dens > 0: only rows thinner than one non-zero per `dens` columns or wider than 32 dense
        // tiles (the rest belong to k_num_tiled); dens <= 0: every row
        if (!SYM && dens > 0 && (long long)row_nnz * dens >= span && span <= 32 * tiled_w) continue;
        if (!SYM && list_work > 0 && list_wanted(row_nnz, row_prod[rid], list_work) &&
            (list_off == nullptr || list_off[rid] >= 0))
            continue;
        // numeric, tcol != nullptr and a list for this row: LIST-DRIVEN tiles (below)
        const int *__restrict__ rlist = nullptr;
        if (!SYM && tcol != nullptr) {  // (list_off == nullptr: a numeric-only re-run, the list is C.col itself)
            if (list_off == nullptr) rlist = tcol + pos;
            else if (list_off[rid] >= 0) rlist = tcol + list_off[rid];
        }
        if (kExperiments && !SYM && skip_listed && rlist != nullptr) continue;
        bool listing = false;
        if (SYM && tcol != nullptr) {
            const int np = row_prod[rid];
            const int lcap = np < span ? np : span;
            // list_work < 0: a list for every row that can be heavy in the numeric phase (the list-driven tiles of
            // this kernel's numeric twin); > 0: only the rows the listed kernel will take (common.h: list_wanted)
            listing = list_work < 0 ? np > kListMinNnz : list_wanted(lcap, np, list_work);  // (-2: diagnostics, no stores)
            if (listing && threadIdx.x == 0) {
                s_off = (long long)atomicAdd(&bs->list_cursor, (unsigned long long)lcap);
                list_off[rid] = list_work == -2 ? -1 : s_off;
            }
        }
        const int a_beg = arpt[rid], alen = arpt[rid + 1] - a_beg;
        const int split = 64 * ((span + W - 1) / W + row_nnz / CAP + 1);
        auto init_entry = [&](int e, int &cur, int &end, real &av) -> bool {
            const int c = acol[a_beg + e];
            cur = brpt[c];
            end = brpt[c + 1];
            av = SYM ? (real)0 : aval[a_beg + e];
            const int len = end - cur;
            if (len <= LONG_LEN) return true;
            int V = (len + split - 1) / split;
            V = V < 1 ? 1 : (V > VMAX ? VMAX : V);
            const int li = atomicAdd(&s_nlong, V);
            const bool fits = li + V <= LCAP;
            for (int v = 0; v < V && li + v < LCAP; v++) {
                l_meta[li + v] = fits ? make_int4(cur + 64 * v, end, 64 * V, 0) : make_int4(0, 0, 64, 0);
                if (!SYM) l_av[li + v] = av;
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
                e_nc[u] = NSP_LDC((SYM ? FC_RANKED_SYM : FC_RANKED), bcol, e_cur[u]);
        }
        for (int e = threadIdx.x + EPT * BS; e < alen; e += BS) {
            int cur, end;
            real av;
            const bool serial = init_entry(e, cur, end, av);
            st_cur[e] = cur;
            st_end[e] = end;
            st_next[e] = (serial && cur < end) ? NSP_LDC((SYM ? FC_RANKED_SYM : FC_RANKED), bcol, cur) : INF;
            st_av[e] = av;
        }
        __syncthreads();
        const int nlong = s_nlong < LCAP ? s_nlong : LCAP;
        const int row_end = lo + span;
        int t_lo = lo;
        int sym_cnt = 0;
        tick(0);
        if (kExperiments && prof) {
            t_acc[6]++;
            t_acc[10] += nlong;
            t_acc[11] += alen;
            t_acc[12] += s_nlong > LCAP;
        }
        // LIST-DRIVEN tiles (round 3): the symbolic phase has written the row's sorted columns (common.h:
        // bits_to_list), so a tile needs neither pass 1 nor the scan nor the cut: its bitmap is set from the
        // next <= CAP entries of the list (coalesced), the prefix of a bitmap word is the list position of the
        // word's first entry (a plain store by the thread that holds it: every word that is ever looked up has
        // one), the tile starts AT the next listed column and its columns leave as a copy of the list.  One
        // cursor walk per tile instead of two: 14.7 -> 7.9 us per tile, R-MAT-22 numeric heavy bin 41 -> 30 ms.
        //
        // Tiles BY ENTRIES were tried as well: the next ~11 K entries of the list whatever width they span -- a thin
        // R-MAT-22 row is 2 tiles instead of 8.6 -- as columns + accumulators in the same LDS, a product's
        // accumulator found by a bucket table over 2048 column ranges and a short binary search.  360 K tiles instead
        // of 726 K, and the heavy bin took 36.5 ms against 34.1 on the same box: a tile of four times the products
        // is four times the dependent steps of the cursor walk, the search diverges in the dense low-column stretch
        // of a power-law row -- and the second code path cost the WHOLE kernel its registers (50 scratch
        // instructions: the rows without a list went from 41 to 57 ms).  Removed.
        if (!SYM && rlist != nullptr) t_lo = rlist[0];
        while (SYM || rlist == nullptr ? t_lo < row_end : pos - pos0 < row_nnz) {
            // Pass 1 runs to t_max and everything beyond the cut is walked again by the next tile,
            // so t_max aims at ~7/8 of CAP columns at the density of what is left of the row.
            int t_max;
            if (!SYM && rlist != nullptr) {
                t_max = row_end - t_lo <= W ? row_end : t_lo + W;
            } else if (SYM) {
                t_max = row_end - t_lo <= W ? row_end : t_lo + W;
            } else {
                const long long rem_nnz = row_nnz - (pos - pos0), rem_span = row_end - t_lo;
                long long wd_est = rem_nnz > 0 ? (long long)(CAP - CAP / 8) * rem_span / rem_nnz : rem_span;
                wd_est = (wd_est + 31) & ~31LL;
                if (wd_est > W) wd_est = W;
                if (wd_est < 1024) wd_est = 1024;
                t_max = rem_span <= wd_est ? row_end : t_lo + (int)wd_est;
            }
            if (threadIdx.x == 0) NSP_COUNT(SYM ? FC_RANKED_SYM : FC_RANKED, 3, 1);
            // One walk of everything inside [t_lo, t_hi).  PASS2 = false: mark columns.
            // PASS2 = true: accumulate by rank and commit the cursors.
            auto touch = [&](auto pass2, int col, real x) {
                const unsigned int idx = (unsigned int)(col - t_lo);
                if (decltype(pass2)::value) NSP_COUNT(SYM ? FC_RANKED_SYM : FC_RANKED, 2, 1);
                if (SYM || !decltype(pass2)::value) {
                    atomicOr(&bits[idx >> 5], 1u << (idx & 31));
                } else {
                    const unsigned int below = bits[idx >> 5] & ((1u << (idx & 31)) - 1u);
                    unsafeAtomicAdd(vals + (int)pref[idx >> 5] + __popc(below), (acc_t)x);
                }
            };
            // lane-serial entries: LA consecutive (column, value) pairs per round trip; the first
            // batch of all register entries is requested before any of it is used
            auto load_batch = [&](auto pass2, int k, int end, int c0, int(&c)[LA], real(&v)[LA]) {
                c[0] = c0;
#pragma unroll
                for (int j = 1; j < LA; j++) c[j] = k + j < end ? NSP_LDC((SYM ? FC_RANKED_SYM : FC_RANKED), bcol, k + j) : INF;
                if (!SYM && decltype(pass2)::value) {
#pragma unroll
                    for (int j = 0; j < LA; j++) v[j] = k + j < end ? NSP_LDV((SYM ? FC_RANKED_SYM : FC_RANKED), bval, k + j) : (real)0;
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
                            touch(pass2, c[j], (!SYM && decltype(pass2)::value) ? av * v[j] : (real)0);
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
                col = k < end ? NSP_LDC((SYM ? FC_RANKED_SYM : FC_RANKED), bcol, k) : INF;
                while (col < t_hi) {
                    int c[LA];
                    real v[LA];
                    load_batch(pass2, k, end, col, c, v);
                    const int n = consume(pass2, t_hi, c, v, av, col);
                    k += n;
                    if (n < LA) return;
                    col = k < end ? NSP_LDC((SYM ? FC_RANKED_SYM : FC_RANKED), bcol, k) : INF;
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
                        col[j] = kk < mt[j].y ? NSP_LDC((SYM ? FC_RANKED_SYM : FC_RANKED), bcol, kk) : INF;
                        bv[j] = 0;
                        if (P2 && !SYM) bv[j] = kk < mt[j].y ? NSP_LDV((SYM ? FC_RANKED_SYM : FC_RANKED), bval, kk) : (real)0;
                    }
                };
                auto sweep_use = [&](int i0, const int4(&mt)[SB], const int(&col)[SB], const real(&bv)[SB]) {
#pragma unroll
                    for (int j = 0; j < SB; j++) {
                        const int i = i0 + j * NW;
                        if (i >= nlong) continue;
                        const real av = SYM ? (real)0 : l_av[i];
                        int k = mt[j].x, c = col[j];
                        real x = bv[j];
                        while (true) {
                            if (c >= t_lo && c < t_hi) touch(pass2, c, av * x);
                            if (__builtin_amdgcn_readlane(c, 63) >= t_hi) break;
                            k += mt[j].z;
                            const int kk = k + lane;
                            c = kk < mt[j].y ? NSP_LDC((SYM ? FC_RANKED_SYM : FC_RANKED), bcol, kk) : INF;
                            if (P2 && !SYM) x = kk < mt[j].y ? NSP_LDV((SYM ? FC_RANKED_SYM : FC_RANKED), bval, kk) : (real)0;
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
            if (SYM) {  // one committing walk that only sets bits, then count (list) and clear
                walk(std::true_type{}, t_max);
                lds_barrier();
                sym_cnt += bits_to_list<BS, true>(bits, NWORD, t_lo, listing ? tcol + s_off + sym_cnt : (int *)nullptr, s_wsum,
                                                  list_work == -2);
                t_lo = t_max;
                continue;
            }
            int t_hi_l = 0, ntile_l = 0;
            if (rlist != nullptr) {
                const int k0 = pos - pos0;
                const int cnt = row_nnz - k0 < CAP ? row_nnz - k0 : CAP;
                if (threadIdx.x == 0) {
                    s_ntile = cnt;  // unless an entry beyond the window says otherwise (below)
                    s_cut = k0 + cnt < row_nnz ? rlist[k0 + cnt] : row_end;
                }
                lds_barrier();
                for (int k = threadIdx.x; k < cnt; k += BS) {
                    const int c = rlist[k0 + k];
                    if (c < t_max) {
                        const unsigned int idx = (unsigned int)(c - t_lo);
                        atomicOr(&bits[idx >> 5], 1u << (idx & 31));
                        const int cp = k > 0 ? rlist[k0 + k - 1] : -1;
                        if (k == 0 || ((unsigned int)(cp - t_lo) >> 5) != (idx >> 5)) pref[idx >> 5] = (unsigned short)k;
                    } else if (k == 0 || rlist[k0 + k - 1] < t_max) {
                        s_ntile = k;   // the first entry beyond the window ends the tile (one thread sees it)
                        s_cut = t_max;
                    }
                }
                lds_barrier();
                t_hi_l = s_cut;
                ntile_l = s_ntile;
                tick(t_lo == lo ? 7 : 1);
            } else {
            walk(std::false_type{}, t_max);
            lds_barrier();
            tick(t_lo == lo ? 7 : 1);
            // ---- scan: thread t scans its WPT consecutive words (one wave scan, one barrier); the
            // columns are written out below with strided ownership (thread t: words t, t + BS, ...),
            // which shares the dense low-column stretch of a power-law row among many threads.
            // (Strided ownership in the scan as well costs a wave scan per word slot: R-MAT-22 +3 %;
            //  consecutive ownership in the emission costs balance: +5 %.)
            auto widx = [&](int j) { return (int)threadIdx.x * WPT + j; };
            unsigned int wd[WPT];
            int pc[WPT];  // becomes the exclusive prefix of the word
            int total;
            int tsum = 0;
            static_assert(WPT % 4 == 0, "vector loads of the thread's words");
#pragma unroll
            for (int j = 0; j < WPT; j += 4) {
                const uint4 q = reinterpret_cast<const uint4*>(bits)[(threadIdx.x * WPT + j) >> 2];
                wd[j] = q.x, wd[j + 1] = q.y, wd[j + 2] = q.z, wd[j + 3] = q.w;
            }
#pragma unroll
            for (int j = 0; j < WPT; j++) {
                pc[j] = tsum;
                tsum += __popc(wd[j]);
            }
            const int incl = wave_incl_scan(tsum);
            if (lane == 63) s_wsum[w] = incl;
            if (threadIdx.x == 0) s_cut = t_max;
            lds_barrier();
            int base = incl - tsum;
            total = 0;
#pragma unroll
            for (int u = 0; u < NW; u++) {
                const int c = s_wsum[u];
                base += u < w ? c : 0;
                total += c;
            }
#pragma unroll
            for (int j = 0; j < WPT; j++) pc[j] += base;
#pragma unroll
            for (int j = 0; j < WPT; j += 8) {
                uint4 q;
                // (prefixes past the cut may exceed 16 bits: they are never read, but must not spill
                //  into their neighbour)
                q.x = (pc[j] & 0xffff) | (pc[j + 1] << 16), q.y = (pc[j + 2] & 0xffff) | (pc[j + 3] << 16);
                q.z = (pc[j + 4] & 0xffff) | (pc[j + 5] << 16), q.w = (pc[j + 6] & 0xffff) | (pc[j + 7] << 16);
                reinterpret_cast<uint4*>(pref)[(threadIdx.x * WPT + j) >> 3] = q;
            }
            if (total > CAP) {  // cut at the word where the running count would pass CAP
#pragma unroll
                for (int j = 0; j < WPT; j++) {
                    const int p1 = pc[j] + __popc(wd[j]);
                    if (pc[j] <= CAP && p1 > CAP) {
                        s_cut = t_lo + 32 * widx(j);
                        s_ntile = pc[j];
                    }
                }
            } else if (threadIdx.x == 0) {
                s_ntile = total;
            }
            lds_barrier();
            }  // pass 1 + scan (rows without a list)
            const int t_hi = rlist ? t_hi_l : s_cut, ntile = rlist ? ntile_l : s_ntile;
            tick(2);
            if (kExperiments && prof && t_hi != t_max) t_acc[9]++;
            walk(std::true_type{}, t_hi);
            lds_barrier();
            tick(t_lo == lo ? 8 : 3);
            // ---- emission (words and prefixes are read back: not kept live across pass 2) ------
            if ((write_col & 1) && rlist != nullptr) {
                const int k0 = pos - pos0;
                for (int r = threadIdx.x; r < ntile; r += BS) ccol[pos + r] = rlist[k0 + r];
            } else if (write_col & 1) {
#pragma unroll
                for (int j = 0; j < WPT; j++) {
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
                cval[pos + r] = (real)vals[r];
                vals[r] = 0;
            }
#pragma unroll
            for (int j = 0; j < WPT; j++) bits[threadIdx.x + j * BS] = 0;
            pos += ntile;
            t_lo = (rlist != nullptr && pos - pos0 < row_nnz) ? rlist[pos - pos0] : t_hi;
            lds_barrier();
            tick(4);
            if (kExperiments && prof) t_acc[5]++;
        }
        if (SYM && threadIdx.x == 0) row_nz_out[rid] = sym_cnt;  // nnz of the row = bits seen over all tiles (uniform)
    }
    if (kExperiments && prof && threadIdx.x == 0)
        for (int i = 0; i < 13; i++) atomicAdd(prof + 16 + i, t_acc[i]);
}

}  // namespace spgemm
}  // namespace nsp
