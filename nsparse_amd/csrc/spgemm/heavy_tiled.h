// spgemm/heavy_tiled.h -- heavy numeric rows, dense column tiles.
// Part of the spgemm_hash.hip translation unit (kernels are launched from its host code).
#pragma once
#include "common.h"

namespace nsp {
namespace spgemm {

// ===================================================================================
//  heavy numeric rows: column-tiled dense windows
// ===================================================================================
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
                                                  int LONG_LEN, unsigned long long *prof, int dens,
                                                  const long long *__restrict__ list_off = nullptr,
                                                  long long list_work = 0, const int *__restrict__ row_prod = nullptr)
{
    // prof (NSPARSE_TILED_PROF=1): thread 0 adds 100 MHz ticks per phase -- 0 cursor set-up,
    // 2 register-fed accumulation, 3 overflow paths, 4 emission, 5 tiles, 6 rows
    // (kept in registers, one atomic per counter when the workgroup retires)
    unsigned long long tk = (kExperiments && prof) ? wall_clock64() : 0;
    unsigned long long t_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto tick = [&](int phase) {
        if (kExperiments && prof) {
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
    __shared__ __attribute__((aligned(16))) acc_t dense[W];
    __shared__ __attribute__((aligned(16))) unsigned int flag4[W / 4];
    __shared__ int4 l_meta[LCAP];   // sweep list: (chunk position, row end, stride, next unused column)
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
        // rows that have a column list and that common.h: list_wanted picks belong to k_num_listed (listed.h)
        if (list_work > 0 && list_wanted(crpt[rid + 1] - crpt[rid], row_prod[rid], list_work) &&
            (list_off == nullptr || list_off[rid] >= 0))
            continue;
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
                    e_c0[u] = NSP_LDC(FC_TILED, bcol, k);
                    e_v0[u] = NSP_LDV(FC_TILED, bval, k);
                }
                if (k + 1 < end) {
                    e_c1[u] = NSP_LDC(FC_TILED, bcol, k + 1);
                    e_v1[u] = NSP_LDV(FC_TILED, bval, k + 1);
                }
            }
        }
        for (int e = threadIdx.x + EPT * BS; e < alen; e += BS) {
            int cur, end;
            real av;
            const bool serial = init_entry(e, cur, end, av);
            st_cur[e] = cur;
            st_end[e] = end;
            st_next[e] = (serial && cur < end) ? NSP_LDC(FC_TILED, bcol, cur) : INF;
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
                    pa_col[s] = NSP_LDC(FC_TILED, bcol, ka);
                    pa_val[s] = NSP_LDV(FC_TILED, bval, ka);
                }
                if (kb < mt.y) {
                    pb_col[s] = NSP_LDC(FC_TILED, bcol, kb);
                    pb_val[s] = NSP_LDV(FC_TILED, bval, kb);
                }
            }
        }
        tick(0);
        if (kExperiments && prof) t_acc[6]++;
        int pos = crpt[rid];
        for (int t0 = 0; t0 < span; t0 += W) {
            const int tw = span - t0 < W ? span - t0 : W;  // columns in this tile
            if (threadIdx.x == 0) NSP_COUNT(FC_TILED, 3, 1);
            const int c0 = lo + t0, tile_end = c0 + tw;
            auto acc = [&](int col, real x) {
                const int idx = col - c0;
                NSP_COUNT(FC_TILED, 2, 1);
                flag[idx] = 1;
                unsafeAtomicAdd(dense + idx, (acc_t)x);
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
                if (kExperiments && prof) t_acc[10]++;
#pragma unroll
                for (int s = 0; s < KS; s++) {
                    if (!fresh[s]) continue;  // wave-uniform
                    if ((unsigned)(pa_col[s] - c0) < (unsigned)tw) acc(pa_col[s], l_av[w + s * NW] * pa_val[s]);
                    fresh[s] = __builtin_amdgcn_readlane(pa_col[s], 63) < tile_end;
                    if (fresh[s]) {  // chunk A used up: B moves in, the one after B is requested
                        const int i = w + s * NW;
                        int4 mt = l_meta[i];
                        // every lane has read the record before lane 0 advances it: program order of the wavefront,
                        // said aloud (a scheduling barrier, no instruction; the sync point of tests/emu)
                        __builtin_amdgcn_wave_barrier();
                        mt.x += mt.z;
                        if (lane == 0) l_meta[i].x = mt.x;
                        pa_col[s] = pb_col[s];
                        pa_val[s] = pb_val[s];
                        const int k = mt.x + mt.z + lane;
                        const bool ok = k < mt.y;
                        pb_col[s] = ok ? NSP_LDC(FC_TILED, bcol, k) : INF;
                        pb_val[s] = ok ? NSP_LDV(FC_TILED, bval, k) : (real)0;
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
                            e_c0[u] = ok ? NSP_LDC(FC_TILED, bcol, k) : INF;
                            e_v0[u] = ok ? NSP_LDV(FC_TILED, bval, k) : (real)0;
                            more = true;
                        } else {
                            e_cur[u] += 1;
                            e_c0[u] = e_c1[u];
                            e_v0[u] = e_v1[u];
                        }
                        const int k1 = e_cur[u] + 1;
                        const bool ok1 = k1 < e_end[u];
                        e_c1[u] = ok1 ? NSP_LDC(FC_TILED, bcol, k1) : INF;
                        e_v1[u] = ok1 ? NSP_LDV(FC_TILED, bval, k1) : (real)0;
                    }
                }
            } while (__any(more));
            if (kExperiments && prof) {
                __syncthreads();
                tick(2);
            }
            // ---- overflow paths: state in LDS / global memory --------------------------------
            // sweep slots beyond the register ones: same chunk walk with the state in LDS.  A slot
            // remembers the first column it has not used yet (l_meta.w), so tiles it has nothing in
            // cost no load, and the current chunks of SB slots are requested together.
            constexpr int SB = 4;
            for (int i0 = w + KS * NW; i0 < nlong; i0 += SB * NW) {
                int4 mt[SB];
                int col[SB];
                real bv[SB];
#pragma unroll
                for (int j = 0; j < SB; j++) {
                    const int i = i0 + j * NW;
                    mt[j] = i < nlong ? l_meta[i] : make_int4(0, 0, 64, INF);
                    const int k = mt[j].x + lane;
                    const bool ld = mt[j].w < tile_end && k < mt[j].y;
                    col[j] = ld ? NSP_LDC(FC_TILED, bcol, k) : INF;
                    bv[j] = ld ? NSP_LDV(FC_TILED, bval, k) : (real)0;
                }
#pragma unroll
                for (int j = 0; j < SB; j++) {
                    if (mt[j].w >= tile_end) continue;  // wave-uniform
                    const int i = i0 + j * NW;
                    const real av = l_av[i];
                    int c = col[j];
                    real x = bv[j];
                    int cp = mt[j].x;  // chunk position
                    while (true) {
                        if ((unsigned)(c - c0) < (unsigned)tw) acc(c, av * x);
                        if (__builtin_amdgcn_readlane(c, 63) >= tile_end) break;
                        cp += mt[j].z;
                        const int k = cp + lane;
                        c = k < mt[j].y ? NSP_LDC(FC_TILED, bcol, k) : INF;
                        x = k < mt[j].y ? NSP_LDV(FC_TILED, bval, k) : (real)0;
                    }
                    // columns ascend across the lanes: the first lane at or beyond the tile end holds
                    // the next column this slot will contribute
                    const unsigned long long beyond = __ballot(c >= tile_end);
                    const int nxt = __builtin_amdgcn_readlane(c, __ffsll((long long)beyond) - 1);
                    if (lane == 0) {
                        l_meta[i].x = cp;
                        l_meta[i].w = nxt;
                    }
                }
            }
            // A entries beyond EPT * BS (cursor in the global slice): the state reads of all of a
            // thread's entries go out together, then LA look-ahead pairs per round trip
            constexpr int LA = 4;
            for (int e = threadIdx.x + EPT * BS; e < alen; e += BS) {
                int col = st_next[e];
                if (col < tile_end) {
                    int cur = st_cur[e];
                    const int end = st_end[e];
                    const real av = st_av[e];
                    while (col < tile_end) {
                        int c[LA];
                        real v[LA];
                        c[0] = col;
#pragma unroll
                        for (int j = 1; j < LA; j++) c[j] = cur + j < end ? NSP_LDC(FC_TILED, bcol, cur + j) : INF;
#pragma unroll
                        for (int j = 0; j < LA; j++) v[j] = cur + j < end ? NSP_LDV(FC_TILED, bval, cur + j) : (real)0;
                        int n = 0;
                        col = INF;
#pragma unroll
                        for (int j = 0; j < LA; j++) {
                            if (n == j) {
                                if (c[j] < tile_end) {
                                    acc(c[j], av * v[j]);
                                    n = j + 1;
                                } else {
                                    col = c[j];
                                }
                            }
                        }
                        cur += n;
                        if (n == LA) col = cur < end ? NSP_LDC(FC_TILED, bcol, cur) : INF;
                    }
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
                    cval[p] = (real)dense[idx];
                    dense[idx] = 0;  // leave the window clean for the next tile
                    flag[idx] = 0;
                }
                wpos += __popcll(m);
            }
            pos += total;
            tick(9);
            lds_barrier();
            tick(4);
            if (kExperiments && prof) t_acc[5]++;
        }
    }
    if (kExperiments && prof && threadIdx.x == 0)
        for (int i = 0; i < 12; i++) atomicAdd(prof + i, t_acc[i]);
}

}  // namespace spgemm
}  // namespace nsp
