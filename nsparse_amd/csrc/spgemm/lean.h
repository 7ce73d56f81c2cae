// spgemm/lean.h -- the hash bins on an instruction diet (round 4).
// Part of the spgemm_hash.hip translation unit (kernels are launched from its host code).
//
// Same algorithm as symbolic.h / numeric.h (set_row_nz_bin_each_tb :399-472, calculate_value_col_bin_each_tb
// :829-927 of kernel_spgemm_hash_d.cu: one workgroup per row, LDS open-addressing table, linear probing, compaction,
// ascending columns), re-written after counting where the instructions of k_num_tb<64, 256, 256> went on the 27-point
// stencil (rocprofv3: 814 VALU + 841 SALU + ~200 LDS wave-instructions per row of 729 products; the kernel is
// VALU-bound: 814 x 4 cycles x 977 rows per SIMD = 1.33 of its 1.88 ms):
//
//   * the product walk (54 VALU per step of 4 products per lane, 4.5 steps per row): lane groups per B row,
//     a per-group state machine over (entry, chunk), parked long rows, a "mixed" decision per row.  Here: every
//     thread parks ONE entry of A with its B extent, a scan gives every 4-entry chunk of the batch a number, an
//     OWNER ARRAY in LDS (one byte per chunk: scatter of the entry numbers at their first chunk, max-scan) tells
//     chunk q its entry, and thread t simply takes chunks t, t + BS, ...  No groups, no deferral, no decisions;
//     ~25 VALU per step, every lane busy whatever the B row lengths (stencil: 3 steps per row instead of 4.5).
//   * the probe: CAS(slot, -1, key) for the 4 keys back to back as before; elements of a partial chunk duplicate
//     element 0 instead of being predicated (no exec-mask region per element).
//   * the hash: v_mul_u32_u24 (full rate) instead of v_mul_lo_u32 (quarter rate: 16 cycles per wavefront).
//   * the sort of <= 128 keys of a one-wavefront row: all 28 stages of the bitonic network in registers in its
//     "flip" form (every comparator points the same way: no direction masks), partner values by DPP where the
//     pattern is one (xor 1 / 2 / 8, mirrors of 4 / 8 / 16 lanes) and by ds_swizzle / ds_bpermute otherwise (the LDS
//     crossbar costs no VALU slot): ~190 VALU against 329.
#pragma once
#include "common.h"

namespace nsp {
namespace spgemm {

// Slot of a key in a table of 2^L slots: multiplicative (Fibonacci) hashing in 24-bit arithmetic -- the top L of the
// low 24 bits of key * 0x9E3779 (= 2^24 / phi): one full-rate v_mul_u32_u24 and one v_bfe_u32, where the 32-bit form
// (common.h: hash_slot) needs the quarter-rate v_mul_lo_u32.  Columns that differ only above bit 23 share a slot
// sequence: matrices beyond 16 M columns probe longer, nothing else.  `shift` = 24 - L.
__device__ __forceinline__ int lean_slot(int key, int shift, int bits)
{
    return (int)__builtin_amdgcn_ubfe((unsigned)__umul24((unsigned)key & 0xffffffu, 0x9E3779u), (unsigned)shift, (unsigned)bits);
}

// inclusive max-scan over the 64 lanes (values >= 0)
__device__ __forceinline__ int wave_incl_max(int v)
{
    int t;
    t = __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false); v = v > t ? v : t;  // row_shr:1
    t = __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false); v = v > t ? v : t;  // row_shr:2
    t = __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false); v = v > t ? v : t;  // row_shr:4
    t = __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false); v = v > t ? v : t;  // row_shr:8
    t = __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); v = v > t ? v : t;  // row_bcast:15
    t = __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false); v = v > t ? v : t;  // row_bcast:31
    return v;
}

template <int BS, bool WITH_VAL>
struct LeanScratch {
    static constexpr int CAP = 4 * BS;  // chunks per window of the owner array
    using own_t = typename std::conditional<(BS <= 128), unsigned char, unsigned short>::type;  // entry number + 1
    int2 ent[BS];                       // b - 4 * (first chunk), e: chunk q of the batch starts at entry .x + 4 q
    real av[WITH_VAL ? BS : 1];
    own_t own[CAP];
    int wtot[BS / 64 + 1];
    int wmax[BS / 64 + 1];
};

// Walk every intermediate product of one C row; `consume(k, v, n, sc)` gets 1 <= n <= 4 consecutive entries of a
// B row and the A value.  Must be called by every thread of the workgroup (barriers inside when BS > 64).
// U chunks are requested per thread before the first is consumed.
// PIPE (experiments): one chunk in flight behind the one being hashed instead of U requested together.
template <int BS, bool WITH_VAL, int U, bool PIPE, typename F>
__device__ __forceinline__ void lean_walk(const int *__restrict__ acol, const real *__restrict__ aval,
                                          const int *__restrict__ brpt, const int *__restrict__ bcol,
                                          const real *__restrict__ bval, int bnnz, int a_beg, int a_end,
                                          int np, int maxb, LeanScratch<BS, WITH_VAL> *ls, F &&consume)
{
    constexpr int NW = BS / 64, CAP = 4 * BS, V = VW;
    static_assert(V == 4 && (4 % U) == 0, "four rounds per window");
    using own_t = typename LeanScratch<BS, WITH_VAL>::own_t;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    auto sync = [&]() {
        if (NW == 1) wave_lds_sync(); else __syncthreads();
    };
    // DIRECT rounds: every thread walks the B row of ITS entry, chunk r in round r -- no parking, no owner array, no
    // barrier.  That is ceil(alen / BS) * ceil(maxb / 4) rounds against about chunks / BS + 1.5 per batch for the
    // owner array (its set-up is worth a round and a half): the choice for the rows of power-law inputs whose
    // hundreds of B rows have two or three entries each (np, maxb: products and longest B row of this C row).
    {
        const int alen = a_end - a_beg, nb = (alen + BS - 1) / BS, mbc = (maxb + V - 1) >> 2;
        const int ce = (np >> 2) + (alen >> 1);  // chunks, roughly: a partial one for every other entry
        if (2 * nb * mbc <= 2 * ((ce + BS - 1) / BS) + 3 * nb) {
            // the caller's tables are cleared by all lanes before the first probe: a workgroup barrier, or -- one
            // wavefront -- its program order, said aloud (LDS fence + scheduling barrier; the sync point of tests/emu)
            if (NW > 1) __syncthreads(); else wave_lds_sync();
            for (int b0 = a_beg; b0 < a_end; b0 += BS) {
                int b = 0, e = 0;
                real av = 0;
                const int j = b0 + (int)threadIdx.x;
                if (j < a_end) {
                    const int c = __builtin_nontemporal_load(acol + j);
                    if (WITH_VAL) av = __builtin_nontemporal_load(aval + j);
                    struct __attribute__((aligned(4))) I2 {
                        int b, e;
                    };
                    const I2 r = *reinterpret_cast<const I2 *>(brpt + c);
                    b = r.b;
                    e = r.e;
                }
                for (int r0 = 0; r0 < mbc; r0 += U) {
                    IVecT<V> pk[U];
                    RVecT<WITH_VAL ? V : 1> pv[U];
                    int pn[U];
#pragma unroll
                    for (int u = 0; u < U; u++) pn[u] = fetch_chunk<WITH_VAL, V>(bcol, bval, b + (r0 + u) * V, e, bnnz, pk[u], pv[u]);
#pragma unroll
                    for (int u = 0; u < U; u++)
                        if (pn[u] > 0) consume(pk[u], pv[u], pn[u], av);
                }
            }
            sync();
            return;
        }
    }
    for (int b0 = a_beg; b0 < a_end; b0 += BS) {
        int b = 0, e = 0;
        real av = 0;
        {
            const int j = b0 + (int)threadIdx.x;
            if (j < a_end) {
                const int c = __builtin_nontemporal_load(acol + j);
                if (WITH_VAL) av = __builtin_nontemporal_load(aval + j);
                struct __attribute__((aligned(4))) I2 {
                    int b, e;
                };
                const I2 r = *reinterpret_cast<const I2 *>(brpt + c);  // one 8-byte gather
                b = r.b;
                e = r.e;
            }
        }
        const int nch = (e - b + V - 1) >> 2;
        const int incl = wave_incl_scan(nch);
        int cp, total;
        if (NW == 1) {
            cp = incl - nch;
            total = __builtin_amdgcn_readlane(incl, 63);
        } else {
            if (lane == 63) ls->wtot[wv] = incl;
            __syncthreads();
            int base = 0;
            total = 0;
#pragma unroll
            for (int u = 0; u < NW; u++) {
                const int c = ls->wtot[u];
                base += u < wv ? c : 0;
                total += c;
            }
            cp = base + incl - nch;
        }
        ls->ent[threadIdx.x] = make_int2(b - V * cp, e);
        if (WITH_VAL) ls->av[threadIdx.x] = av;
        // the owner array is cleared once per batch: inside a batch the entry numbers only grow with the chunk
        // number, so what an earlier window left behind never wins the max-scan
        *reinterpret_cast<typename std::conditional<(sizeof(own_t) == 1), unsigned int, uint2>::type *>(
            ls->own + 4 * threadIdx.x) = {};
        for (int w0 = 0; w0 < total; w0 += CAP) {
            sync();  // entries parked / owner array cleared (first window); rounds of the previous window done
            // my entry owns the chunk where it starts -- or chunk 0 of the window when it straddles its start
            {
                const int rel = cp - w0;
                if (nch > 0 && rel < CAP && rel + nch > 0) ls->own[rel > 0 ? rel : 0] = (own_t)(threadIdx.x + 1);
            }
            sync();
            {
                // inclusive max-scan of the array, four slots per thread
                int o[4];
                if (sizeof(own_t) == 1) {
                    const unsigned int w = *reinterpret_cast<const unsigned int *>(ls->own + 4 * threadIdx.x);
                    o[0] = w & 0xff, o[1] = (w >> 8) & 0xff, o[2] = (w >> 16) & 0xff, o[3] = w >> 24;
                } else {
                    const uint2 w = *reinterpret_cast<const uint2 *>(ls->own + 4 * threadIdx.x);
                    o[0] = w.x & 0xffff, o[1] = w.x >> 16, o[2] = w.y & 0xffff, o[3] = w.y >> 16;
                }
                o[1] = o[1] > o[0] ? o[1] : o[0];
                o[2] = o[2] > o[1] ? o[2] : o[1];
                o[3] = o[3] > o[2] ? o[3] : o[2];
                const int inc = wave_incl_max(o[3]);
                int ex = __builtin_amdgcn_update_dpp(0, inc, 0x138, 0xf, 0xf, false);  // wave_shr:1: the lanes before me
                if (NW > 1) {
                    if (lane == 63) ls->wmax[wv] = inc;
                    __syncthreads();
#pragma unroll
                    for (int u = 0; u < NW; u++) {
                        const int c = ls->wmax[u];
                        ex = (u < wv && c > ex) ? c : ex;
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; q++) o[q] = o[q] > ex ? o[q] : ex;
                if (sizeof(own_t) == 1) {
                    *reinterpret_cast<unsigned int *>(ls->own + 4 * threadIdx.x) =
                        (unsigned)o[0] | ((unsigned)o[1] << 8) | ((unsigned)o[2] << 16) | ((unsigned)o[3] << 24);
                } else {
                    *reinterpret_cast<uint2 *>(ls->own + 4 * threadIdx.x) =
                        make_uint2((unsigned)o[0] | ((unsigned)o[1] << 16), (unsigned)o[2] | ((unsigned)o[3] << 16));
                }
            }
            sync();
            const int wn = total - w0 < CAP ? total - w0 : CAP;  // chunks of this window
            if constexpr (PIPE) {
            // one chunk in flight BEHIND the one being hashed (round r + 1 is requested before round r is consumed):
            // the memory round trip of a round hides behind the probes of the round before, at 12 registers per buffer
            {
                IVecT<V> pk[2];
                RVecT<WITH_VAL ? V : 1> pv[2];
                int pn[2] = {0, 0};
                real sc[2] = {0, 0};
                auto request = [&](int r) {
                    const int q = r * BS + (int)threadIdx.x, sl = r & 1;
                    pn[sl] = 0;
                    if (q < wn) {
                        const int i = (int)ls->own[q] - 1;
                        const int2 x = ls->ent[i];
                        if (WITH_VAL) sc[sl] = ls->av[i];
                        pn[sl] = fetch_chunk<WITH_VAL, V>(bcol, bval, x.x + (w0 + q) * V, x.y, bnnz, pk[sl], pv[sl]);
                    }
                };
                request(0);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    if (r * BS >= wn) break;  // uniform
                    const int sl = r & 1;
                    const IVecT<V> ck = pk[sl];
                    const RVecT<WITH_VAL ? V : 1> cv = pv[sl];
                    const int cn = pn[sl];
                    const real cs = sc[sl];
                    if (r < 3 && (r + 1) * BS < wn) request(r + 1);
                    if (cn > 0) consume(ck, cv, cn, cs);
                }
            }
            } else {
#pragma unroll
            for (int r0 = 0; r0 < 4; r0 += U) {
                if (r0 * BS >= wn) break;  // uniform
                IVecT<V> pk[U];
                RVecT<WITH_VAL ? V : 1> pv[U];
                int pn[U];
                real sc[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int q = (r0 + u) * BS + (int)threadIdx.x;
                    pn[u] = 0;
                    sc[u] = 0;
                    if (q < wn) {
                        const int i = (int)ls->own[q] - 1;
                        const int2 x = ls->ent[i];
                        if (WITH_VAL) sc[u] = ls->av[i];
                        pn[u] = fetch_chunk<WITH_VAL, V>(bcol, bval, x.x + (w0 + q) * V, x.y, bnnz, pk[u], pv[u]);
                    }
                }
#pragma unroll
                for (int u = 0; u < U; u++)
                    if (pn[u] > 0) consume(pk[u], pv[u], pn[u], sc[u]);
            }
            }
        }
        sync();  // the next batch overwrites the parked entries
    }
}

// Four find-or-inserts, the probes of a round issued back to back and NO branch inside a round: a key that is done
// re-probes the slot it owns (CAS(slot, -1, key) on a slot that holds key writes nothing and returns key).  Elements
// n .. 3 of a partial chunk are copies of element 0.  fresh += keys this call inserted.
// BF (experiments): branch-free retry rounds over all four keys instead of a block per key.
template <bool BF>
__device__ __forceinline__ void lean_insert4(int *tab, int mask, int shift, int bits, const IVec &k, int n, int (&h)[VW], int &fresh)
{
    int kk[VW], old[VW];
    kk[0] = k.v[0];
#pragma unroll
    for (int i = 1; i < VW; i++) kk[i] = i < n ? k.v[i] : kk[0];
#pragma unroll
    for (int i = 0; i < VW; i++) h[i] = lean_slot(kk[i], shift, bits);
#pragma unroll
    for (int i = 0; i < VW; i++) old[i] = atomicCAS(tab + h[i], -1, kk[i]);
    bool pend[VW], any = false;
    NSP_COUNT(FC_HASH_LEAN, 0, n);
    NSP_COUNT(FC_HASH_LEAN, 1, VW);  // (elements n .. 3 of a partial chunk are copies of element 0: their CAS is issued too)
#pragma unroll
    for (int i = 0; i < VW; i++) {
        fresh += old[i] == -1;
        pend[i] = old[i] != -1 && old[i] != kk[i];
        any |= pend[i];
        if (pend[i] && i < n) NSP_COUNT(FC_HASH_LEAN, 2, 1);
    }
    // (the retries: a block per key that only the lanes still probing that key enter -- five vector instructions per
    //  executed block.  A branch-free retry round over all four keys was measured: 36 VALU per round whether one lane
    //  retries or all, stencil numeric 1.81 -> 2.47 ms at load factor 1/2.)
    if constexpr (BF) {
    while (any) {  // branch-free round: a key that is done re-probes the slot it owns (a no-op)
#pragma unroll
        for (int i = 0; i < VW; i++) h[i] = (h[i] + (pend[i] ? 1 : 0)) & mask;
#pragma unroll
        for (int i = 0; i < VW; i++) old[i] = atomicCAS(tab + h[i], -1, kk[i]);
        NSP_COUNT(FC_HASH_LEAN, 1, VW);
        any = false;
#pragma unroll
        for (int i = 0; i < VW; i++) {
            fresh += old[i] == -1;
            pend[i] = old[i] != -1 && old[i] != kk[i];
            any |= pend[i];
        }
    }
    } else {
    while (any) {
        any = false;
#pragma unroll
        for (int i = 0; i < VW; i++) {
            if (pend[i]) {
                h[i] = (h[i] + 1) & mask;
                NSP_COUNT(FC_HASH_LEAN, 1, 1);
                const int o = atomicCAS(tab + h[i], -1, kk[i]);
                fresh += o == -1;
                pend[i] = o != -1 && o != kk[i];
                any |= pend[i];
            }
        }
    }
    }
}

// ---- one-wavefront sort of up to 128 distinct keys in registers ------------------------------------------------------
// Element e of the sequence lives in lane e & 63, register e >> 6.  Flip form of the bitonic network: merge size k
// starts with the comparator (i, i ^ (k - 1)) and continues with (i, i ^ j), j = k/4 ... 1; the lower index always
// keeps the minimum.
// Compare-exchange against the partner a DPP pattern names: min and max with the pattern folded into the instruction
// (the compiler turns update_dpp + min + max into copy, v_mov_dpp, v_min, v_max, v_cndmask: five instructions; here
// three).  s_nop 1: a DPP operand written by the VALU instruction before needs two wait states on gfx9.
#ifndef NSP_EMU
#define NSP_CEX_DPP(x, lower, CTRL)                                                                          \
    do {                                                                                                     \
        int lo_, hi_;                                                                                        \
        asm("s_nop 1\n\tv_min_i32_dpp %0, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                   \
            "v_max_i32_dpp %1, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf"                                   \
            : "=&v"(lo_), "=&v"(hi_)                                                                         \
            : "v"(x));                                                                                       \
        x = (lower) ? lo_ : hi_;                                                                             \
    } while (0)
// ... against a partner value q fetched through the LDS crossbar (ds_swizzle / ds_bpermute), where the lanes that keep
// the minimum are whole banks (4 lanes) or rows (16 lanes): the identity DPP pattern with a bank / row mask is a
// predicated write -- two instructions, no select, no mask register.
#define NSP_CEX_MASKED(x, q, LO, HI)                                                                         \
    asm("v_min_i32_dpp %0, %1, %0 quad_perm:[0,1,2,3] " LO "\n\tv_max_i32_dpp %0, %1, %0 quad_perm:[0,1,2,3] " HI \
        : "+v"(x)                                                                                            \
        : "v"(q))
// ... of x against the DPP partner of a SECOND register y (the eight-per-lane LDS sort below)
#define NSP_CEX_DPP2(out, x, y, lower, CTRL)                                                                 \
    do {                                                                                                     \
        int lo_, hi_;                                                                                        \
        asm("s_nop 1\n\tv_min_i32_dpp %0, %3, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                   \
            "v_max_i32_dpp %1, %3, %2 " CTRL " row_mask:0xf bank_mask:0xf"                                   \
            : "=&v"(lo_), "=&v"(hi_)                                                                         \
            : "v"(x), "v"(y));                                                                               \
        out = (lower) ? lo_ : hi_;                                                                           \
    } while (0)
#else
// tests/emu (lane-by-lane CPU emulation of the kernels, test infrastructure): the SAME operand strings, interpreted
// with the instruction's rules (a lane the masks disable or whose DPP source is invalid is not written: 0x7fffdead
// shows up in the result if the sort ever relied on one)
#define NSP_CEX_DPP(x, lower, CTRL)                                                                          \
    do {                                                                                                     \
        const int lo_ = emu_dpp_asm(false, 0x7fffdead, x, x, CTRL " row_mask:0xf bank_mask:0xf");             \
        const int hi_ = emu_dpp_asm(true, 0x7fffdead, x, x, CTRL " row_mask:0xf bank_mask:0xf");              \
        x = (lower) ? lo_ : hi_;                                                                             \
    } while (0)
#define NSP_CEX_MASKED(x, q, LO, HI)                                                                         \
    do {                                                                                                     \
        x = emu_dpp_asm(false, x, q, x, "quad_perm:[0,1,2,3] " LO);                                           \
        x = emu_dpp_asm(true, x, q, x, "quad_perm:[0,1,2,3] " HI);                                            \
    } while (0)
#define NSP_CEX_DPP2(out, x, y, lower, CTRL)                                                                 \
    do {                                                                                                     \
        const int lo_ = emu_dpp_asm(false, 0x7fffdead, y, x, CTRL " row_mask:0xf bank_mask:0xf");             \
        const int hi_ = emu_dpp_asm(true, 0x7fffdead, y, x, CTRL " row_mask:0xf bank_mask:0xf");              \
        out = (lower) ? lo_ : hi_;                                                                           \
    } while (0)
#endif

// P = pow2 >= 2 elements; r1 is touched only when P == 128
template <int P>
__device__ __forceinline__ void wave_sort_regs(int &r0, int &r1, int lane)
{
    const bool b0 = (lane & 1) == 0, b1 = (lane & 2) == 0, b2 = (lane & 4) == 0, b3 = (lane & 8) == 0;
    constexpr bool TWO = P == 128;
    const int a32 = (lane ^ 32) << 2, amir = (63 - lane) << 2;
    int x0 = r0, x1 = r1;  // (locals, not the references: the asm operands must be registers, not memory)
#define NSP_BOTH(STMT0, STMT1) do { STMT0; if (TWO) { STMT1; } } while (0)
#define NSP_X1() NSP_BOTH(NSP_CEX_DPP(x0, b0, "quad_perm:[1,0,3,2]"), NSP_CEX_DPP(x1, b0, "quad_perm:[1,0,3,2]"))
#define NSP_X2() NSP_BOTH(NSP_CEX_DPP(x0, b1, "quad_perm:[2,3,0,1]"), NSP_CEX_DPP(x1, b1, "quad_perm:[2,3,0,1]"))
#define NSP_X8() NSP_BOTH(NSP_CEX_DPP(x0, b3, "row_ror:8"), NSP_CEX_DPP(x1, b3, "row_ror:8"))
    // partner through the LDS crossbar, minimum kept by whole banks / rows (LO / HI: the DPP masks of the two halves)
#define NSP_XQ(FETCH0, FETCH1, LO, HI)                                                                       \
    do {                                                                                                     \
        const int q0_ = FETCH0, q1_ = TWO ? FETCH1 : 0;                                                      \
        NSP_BOTH(NSP_CEX_MASKED(x0, q0_, LO, HI), NSP_CEX_MASKED(x1, q1_, LO, HI));                          \
    } while (0)
#define NSP_X4() NSP_XQ(__builtin_amdgcn_ds_swizzle(x0, (4 << 10) | 0x1f), __builtin_amdgcn_ds_swizzle(x1, (4 << 10) | 0x1f), \
                        "row_mask:0xf bank_mask:0x5", "row_mask:0xf bank_mask:0xa")  /* bit 2 clear = banks 0, 2 */
#define NSP_X16() NSP_XQ(__builtin_amdgcn_ds_swizzle(x0, (16 << 10) | 0x1f), __builtin_amdgcn_ds_swizzle(x1, (16 << 10) | 0x1f), \
                         "row_mask:0x5 bank_mask:0xf", "row_mask:0xa bank_mask:0xf")  /* bit 4 clear = rows 0, 2 */
#define NSP_X32() NSP_XQ(__builtin_amdgcn_ds_bpermute(a32, x0), __builtin_amdgcn_ds_bpermute(a32, x1),      \
                         "row_mask:0x3 bank_mask:0xf", "row_mask:0xc bank_mask:0xf")  /* bit 5 clear = rows 0, 1 */
    // k = 2
    NSP_X1();
    if (P >= 4) {  // flip 4: quad_perm [3,2,1,0]
        NSP_BOTH(NSP_CEX_DPP(x0, b1, "quad_perm:[3,2,1,0]"), NSP_CEX_DPP(x1, b1, "quad_perm:[3,2,1,0]"));
        NSP_X1();
    }
    if (P >= 8) {  // flip 8: row_half_mirror
        NSP_BOTH(NSP_CEX_DPP(x0, b2, "row_half_mirror"), NSP_CEX_DPP(x1, b2, "row_half_mirror"));
        NSP_X2();
        NSP_X1();
    }
    if (P >= 16) {  // flip 16: row_mirror
        NSP_BOTH(NSP_CEX_DPP(x0, b3, "row_mirror"), NSP_CEX_DPP(x1, b3, "row_mirror"));
        NSP_X4();
        NSP_X2();
        NSP_X1();
    }
    if (P >= 32) {  // flip 32: lane ^ 31 (swizzle works inside groups of 32 lanes); bit 4 clear keeps the minimum
        NSP_XQ(__builtin_amdgcn_ds_swizzle(x0, (0x1f << 10) | 0x1f), __builtin_amdgcn_ds_swizzle(x1, (0x1f << 10) | 0x1f),
               "row_mask:0x5 bank_mask:0xf", "row_mask:0xa bank_mask:0xf");
        NSP_X8();
        NSP_X4();
        NSP_X2();
        NSP_X1();
    }
    if (P >= 64) {  // flip 64: lane 63 - l; bit 5 clear keeps the minimum
        NSP_XQ(__builtin_amdgcn_ds_bpermute(amir, x0), __builtin_amdgcn_ds_bpermute(amir, x1),
               "row_mask:0x3 bank_mask:0xf", "row_mask:0xc bank_mask:0xf");
        NSP_X16();
        NSP_X8();
        NSP_X4();
        NSP_X2();
        NSP_X1();
    }
    if (TWO) {  // flip 128: element e against 127 - e = register 1 of lane 63 - l
        const int m0 = __builtin_amdgcn_ds_bpermute(amir, x1), m1 = __builtin_amdgcn_ds_bpermute(amir, x0);
        x0 = x0 < m0 ? x0 : m0;
        x1 = x1 > m1 ? x1 : m1;
        NSP_X32();
        NSP_X16();
        NSP_X8();
        NSP_X4();
        NSP_X2();
        NSP_X1();
    }
    r0 = x0;
    if (TWO) r1 = x1;
#undef NSP_X1
#undef NSP_X2
#undef NSP_X4
#undef NSP_X8
#undef NSP_X16
#undef NSP_X32
#undef NSP_XQ
#undef NSP_BOTH
}

__device__ __forceinline__ void wave_sort128(int &r0, int &r1, int P, int lane)
{
    switch (P) {
        case 128: wave_sort_regs<128>(r0, r1, lane); break;
        case 64: wave_sort_regs<64>(r0, r1, lane); break;
        case 32: wave_sort_regs<32>(r0, r1, lane); break;
        case 16: wave_sort_regs<16>(r0, r1, lane); break;
        case 8: wave_sort_regs<8>(r0, r1, lane); break;
        case 4: wave_sort_regs<4>(r0, r1, lane); break;
        case 2: wave_sort_regs<2>(r0, r1, lane); break;
        default: break;
    }
}

// ---- workgroup sort of 1024 .. 8192 keys in LDS, flip form (big-table bins) -------------------------------------------
// bitonic_sort_lds (common.h) is the direction form of the network: every compare-exchange of its register stages is a
// ds_swizzle, a direction bit, min, max and two selects.  The flip form lets the register stages use the folded DPP /
// masked-write compare-exchanges of the one-wavefront sort above: element e = segment * 512 + lane * 8 + i, eight per
// lane; distances 1 / 2 / 4 stay inside the thread (min + max per pair), 8 .. 256 are lane distances 1 .. 32, and a flip
// of merge size k <= 512 pairs (lane, i) with (mirror of the lane inside k/8 lanes, 7 - i).  Merge sizes beyond 512 do
// their flip and their distances >= 512 through LDS with a workgroup barrier each, then the distances 256 .. 1 in
// registers again.  (numpy model of exactly this decomposition: DESIGN 4.1.)  The round-3 profile of the 8192-slot
// numeric bin had the sort at 16 of the 41 us of a row.
template <int BS>
__device__ __forceinline__ void flip_sort_lds(int *s, int P)
{
    constexpr int NW = BS / 64;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const bool b0 = (lane & 1) == 0, b1 = (lane & 2) == 0, b2 = (lane & 4) == 0, b3 = (lane & 8) == 0;
    const int a32 = (lane ^ 32) << 2, amir = (63 - lane) << 2;
    int r[8];
#define NSP_LOAD8(eb)                                                                                        \
    do {                                                                                                     \
        const int4 a_ = *reinterpret_cast<const int4 *>(s + (eb)), b_ = *reinterpret_cast<const int4 *>(s + (eb) + 4); \
        r[0] = a_.x, r[1] = a_.y, r[2] = a_.z, r[3] = a_.w, r[4] = b_.x, r[5] = b_.y, r[6] = b_.z, r[7] = b_.w; \
    } while (0)
#define NSP_STORE8(eb)                                                                                       \
    do {                                                                                                     \
        *reinterpret_cast<int4 *>(s + (eb)) = make_int4(r[0], r[1], r[2], r[3]);                             \
        *reinterpret_cast<int4 *>(s + (eb) + 4) = make_int4(r[4], r[5], r[6], r[7]);                         \
    } while (0)
    // inside the thread: pair (i, p), i < p
#define NSP_PAIR(i, p)                                                                                       \
    do {                                                                                                     \
        const int lo_ = r[i] < r[p] ? r[i] : r[p], hi_ = r[i] < r[p] ? r[p] : r[i];                          \
        r[i] = lo_;                                                                                          \
        r[p] = hi_;                                                                                          \
    } while (0)
#define NSP_TX1() do { NSP_PAIR(0, 1); NSP_PAIR(2, 3); NSP_PAIR(4, 5); NSP_PAIR(6, 7); } while (0)
#define NSP_TX2() do { NSP_PAIR(0, 2); NSP_PAIR(1, 3); NSP_PAIR(4, 6); NSP_PAIR(5, 7); } while (0)
#define NSP_TX4() do { NSP_PAIR(0, 4); NSP_PAIR(1, 5); NSP_PAIR(2, 6); NSP_PAIR(3, 7); } while (0)
#define NSP_TF4() do { NSP_PAIR(0, 3); NSP_PAIR(1, 2); NSP_PAIR(4, 7); NSP_PAIR(5, 6); } while (0)
#define NSP_TF8() do { NSP_PAIR(0, 7); NSP_PAIR(1, 6); NSP_PAIR(2, 5); NSP_PAIR(3, 4); } while (0)
    // lane distance L by a DPP pattern: every element against the same element of the partner lane
#define NSP_LX_DPP(lower, CTRL)                                                                              \
    do {                                                                                                     \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; i_++) NSP_CEX_DPP(r[i_], lower, CTRL);                    \
    } while (0)
    // ... through the LDS crossbar, minimum kept by whole banks / rows
#define NSP_LX_Q(FETCH, LO, HI)                                                                              \
    do {                                                                                                     \
        int q_[8];                                                                                           \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; i_++) q_[i_] = FETCH(r[i_]);                              \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; i_++) NSP_CEX_MASKED(r[i_], q_[i_], LO, HI);              \
    } while (0)
#define NSP_F_SWZ4(v) __builtin_amdgcn_ds_swizzle(v, (4 << 10) | 0x1f)
#define NSP_F_SWZ16(v) __builtin_amdgcn_ds_swizzle(v, (16 << 10) | 0x1f)
#define NSP_F_SWZ31(v) __builtin_amdgcn_ds_swizzle(v, (0x1f << 10) | 0x1f)
#define NSP_F_X32(v) __builtin_amdgcn_ds_bpermute(a32, v)
#define NSP_F_MIR(v) __builtin_amdgcn_ds_bpermute(amir, v)
#define NSP_LX1() NSP_LX_DPP(b0, "quad_perm:[1,0,3,2]")
#define NSP_LX2() NSP_LX_DPP(b1, "quad_perm:[2,3,0,1]")
#define NSP_LX4() NSP_LX_Q(NSP_F_SWZ4, "row_mask:0xf bank_mask:0x5", "row_mask:0xf bank_mask:0xa")
#define NSP_LX8() NSP_LX_DPP(b3, "row_ror:8")
#define NSP_LX16() NSP_LX_Q(NSP_F_SWZ16, "row_mask:0x5 bank_mask:0xf", "row_mask:0xa bank_mask:0xf")
#define NSP_LX32() NSP_LX_Q(NSP_F_X32, "row_mask:0x3 bank_mask:0xf", "row_mask:0xc bank_mask:0xf")
    // flip over a mirror group of G lanes: element i against element 7 - i of the mirrored lane
#define NSP_LF_DPP(lower, CTRL)                                                                              \
    do {                                                                                                     \
        int t_[8];                                                                                           \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; i_++) NSP_CEX_DPP2(t_[i_], r[i_], r[7 - i_], lower, CTRL); \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; i_++) r[i_] = t_[i_];                                     \
    } while (0)
#define NSP_LF_Q(FETCH, LO, HI)                                                                              \
    do {                                                                                                     \
        int q_[8];                                                                                           \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; i_++) q_[i_] = FETCH(r[7 - i_]);                          \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; i_++) NSP_CEX_MASKED(r[i_], q_[i_], LO, HI);              \
    } while (0)
    // the distances 4 .. 1 inside the thread, then nothing left
#define NSP_TAIL_T() do { NSP_TX4(); NSP_TX2(); NSP_TX1(); } while (0)
    for (int seg = wid; seg * 512 < P; seg += NW) {
        const int eb = seg * 512 + lane * 8;
        NSP_LOAD8(eb);
        NSP_TX1();                                                            // k = 2
        NSP_TF4(); NSP_TX1();                                                 // k = 4
        NSP_TF8(); NSP_TX2(); NSP_TX1();                                      // k = 8
        NSP_LF_DPP(b0, "quad_perm:[1,0,3,2]"); NSP_TAIL_T();                  // k = 16: 2 lanes
        NSP_LF_DPP(b1, "quad_perm:[3,2,1,0]"); NSP_LX1(); NSP_TAIL_T();       // k = 32: 4 lanes
        NSP_LF_DPP(b2, "row_half_mirror"); NSP_LX2(); NSP_LX1(); NSP_TAIL_T();  // k = 64: 8 lanes
        NSP_LF_DPP(b3, "row_mirror"); NSP_LX4(); NSP_LX2(); NSP_LX1(); NSP_TAIL_T();  // k = 128: 16 lanes
        NSP_LF_Q(NSP_F_SWZ31, "row_mask:0x5 bank_mask:0xf", "row_mask:0xa bank_mask:0xf");  // k = 256: 32 lanes
        NSP_LX8(); NSP_LX4(); NSP_LX2(); NSP_LX1(); NSP_TAIL_T();
        NSP_LF_Q(NSP_F_MIR, "row_mask:0x3 bank_mask:0xf", "row_mask:0xc bank_mask:0xf");    // k = 512: 64 lanes
        NSP_LX16(); NSP_LX8(); NSP_LX4(); NSP_LX2(); NSP_LX1(); NSP_TAIL_T();
        NSP_STORE8(eb);
    }
    __syncthreads();
    for (int k = 1024; k <= P; k <<= 1) {
        {   // flip through LDS: t-th pair = (lo, lo ^ (k - 1)), lo = t with a zero inserted at bit log2(k / 2)
            const int hb = (k >> 1) - 1;
            for (int t = threadIdx.x; t < P / 2; t += BS) {
                const int lo = ((t & ~hb) << 1) | (t & hb), hi = lo ^ (k - 1);
                const int a = s[lo], b = s[hi];
                if (a > b) { s[lo] = b; s[hi] = a; }
            }
            __syncthreads();
        }
        for (int j = k >> 2; j >= 512; j >>= 1) {
            for (int t = threadIdx.x; t < P / 2; t += BS) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                const int a = s[lo], b = s[hi];
                if (a > b) { s[lo] = b; s[hi] = a; }
            }
            __syncthreads();
        }
        for (int seg = wid; seg * 512 < P; seg += NW) {
            const int eb = seg * 512 + lane * 8;
            NSP_LOAD8(eb);
            NSP_LX32(); NSP_LX16(); NSP_LX8(); NSP_LX4(); NSP_LX2(); NSP_LX1(); NSP_TAIL_T();
            NSP_STORE8(eb);
        }
        __syncthreads();
    }
#undef NSP_LOAD8
#undef NSP_STORE8
#undef NSP_PAIR
#undef NSP_TX1
#undef NSP_TX2
#undef NSP_TX4
#undef NSP_TF4
#undef NSP_TF8
#undef NSP_LX_DPP
#undef NSP_LX_Q
#undef NSP_F_SWZ4
#undef NSP_F_SWZ16
#undef NSP_F_SWZ31
#undef NSP_F_X32
#undef NSP_F_MIR
#undef NSP_LX1
#undef NSP_LX2
#undef NSP_LX4
#undef NSP_LX8
#undef NSP_LX16
#undef NSP_LX32
#undef NSP_LF_DPP
#undef NSP_LF_Q
#undef NSP_TAIL_T
}

// ---- symbolic, bins 1..4 (set_row_nz_bin_each_tb :399-472) -----------------------------------------------------------
// LIST: big-table bins may leave the sorted column list of a heavy row for the ranked numeric kernel (symbolic.h).
// FORM (experiments build; 0 in every launch of the default path): bit 0 = branch-free retry rounds, bit 1 = pipelined walk
template <int BS, int TMAX, int U, int FORM = 0>
__global__ __launch_bounds__(BS) void k_sym_lean(const int *__restrict__ arpt, const int *__restrict__ acol,
                                                 const int *__restrict__ brpt, const int *__restrict__ bcol,
                                                 const int *__restrict__ row_perm, const int *__restrict__ row_prod,
                                                 const int *__restrict__ row_maxb,
                                                 int *__restrict__ row_nz, int bin_off, int bin_size, int bnnz,
                                                 BinState *bs, int *__restrict__ tcol, long long *__restrict__ list_off,
                                                 const int *__restrict__ row_span, int dens, int tiled_w)
{
    __shared__ __attribute__((aligned(16))) int tab[TMAX];
    __shared__ __attribute__((aligned(16))) LeanScratch<BS, false> s_ls;
    __shared__ int s_nz;
    const int slot = xcd_row_slot(bin_size);
    if (slot < 0) return;
    const int rid = row_perm[bin_off + slot];
    const int np = row_prod[rid], mb_row = row_maxb[rid];
    const int a_beg = arpt[rid], a_end = arpt[rid + 1];
    int T = pow2_ceil(np + (np >> 1));  // load factor <= 2/3 where the bin's table allows
    if (T < 64) T = 64;
    if (T > TMAX) T = TMAX;
    const int mask = T - 1, bits = 31 - __builtin_clz((unsigned)T), shift = 24 - bits;
    {
        int4 *t4 = reinterpret_cast<int4 *>(tab);
        const int4 m1 = make_int4(-1, -1, -1, -1);
        for (int i = threadIdx.x; i < T / 4; i += BS) t4[i] = m1;
    }
    if (threadIdx.x == 0) s_nz = 0;
    // (the walk's first barrier also publishes the cleared table)
    int cnt = 0;
    lean_walk<BS, false, U, (FORM & 2) != 0>(acol, (const real *)nullptr, brpt, bcol, (const real *)nullptr, bnnz, a_beg, a_end, np, mb_row, &s_ls,
                            [&](const IVec &k, const RVecT<1> &, int n, real) {
                                int h[VW];
                                lean_insert4<(FORM & 1) != 0>(tab, mask, shift, bits, k, n, h, cnt);
                            });
    cnt = wave_sum(cnt);
    if (BS == 64) {
        if (threadIdx.x == 0) row_nz[rid] = cnt;
    } else {
        if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&s_nz, cnt);
        __syncthreads();
        if (threadIdx.x == 0) row_nz[rid] = s_nz;
    }
    if constexpr (TMAX >= 8192) {
        const int nz = s_nz;
        bool want = tcol != nullptr && nz > kListMinNnz;
        if (want) {
            const int sp = row_span[rid];
            want = !(dens > 0 && (long long)nz * dens >= sp && sp <= 32 * tiled_w);  // k_num_tiled's rows need no list
        }
        if (want) {
            constexpr int SPT = TMAX / BS, NWV = BS / 64;
            __shared__ int s_ws[NWV];
            __shared__ long long s_off;
            int keys[SPT], mine = 0;
#pragma unroll
            for (int j = 0; j < SPT; j++) {
                const int i = (int)threadIdx.x * SPT + j;
                keys[j] = i < T ? tab[i] : -1;
                mine += keys[j] != -1;
            }
            const int incl = wave_incl_scan(mine);
            if ((threadIdx.x & 63) == 63) s_ws[threadIdx.x >> 6] = incl;
            if (threadIdx.x == 0) {
                s_off = (long long)atomicAdd(&bs->list_cursor, (unsigned long long)nz);
                list_off[rid] = s_off;
            }
            __syncthreads();  // every slot has been read: the table may be overwritten
            int at = incl - mine;
#pragma unroll
            for (int u = 0; u < NWV; u++) at += u < (int)(threadIdx.x >> 6) ? s_ws[u] : 0;
#pragma unroll
            for (int j = 0; j < SPT; j++)
                if (keys[j] != -1) tab[at++] = keys[j];
            const int P = pow2_ceil(nz);  // <= T: the table was sized for the products
            for (int i = nz + threadIdx.x; i < P; i += BS) tab[i] = 0x7fffffff;
            __syncthreads();
            if (P >= 1024) flip_sort_lds<BS>(tab, P);
            else bitonic_sort_lds<BS>(tab, P);
            int *dst = tcol + s_off;
            for (int i = threadIdx.x; i < nz; i += BS) dst[i] = tab[i];
        }
    }
}

// ---- numeric, bins 1..4 (calculate_value_col_bin_each_tb :829-927) ---------------------------------------------------
// write_col: bit 0 write C.col (0: numeric-only re-run), bit 1 unsorted output.
template <int BS, int TMAX, int U, int FORM = 0>
__global__ __launch_bounds__(BS) void k_num_lean(const int *__restrict__ arpt, const int *__restrict__ acol,
                                                 const real *__restrict__ aval, const int *__restrict__ brpt,
                                                 const int *__restrict__ bcol, const real *__restrict__ bval,
                                                 const int *__restrict__ crpt, int *__restrict__ ccol,
                                                 real *__restrict__ cval, const int *__restrict__ row_perm,
                                                 const int *__restrict__ row_prod, const int *__restrict__ row_maxb,
                                                 int bin_off, int bin_size, int bnnz, int write_col)
{
    __shared__ __attribute__((aligned(16))) acc_t vals[TMAX];
    __shared__ __attribute__((aligned(16))) int keys[TMAX];
    // the scratch of the walk and the sort buffer are never alive together
    union Overlay {
        LeanScratch<BS, true> w;
        int srt[TMAX];
    };
    __shared__ __attribute__((aligned(16))) Overlay s_ov;
    __shared__ int s_cnt;
    int *srt = s_ov.srt;
    const int slot = xcd_row_slot(bin_size);
    if (slot < 0) return;
    const int rid = row_perm[bin_off + slot];
    const int off = crpt[rid];
    const int n = crpt[rid + 1] - off;
    const int a_beg = arpt[rid], a_end = arpt[rid + 1];
    const int np_row = row_prod[rid], mb_row = row_maxb[rid];
    int T = pow2_ceil(n + (n >> 1));
    if (T < 64) T = 64;
    if (T > TMAX) T = TMAX;
    const int mask = T - 1, bits = 31 - __builtin_clz((unsigned)T), shift = 24 - bits;
    {
        int4 *k4 = reinterpret_cast<int4 *>(keys);
        double2 *v2 = reinterpret_cast<double2 *>(vals);
        const int4 m1 = make_int4(-1, -1, -1, -1);
        for (int i = threadIdx.x; i < T / 4; i += BS) k4[i] = m1;
        for (int i = threadIdx.x; i < T / 2; i += BS) v2[i] = make_double2(0.0, 0.0);
    }
    if (threadIdx.x == 0) s_cnt = 0;
    // (the walk's first barrier also publishes the cleared tables)
    lean_walk<BS, true, U, (FORM & 2) != 0>(acol, aval, brpt, bcol, bval, bnnz, a_beg, a_end, np_row, mb_row, &s_ov.w,
                           [&](const IVec &k, const RVec &v, int m, real sc) {
                               int h[VW], fresh = 0;
                               lean_insert4<(FORM & 1) != 0>(keys, mask, shift, bits, k, m, h, fresh);
#pragma unroll
                               for (int i = 0; i < VW; i++)
                                   if (i < m) unsafeAtomicAdd(vals + h[i], (acc_t)(sc * v.v[i]));
                           });
    // (the walk ends with a barrier)
    const int lane = threadIdx.x & 63;
    const int P = pow2_ceil(n);
    if (BS == 64 && P <= 128) {
        // one wavefront: compaction with a register scan, sort in registers, read-out from registers
        int filled = 0;
        for (int base = 0; base < T; base += 256) {
            const int4 kq = base + 4 * lane < T ? *reinterpret_cast<const int4 *>(keys + base + 4 * lane) : make_int4(-1, -1, -1, -1);
            const int c = (kq.x != -1) + (kq.y != -1) + (kq.z != -1) + (kq.w != -1);
            const int incl = wave_incl_scan(c);
            int at = filled + incl - c;
            if (kq.x != -1) srt[at++] = kq.x;
            if (kq.y != -1) srt[at++] = kq.y;
            if (kq.z != -1) srt[at++] = kq.z;
            if (kq.w != -1) srt[at++] = kq.w;
            filled += __builtin_amdgcn_readlane(incl, 63);
        }
        wave_lds_sync();
        int r0 = lane < n ? srt[lane] : 0x7fffffff, r1 = lane + 64 < n ? srt[lane + 64] : 0x7fffffff;
        if (!(write_col & 2)) wave_sort128(r0, r1, P, lane);
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int i = lane + 64 * q, key = q ? r1 : r0;
            if (i < n) {
                int h = lean_slot(key, shift, bits);
                // (bounded: a key that is not in the table -- a defect elsewhere -- must be a wrong value, not a hung device)
                for (int g = 0; keys[h] != key && g < T; g++) h = (h + 1) & mask;
                if (write_col & 1) ccol[off + i] = key;
                cval[off + i] = (real)vals[h];
            }
        }
        return;
    }
    // compaction: ballot + popcount inside the wave, one LDS atomic per 64 slots
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
    for (int i = n + threadIdx.x; i < P; i += BS) srt[i] = 0x7fffffff;
    __syncthreads();
    if (P > 1 && !(write_col & 2)) {
        if (P >= 1024) flip_sort_lds<BS>(srt, P);  // (flip form: DPP / masked-write register stages)
        else bitonic_sort_lds<BS>(srt, P);
    }
    for (int i = threadIdx.x; i < n; i += BS) {
        const int key = srt[i];
        int h = lean_slot(key, shift, bits);
        for (int g = 0; keys[h] != key && g < T; g++) h = (h + 1) & mask;  // (bounded, as above)
        if (write_col & 1) ccol[off + i] = key;
        cval[off + i] = (real)vals[h];
    }
}

}  // namespace spgemm
}  // namespace nsp
