// ba_schur_body.h — the body of K5 (linearisation + point-block Schur complement on the f64 matrix cores; see
// ba_schur.hip for what it computes and why), shared by the K5 kernels (ba_schur.hip) and by the one-launch-per-round
// kernel (ba_round.hip).
#pragma once
#include "ba_common.h"
#include "ba_backsub_body.h"

#define IT_L 64                 // landmarks per item, upper bound (= per workgroup; 8 per wave); the actual
                                // count g.it_l is 40 when that puts every item on its own CU (P <= 256 * 40)
#define IT_L_SMALL 40
#define SCH_WAVES 8             // waves per workgroup (2 per SIMD: latency hiding; <= 2 SYRK tiles per wave)
#define SCH_SUBS 8              // lanes sharing one landmark (8 landmarks per wave)
#define YT_STRIDE4 81           // doubles per K-column, NT = 4: 64 rows + 17 (17 mod 32 keeps the two
                                // half-wave column groups of a ds_read_b64 on disjoint banks; 3*81*2 = 6 mod 32
                                // spreads the 16 producer lanes over 16 bank pairs)
#define YT_STRIDE8 145          // NT = 8: 128 rows + 17
#define YT_DOUBLES (192 * YT_STRIDE4)  // 15552 doubles = 124416 B per workgroup of 64 landmarks (NT=8: 96 cols * 145 = 13920)
// doubles of the LDS tile of an item of it_l landmarks: 3 it_l columns of the 4x4-tile class, or two half batches of the 8x8 class
__host__ __device__ __forceinline__ int sch_tile_doubles(int it_l)
{
    const int a = 3 * it_l * YT_STRIDE4, b8 = 3 * (it_l / 2) * YT_STRIDE8;
    return ((a > b8 ? a : b8) + 1) & ~1;
}
#ifndef K5_ALLSETS
#define K5_ALLSETS 1            // every speculative radius in one pass over the tile (syrk_tiles_sets); 0: set by set, the tile transformed in place
#endif
#define SCH_PRE 3               // observation rounds prefetched per lane (covers 12 observations per landmark)
#define SCH_MAXC_LDS 64         // cameras staged in LDS when the window has at most this many

typedef __attribute__((ext_vector_type(4))) double d4;

// ------------------------------------------------------------------ helpers
__device__ __forceinline__ int rank_in_mask(uint64_t m0, uint64_t m1, int s)
{
    if (s < 64) return __popcll(m0 & ((1ull << s) - 1ull));
    return __popcll(m0) + __popcll(m1 & ((1ull << (s - 64)) - 1ull));
}

__device__ __forceinline__ int nth_set_bit(uint64_t m0, uint64_t m1, int r)
{
    // index of the r-th (0-based) set bit of the 128-bit mask
    uint64_t m = m0;
    int base = 0;
    const int c0 = __popcll(m0);
    if (r >= c0) { r -= c0; m = m1; base = 64; }
    for (int i = 0; i < r; i++) m &= m - 1;
    return base + __builtin_ctzll(m);
}

// cholesky of the damped point block; Li = L^-1 (lower, row-major 6 entries: 00 10 11 20 21 22)
// (rsqrt_nr: ba_common.h — no f64 sqrt / divide sequences on the chain)
// L[6] (same order as Li) also returns the factor itself: sqrt(x) = x * rsqrt(x)
__device__ __forceinline__ bool chol3_inv(const double V[6], double Li[6], double I[6], double L[6])
{
    const double l00s = V[0];
    if (!(l00s > 0.0)) return false;
    const double i00 = rsqrt_nr(l00s);
    const double l10 = V[1] * i00, l20 = V[2] * i00;
    const double l11s = V[3] - l10 * l10;
    if (!(l11s > 0.0)) return false;
    const double i11 = rsqrt_nr(l11s);
    const double l21 = (V[4] - l20 * l10) * i11;
    const double l22s = V[5] - l20 * l20 - l21 * l21;
    if (!(l22s > 0.0)) return false;
    const double i22 = rsqrt_nr(l22s);
    const double i10 = -l10 * i00 * i11;
    const double i21 = -l21 * i11 * i22;
    const double i20 = -(l20 * i00 + l21 * i10) * i22;
    Li[0] = i00; Li[1] = i10; Li[2] = i11; Li[3] = i20; Li[4] = i21; Li[5] = i22;
    L[0] = l00s * i00; L[1] = l10; L[2] = l11s * i11; L[3] = l20; L[4] = l21; L[5] = l22s * i22;
    I[0] = i00 * i00 + i10 * i10 + i20 * i20;
    I[1] = i10 * i11 + i20 * i21;
    I[2] = i20 * i22;
    I[3] = i11 * i11 + i21 * i21;
    I[4] = i21 * i22;
    I[5] = i22 * i22;
    return isfinite(I[0]) && isfinite(I[3]) && isfinite(I[5]);
}

// Upper-triangle tile list shared by the 4 waves of a workgroup: tile id t -> (r, c), r <= c.
__device__ __forceinline__ void tile_rc(int t, int NT, int& r, int& c)
{
    // row-major enumeration of the upper triangle of an NT x NT tile grid
    r = 0;
    int rem = t, len = NT;
    while (rem >= len) { rem -= len; len--; r++; }
    c = r + rem;
}

// The K loop of NM tiles of one wave, software-pipelined: the operands of chunk kc + 1 are requested before the
// MFMAs of chunk kc are issued (straight-line code per NM, so that the compiler's lgkmcnt waits are exact: with the
// loads inside wave-uniform branches every MFMA waited for its own ds_read — 230 cycles per MFMA instead of 64).
template <int NM, int STRIDE>
__device__ __forceinline__ void syrk_tiles(const double* yt, int nchunks, const int* tr, const int* tc, d4* acc, int lr, int lk)
{
    double a0[NM], b0[NM], a1[NM], b1[NM];
    const double* col = yt + (size_t)lk * STRIDE + lr;
#pragma unroll
    for (int t = 0; t < NM; t++) { a0[t] = col[16 * tr[t]]; b0[t] = col[16 * tc[t]]; }
    for (int kc = 0; kc < nchunks; kc += 2) {
        const double* c1 = col + (size_t)(4 * min(kc + 1, nchunks - 1)) * STRIDE;
#pragma unroll
        for (int t = 0; t < NM; t++) { a1[t] = c1[16 * tr[t]]; b1[t] = c1[16 * tc[t]]; }
#pragma unroll
        for (int t = 0; t < NM; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[t], b0[t], acc[t], 0, 0, 0);
        const double* c2 = col + (size_t)(4 * min(kc + 2, nchunks - 1)) * STRIDE;
#pragma unroll
        for (int t = 0; t < NM; t++) { a0[t] = c2[16 * tr[t]]; b0[t] = c2[16 * tc[t]]; }
        if (kc + 1 < nchunks) {
#pragma unroll
            for (int t = 0; t < NM; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[t], b1[t], acc[t], 0, 0, 0);
        }
    }
}

// SYRK of the compact item on the matrix cores + scatter into S / rhs.  Only the tiles of the upper triangle that
// carry data are enumerated, and they are dealt to SIMDs (wave w runs on SIMD w % 4), not to waves: SIMD 1, 2, 3, 0,
// 1, ... in turn, then round-robin over the waves of the workgroup on that SIMD.  (f64 MFMA throughput is per SIMD:
// with 5 waves, "tile t to wave t % 5" put 4 of 10 tiles on SIMD 0, which holds waves 0 and 4.)  Every wave runs
// over ALL K-chunks of its tiles.
//   yt : LDS tile, column-major [col][STRIDE]; rows [0, 6*ns] used (row 6*ns = rhs row)
template <int NT, int TPW, int STRIDE>
__device__ __forceinline__ void syrk_scatter(const double* yt, int nchunks, int ns, const int* gslot, int n,
                                             int wave, int nw, double* __restrict__ S, double* __restrict__ rhs)
{
    const int lane = threadIdx.x & 63;
    const int lr = lane & 15, lk = lane >> 4;
    const int nrow = 6 * ns;                 // rhs row index
    const int nt_used = (nrow + 16) / 16;    // tile rows that carry data (incl. the rhs row), <= NT
    const int ntiles = nt_used * (nt_used + 1) / 2;
    const int simd = wave & 3, pos = (simd + 3) & 3;
    const int nws = (nw - simd + 3) >> 2;    // waves of this workgroup on my SIMD; I am number wave >> 2 of them
    d4 acc[TPW];
    int tr[TPW], tc[TPW];
    int nmine = 0;
#pragma unroll
    for (int t = 0; t < TPW; t++) {
        acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
        tr[t] = 0; tc[t] = 0;
        const int tile = pos + 4 * ((wave >> 2) + nws * t);
        if (tile < ntiles) { tile_rc(tile, nt_used, tr[t], tc[t]); nmine = t + 1; }
    }
    static_assert(TPW >= 1 && TPW <= 3, "tiles per wave");
    switch (nmine) {                          // wave-uniform
    case 1: syrk_tiles<1, STRIDE>(yt, nchunks, tr, tc, acc, lr, lk); break;
    case 2: if (TPW >= 2) syrk_tiles<(TPW >= 2 ? 2 : 1), STRIDE>(yt, nchunks, tr, tc, acc, lr, lk); break;
    case 3: if (TPW >= 3) syrk_tiles<(TPW >= 3 ? 3 : 1), STRIDE>(yt, nchunks, tr, tc, acc, lr, lk); break;
    default: break;
    }
    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int t = 0; t < TPW; t++) {
        if (t >= nmine) continue;
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int R = 16 * tr[t] + lk + 4 * reg, Cc = 16 * tc[t] + lr;
            const double v = acc[t][reg];
            if (R >= nrow || Cc > nrow || R > Cc) continue;
            const int gr = 6 * gslot[R / 6] + R % 6;
            if (Cc == nrow) atomicAdd(&rhs[gr], -v);
            else atomicAdd(&S[(size_t)gr * n + 6 * gslot[Cc / 6] + Cc % 6], -v);
        }
    }
}

// ---- every speculative set in ONE pass over the tile (round 4).  With V + Lambda_s = L_s L_s^T the set's tile is
// Y_s = Y_0 (L_s^-1 L_0)^T, so  Y_s Y_s^T = Y_0 G_s Y_0^T  with the 3x3 symmetric  G_s = L_0^T (V + Lambda_s)^-1 L_0  per
// landmark: the A operand of every set is Y_0 as it stands, the B operand of set s is Y_0 G_s, formed on the fly from the
// landmark's three tile columns.  One K loop feeds NS accumulator sets: the tile is read once, no transform pass, no
// barrier between the sets, and the scatter's index arithmetic is shared.  Gt[s - 1][landmark][6] (xx xy xz yy yz zz).
// PLAIN: accumulator slot 0 is set 0 (B operand = Y_0 itself) and slots 1 .. NS - 1 go through Gt[0 .. NS - 2]; otherwise all NS
// slots go through Gt[0 .. NS - 1] (the second pass of a round with more than three radii).
template <int NM, int NS, bool PLAIN, int STRIDE>
__device__ __forceinline__ void syrk_tiles_sets(const double* yt, const double* Gt, int it_l, int nchunks, const int* tr, const int* tc,
                                                d4 (*acc)[3], int lr, int lk)
{
    constexpr int NG = PLAIN ? NS - 1 : NS;
    // column 4 kc + lk = 3 p + kk of the tile
    auto load = [&](int kc, double* a, double* bp, double (*yb)[3], double (*g)[3]) {
        const int col = 4 * kc + lk;
        const int p = col / 3, kk = col - 3 * p;
        const double* c0 = yt + (size_t)(3 * p) * STRIDE + lr;
        const double* ca = yt + (size_t)col * STRIDE + lr;
#pragma unroll
        for (int t = 0; t < NM; t++) {
            a[t] = ca[16 * tr[t]];
            if (PLAIN) bp[t] = ca[16 * tc[t]];             // set 0's B operand (one more LDS read instead of a per-lane select)
            yb[t][0] = c0[16 * tc[t]]; yb[t][1] = c0[STRIDE + 16 * tc[t]]; yb[t][2] = c0[2 * STRIDE + 16 * tc[t]];
        }
        // row kk of the symmetric G: (kk, 0) (kk, 1) (kk, 2) in xx xy xz yy yz zz order
        const int i0 = kk, i1 = kk == 0 ? 1 : (kk == 1 ? 3 : 4), i2 = kk == 0 ? 2 : (kk == 1 ? 4 : 5);
#pragma unroll
        for (int s = 0; s < NG; s++) {
            const double* gp = Gt + ((size_t)s * it_l + p) * 6;
            g[s][0] = gp[i0]; g[s][1] = gp[i1]; g[s][2] = gp[i2];
        }
    };
    auto mma = [&](const double* a, const double* bp, const double (*yb)[3], const double (*g)[3]) {
#pragma unroll
        for (int t = 0; t < NM; t++) {
            if (PLAIN) acc[t][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], bp[t], acc[t][0], 0, 0, 0);
#pragma unroll
            for (int s = 0; s < NG; s++) {
                const double bs = fma(yb[t][2], g[s][2], fma(yb[t][1], g[s][1], yb[t][0] * g[s][0]));
                acc[t][s + (PLAIN ? 1 : 0)] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], bs, acc[t][s + (PLAIN ? 1 : 0)], 0, 0, 0);
            }
        }
    };
    double a0[NM], b0[NM], yb0[NM][3], g0[3][3], a1[NM], b1[NM], yb1[NM][3], g1[3][3];
    load(0, a0, b0, yb0, g0);
    for (int kc = 0; kc < nchunks; kc += 2) {
        load(min(kc + 1, nchunks - 1), a1, b1, yb1, g1);
        mma(a0, b0, yb0, g0);
        load(min(kc + 2, nchunks - 1), a0, b0, yb0, g0);
        if (kc + 1 < nchunks) mma(a1, b1, yb1, g1);
    }
}

template <int NS, bool PLAIN, int STRIDE>
__device__ __forceinline__ void syrk_scatter_sets(const double* yt, const double* Gt, int it_l, int nchunks, int ns, const int* gslot, int n,
                                                  int wave, int nw, double* __restrict__ S0, double* __restrict__ rhs0)
{
    const int lane = threadIdx.x & 63;
    const int lr = lane & 15, lk = lane >> 4;
    const int nrow = 6 * ns;                 // rhs row index
    const int nt_used = (nrow + 16) / 16;    // tile rows that carry data (incl. the rhs row), <= 4
    const int ntiles = nt_used * (nt_used + 1) / 2;
    const int simd = wave & 3, pos = (simd + 3) & 3;
    const int nws = (nw - simd + 3) >> 2;    // waves of this workgroup on my SIMD; I am number wave >> 2 of them
    d4 acc[3][3];                            // [tile][set]
    int tr[3], tc[3];
    int nmine = 0;
#pragma unroll
    for (int t = 0; t < 3; t++) {
#pragma unroll
        for (int s = 0; s < 3; s++) acc[t][s] = (d4){0.0, 0.0, 0.0, 0.0};
        tr[t] = 0; tc[t] = 0;
        const int tile = pos + 4 * ((wave >> 2) + nws * t);
        if (tile < ntiles) { tile_rc(tile, nt_used, tr[t], tc[t]); nmine = t + 1; }
    }
    switch (nmine) {                          // wave-uniform
    case 1: syrk_tiles_sets<1, NS, PLAIN, STRIDE>(yt, Gt, it_l, nchunks, tr, tc, acc, lr, lk); break;
    case 2: syrk_tiles_sets<2, NS, PLAIN, STRIDE>(yt, Gt, it_l, nchunks, tr, tc, acc, lr, lk); break;
    case 3: syrk_tiles_sets<3, NS, PLAIN, STRIDE>(yt, Gt, it_l, nchunks, tr, tc, acc, lr, lk); break;
    default: break;
    }
    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg; one index computation per entry
    // serves every set (set s of S at + s n^2, of rhs at + s n)
    const size_t sS = (size_t)n * n;
#pragma unroll
    for (int t = 0; t < 3; t++) {
        if (t >= nmine) continue;
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int R = 16 * tr[t] + lk + 4 * reg, Cc = 16 * tc[t] + lr;
            if (R >= nrow || Cc > nrow || R > Cc) continue;
            const int gr = 6 * gslot[R / 6] + R % 6;
            if (Cc == nrow) {
#pragma unroll
                for (int s = 0; s < NS; s++) atomicAdd(&rhs0[(size_t)s * n + gr], -acc[t][s][reg]);
            } else {
                double* dst = &S0[(size_t)gr * n + 6 * gslot[Cc / 6] + Cc % 6];
#pragma unroll
                for (int s = 0; s < NS; s++) atomicAdd(dst + s * sS, -acc[t][s][reg]);
            }
        }
    }
}

// The 8x8-tile class (unions of 11 .. 21 cameras; rare in a local window): up to 8 accumulator tiles per wave leave no
// registers for operand double buffering, so this class keeps the plain loop.  Each wave owns the
// tiles  wave, wave + 4, ...  of the upper triangle and runs over ALL K-chunks of the tile.
//   yt : LDS tile, column-major [col][STRIDE]; rows [0, 6*ns] used (row 6*ns = rhs row)
template <int NT, int TPW, int STRIDE>
__device__ __forceinline__ void syrk_scatter_plain(const double* yt, int nchunks, int ns, const int* gslot, int n,
                                             int wave, int nw, double* __restrict__ S, double* __restrict__ rhs)
{
    const int lane = threadIdx.x & 63;
    const int lr = lane & 15, lk = lane >> 4;
    const int nrow = 6 * ns;                 // rhs row index
    const int nt_used = (nrow + 16) / 16;    // tile rows that carry data (incl. the rhs row)
    d4 acc[TPW];
    int tr[TPW], tc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; t++) {
        acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
        int r = 0, c = NT;
        if (wave + nw * t < NT * (NT + 1) / 2) tile_rc(wave + nw * t, NT, r, c);
        tr[t] = r; tc[t] = c;               // c >= nt_used marks an unused slot
    }
#pragma unroll 4
    for (int kc = 0; kc < nchunks; kc++) {
        const double* col = yt + (size_t)(kc * 4 + lk) * STRIDE + lr;
#pragma unroll
        for (int t = 0; t < TPW; t++) {
            if (tc[t] < nt_used) {
                const double a = col[16 * tr[t]], bb = col[16 * tc[t]];
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc[t], 0, 0, 0);
            }
        }
    }
    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int t = 0; t < TPW; t++) {
        if (tc[t] >= nt_used) continue;
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int R = 16 * tr[t] + lk + 4 * reg, Cc = 16 * tc[t] + lr;
            const double v = acc[t][reg];
            if (R >= nrow || Cc > nrow || R > Cc) continue;
            const int gr = 6 * gslot[R / 6] + R % 6;
            if (Cc == nrow) atomicAdd(&rhs[gr], -v);
            else atomicAdd(&S[(size_t)gr * n + 6 * gslot[Cc / 6] + Cc % 6], -v);
        }
    }
}

// ---------------------------------------------------------------------- K5
// One workgroup = one item of IT_L = 64 sorted landmarks (16 per wave).
// PREP_LDS: the per-camera blocks (rotation, left Jacobian, centre) of ALL cameras are staged in LDS (windows of
// up to SCH_MAXC_LDS cameras); larger windows read them from global memory (L2-resident).  The U / gc partial
// sums of a workgroup live in LDS indexed by the camera's RANK in the item's union (<= 21 cameras), so the
// footprint does not grow with the window.
#define SCH_UCAP 21
// ROUND: the body runs as the first phase of an item workgroup of ba_round (ba_solve.hip, one launch per LM round):
// `item` is not blockIdx.x, the workgroup has K7's 512 threads whatever g.it_l is (lanes beyond the item's landmarks
// idle), the state block is owned by K7's workgroup, and at the end the workgroup counts itself on BA_SDONE once its
// atomics have been performed.  *st_sh (shared memory) receives the round's state.
template <bool PREP_LDS, bool ROUND>
static __device__ __forceinline__ void ba_schur_body(const BaDims& d, const BaBufs& b, const BaOpt& opt, const BaGroup& g, int it,
                                                     const int item, BaState* st_sh)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    BA_STAMP_DECL;
#if RS_STAMPS
    const unsigned long long t_item0 = wall_clock64();
#endif
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    double* yt = lds;                                            // WG tile
    double* ulds = lds + sch_tile_doubles(g.it_l);               // [SCH_UCAP][42], by rank in the item's camera union
    double* cprep = ulds + SCH_UCAP * 42;                        // [C][BA_PREP_LDS] camera blocks (PREP_LDS only)
    int* gslot = (int*)(cprep + (PREP_LDS ? (size_t)d.C * BA_PREP_LDS : 0));   // [24]
    const int nlds = SCH_UCAP * 42;
    // ---- everything that does not depend on the LM state goes out before the state barrier: the item's
    // camera mask, this lane's landmark record {landmark, first observation, count} (one 16-byte load
    // instead of the chain sorted -> obs_ptr), LDS zeroing
    const uint64_t um0 = g.item_mask[2 * (size_t)item], um1 = g.item_mask[2 * (size_t)item + 1];
    const int l = lane & 7, sub = lane >> 3;         // 8 landmarks per wave, 8 lanes each
    const int wl = 8 * wave + l;                     // landmark slot inside the item
    const int q = item * g.it_l + wl;
    const int4 lmq = (q < d.P && wl < g.it_l) ? g.lm[q] : make_int4(-1, 0, 0, 0);
    const int p = lmq.x, o0 = lmq.y, nobs = lmq.z;
    for (int i = threadIdx.x; i < nlds; i += blockDim.x) ulds[i] = 0.0;
    const int yt_used = max(3 * g.it_l * YT_STRIDE4, 3 * (g.it_l / 2) * YT_STRIDE8);     // what an item of it_l landmarks can touch
    for (int i = threadIdx.x; i < yt_used; i += blockDim.x) yt[i] = 0.0;      // first batch's tile, under the load latency
    const BaState st = ba_round_state(b, opt, it, st_sh, !ROUND && item == 0);
    if (st.done) return;
    // ---- one more round trip: camera blocks -> LDS, the landmark, the first observation of every lane
    const double* gprep = b.prep + (size_t)st.cur * d.C * BA_PREP;
    if (PREP_LDS)
        for (int i = threadIdx.x; i < d.C * BA_PREP; i += blockDim.x) cprep[(i / BA_PREP) * BA_PREP_LDS + i % BA_PREP] = gprep[i];
    const double* prep = PREP_LDS ? (const double*)cprep : gprep;       // pure LDS or pure global pointer per instantiation
    constexpr int PSTR = PREP_LDS ? BA_PREP_LDS : BA_PREP;              // its row stride
    const double* Xp = b.Xp + (size_t)st.cur * d.P * 3;
    double X[3] = {0, 0, 0};
    if (p >= 0) { X[0] = Xp[3 * (size_t)p]; X[1] = Xp[3 * (size_t)p + 1]; X[2] = Xp[3 * (size_t)p + 2]; }
    // observation j of the lane's landmark, staggered so that the lanes of one round hit different cameras
    int jj0 = sub + l; while (nobs > 0 && jj0 >= nobs) jj0 -= nobs;
    int cs0 = 0;
    float2 uv0 = make_float2(0.f, 0.f);
    if (sub < nobs) { cs0 = g.obs_cs[o0 + jj0]; uv0 = b.obs_uv[o0 + jj0]; }
    const int ns = __popcll(um0) + __popcll(um1);
    if (threadIdx.x < 24) gslot[threadIdx.x] = (int)threadIdx.x < ns ? nth_set_bit(um0, um1, threadIdx.x) : 0;
    __syncthreads();

    BA_STAMP(b, 0);
    const size_t rep_off = (size_t)(item & (BA_UREP - 1)) * b.cam_stride;
    double* rhs_rep = b.rhs + rep_off;
    double cost = 0.0, gmax = 0.0;
    double fail[BA_MAXSETS];
#pragma unroll
    for (int k = 0; k < BA_MAXSETS; k++) fail[k] = 0.0;
    // ---- pass 1: V, g, cost, U/gc
    double V[6] = {0, 0, 0, 0, 0, 0}, gv[3] = {0, 0, 0};
    ObsLin o;
    // A rejected step leaves x where it was: V, g, U, gc and the cost are those of the last fresh linearisation
    // (only the damping changes), so this pass runs on fresh iterations only; V comes back from b.Vc, g from b.gp,
    // and K7 takes U / gc from its own copy.
    if (st.fresh)
    for (int j = sub; j < nobs; j += SCH_SUBS) {
        int cs = cs0;
        float2 uvv = uv0;
        if (j != sub) {
            int jj = j + l; while (jj >= nobs) jj -= nobs;
            cs = g.obs_cs[o0 + jj];
            uvv = b.obs_uv[o0 + jj];
        }
        const int c = cs & 0xFFFF;
        obs_eval<true>(prep + (size_t)c * PSTR, X, uvv, d, o);
        cost += 0.5 * o.rho;
        const double w = o.w;
        V[0] += w * (o.jp[0] * o.jp[0] + o.jp[3] * o.jp[3]);
        V[1] += w * (o.jp[0] * o.jp[1] + o.jp[3] * o.jp[4]);
        V[2] += w * (o.jp[0] * o.jp[2] + o.jp[3] * o.jp[5]);
        V[3] += w * (o.jp[1] * o.jp[1] + o.jp[4] * o.jp[4]);
        V[4] += w * (o.jp[1] * o.jp[2] + o.jp[4] * o.jp[5]);
        V[5] += w * (o.jp[2] * o.jp[2] + o.jp[5] * o.jp[5]);
#pragma unroll
        for (int k = 0; k < 3; k++) gv[k] += w * (o.jp[k] * o.r0 + o.jp[3 + k] * o.r1);
        const int s = (cs >> 16) - 1;
        if (s >= 0) {
            if (ns <= SCH_UCAP) {
                double* u = ulds + rank_in_mask(um0, um1, s) * 42;
#pragma unroll
                for (int a = 0; a < 6; a++) {
#pragma unroll
                    for (int e = a; e < 6; e++) atomicAdd(&u[a * 6 + e], w * (o.jc[a] * o.jc[e] + o.jc[6 + a] * o.jc[6 + e]));
                    atomicAdd(&u[36 + a], w * (o.jc[a] * o.r0 + o.jc[6 + a] * o.r1));
                }
            } else {        // union too large for the LDS table (generic covisibility): straight to the replicas
#pragma unroll
                for (int a = 0; a < 6; a++) {
#pragma unroll
                    for (int e = a; e < 6; e++) atomicAdd(&b.U[rep_off + s * 36 + a * 6 + e], w * (o.jc[a] * o.jc[e] + o.jc[6 + a] * o.jc[6 + e]));
                    atomicAdd(&b.gc[rep_off + 6 * s + a], w * (o.jc[a] * o.r0 + o.jc[6 + a] * o.r1));
                }
            }
        }
    }
    BA_STAMP(b, 1);
    // the 8 sub-lanes of a landmark (lanes l + 8 s) combine their partial sums
#pragma unroll
    for (int k = 0; k < 6; k++) { V[k] += __shfl_xor(V[k], 8, 64); V[k] += __shfl_xor(V[k], 16, 64); V[k] += __shfl_xor(V[k], 32, 64); }
#pragma unroll
    for (int k = 0; k < 3; k++) { gv[k] += __shfl_xor(gv[k], 8, 64); gv[k] += __shfl_xor(gv[k], 16, 64); gv[k] += __shfl_xor(gv[k], 32, 64); }
    if (p >= 0) {
        if (st.fresh) {
            if (sub == 0) {
#pragma unroll
                for (int k = 0; k < 6; k++) b.Vc[6 * (size_t)p + k] = V[k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 6; k++) V[k] = b.Vc[6 * (size_t)p + k];
#pragma unroll
            for (int k = 0; k < 3; k++) gv[k] = b.gp[3 * (size_t)p + k];
        }
    }

    // Jacobi scale of the point block (fixed at the first linearisation) and the gradient norm: once, for all sets
    double sp[3] = {1.0, 1.0, 1.0};
    const double Vd[3] = {V[0], V[3], V[5]};
    if (p >= 0) {
        if (sub == 0) gmax = fmax(fabs(gv[0]), fmax(fabs(gv[1]), fabs(gv[2])));
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (!st.have_scale) sp[k] = opt.jacobi ? 1.0 / (1.0 + sqrt(Vd[k])) : 1.0;
            else sp[k] = b.sp[3 * (size_t)p + k];
        }
        if (sub == 0) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                if (!st.have_scale) b.sp[3 * (size_t)p + k] = sp[k];
                b.gp[3 * (size_t)p + k] = gv[k];
            }
        }
    }
    // ---- per speculative radius (set): damped V^-1, Y, SYRK into the set's own S / rhs (ba_common.h "Speculative radii")
    // Sets >= 1 do not recompute Y: with V + Lambda_s = L_s L_s^T,  Y_s = W L_s^-T = Y_{s-1} (L_{s-1}^T L_s^-T), i.e. the
    // three tile columns of a landmark are recombined by a 3x3 upper-triangular matrix M^T, M = L_s^-1 L_{s-1}; the
    // rhs row (g^T L^-T) transforms the same way.  One pass over the tile in LDS instead of zeroing it and evaluating
    // every observation again (compact 4x4-tile items only: the 8x8 class reuses the tile for two half batches).
    // The workgroup's scalar sums (cost, gradient maximum, failure counts) and its U / gc block sums go to global memory by a
    // handful of atomics.  They are final once pass 1 and the Cholesky factors are done, so the all-sets path issues them
    // BEFORE its SYRK + scatter: behind the scatter's ~24 atomic instructions per wave they queued for ~3 us at the end of the
    // kernel (a wave stalls at 16 - 32 outstanding atomics); the other paths run them at the end as before.
    __shared__ double redw[SCH_WAVES][2 + BA_MAXSETS];
    bool epilogue_done = false;
    auto epilogue = [&]() {
        epilogue_done = true;
        cost = wave_sum(cost);
#pragma unroll
        for (int k = 0; k < BA_MAXSETS; k++) fail[k] = wave_sum(fail[k]);
        gmax = wave_max_nonneg(gmax);
        const int nwaves = (int)(blockDim.x >> 6);
        if (lane == 0) {
            redw[wave][0] = cost; redw[wave][1] = gmax;
#pragma unroll
            for (int k = 0; k < BA_MAXSETS; k++) redw[wave][2 + k] = fail[k];
        }
        __syncthreads();
        if (threadIdx.x == 0) {      // one atomic per workgroup, spread over BA_NSLOT lines
            double c = 0.0, gm = 0.0, f[BA_MAXSETS];
#pragma unroll
            for (int k = 0; k < BA_MAXSETS; k++) f[k] = 0.0;
            for (int w = 0; w < nwaves; w++) {
                c += redw[w][0]; gm = fmax(gm, redw[w][1]);
#pragma unroll
                for (int k = 0; k < BA_MAXSETS; k++) f[k] += redw[w][2 + k];
            }
            const size_t slot = (size_t)(item & (BA_NSLOT - 1)) * BA_SLOT_STRIDE;
            if (c != 0.0) atomicAdd(&b.scal[slot], c);
#pragma unroll
            for (int k = 0; k < BA_MAXSETS; k++)
                if (f[k] > 0.0) atomicAdd(&b.scal[slot + 1 + k], f[k]);      // slot field 1 + set
            if (gm > 0.0) atomic_max_nonneg(&b.gmax[slot], gm);
        }
        for (int i = threadIdx.x; i < min(ns, SCH_UCAP) * 42; i += blockDim.x) {
            const int s = gslot[i / 42], k = i % 42;          // rank in the union -> free-camera slot
            const double v = ulds[i];
            if (v != 0.0) {
                if (k < 36) atomicAdd(&b.U[rep_off + s * 36 + k], v);
                else atomicAdd(&b.gc[rep_off + 6 * s + (k - 36)], v);
            }
        }
    };
    double Lprev[6] = {0, 0, 0, 0, 0, 0};
    bool prev_all_ok = false;
    double* Mt = (double*)(gslot + 32);                       // [it_l][6] behind the slot table ([2][it_l][6] as Gt)
    int set_first = 0;
    // ---- several radii, compact 4x4-tile item: every set in ONE pass over the tile (syrk_tiles_sets above)
    if (K5_ALLSETS && st.nact > 1 && ns > 0 && ns <= 10) {
        double Li0[6] = {0, 0, 0, 0, 0, 0}, L0[6] = {0, 0, 0, 0, 0, 0};
        // damped block of one set: damping, Cholesky, inverse; false when the block is not positive definite
        auto damped = [&](int set, double lam[3], double Li[6], double I[6], double Lc[6]) -> bool {
            const double radius = ba_set_radius(st, set);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const double s2 = sp[k] * sp[k];
                lam[k] = clampd(s2 * Vd[k], opt.dmin, opt.dmax) / (radius * s2);
            }
            const double Vdm[6] = {V[0] + lam[0], V[1], V[2], V[3] + lam[1], V[4], V[5] + lam[2]};
            const bool ok = chol3_inv(Vdm, Li, I, Lc);
            if (!ok) {
#pragma unroll
                for (int k = 0; k < 6; k++) { I[k] = 0.0; Li[k] = 0.0; Lc[k] = 0.0; }
            }
            return ok;
        };
        double lam0[3] = {0, 0, 0}, I0[6] = {0, 0, 0, 0, 0, 0};
        bool ok0 = false;
        if (p >= 0) ok0 = damped(0, lam0, Li0, I0, L0);
        // a landmark whose block is not positive definite at the FIRST radius has no Y_0 to derive the later sets from:
        // such an item (not seen on real windows) takes the set-by-set path below
        if (__syncthreads_or((p >= 0 && !ok0) ? 1 : 0) == 0) {
#pragma unroll
            for (int set = 0; set < BA_MAXSETS; set++) {
                if (set >= st.nact) continue;
                double lam[3] = {lam0[0], lam0[1], lam0[2]}, Li[6], I[6] = {I0[0], I0[1], I0[2], I0[3], I0[4], I0[5]}, Lc[6];
                bool ok = ok0;
                if (set > 0 && p >= 0) ok = damped(set, lam, Li, I, Lc);
                if (set > 0 && p < 0) {
#pragma unroll
                    for (int k = 0; k < 6; k++) I[k] = 0.0;
                }
                if (p >= 0 && sub == 0) {
                    if (!ok) {
#pragma unroll
                        for (int k = 0; k < BA_MAXSETS; k++) fail[k] = (k == set) ? 1.0 : fail[k];
                    }
                    double* lamp_set = b.lamp + ((size_t)set * d.P + p) * 3;
                    double* vinv_set = b.Vinv + ((size_t)set * d.P + p) * 6;
#pragma unroll
                    for (int k = 0; k < 3; k++) lamp_set[k] = lam[k];
#pragma unroll
                    for (int k = 0; k < 6; k++) vinv_set[k] = I[k];
                }
                if (set > 0 && sub == 0 && wl < g.it_l) {
                    // G_s = L_0^T I_s L_0 (symmetric; L_0 lower: 00 10 11 20 21 22; I_s: xx xy xz yy yz zz); zero for an empty slot
                    // T = I L_0 (3x3): T[r][c] = sum_k I[r][k] L0[k][c]
                    const double T00 = I[0] * L0[0] + I[1] * L0[1] + I[2] * L0[3], T01 = I[1] * L0[2] + I[2] * L0[4], T02 = I[2] * L0[5];
                    const double T10 = I[1] * L0[0] + I[3] * L0[1] + I[4] * L0[3], T11 = I[3] * L0[2] + I[4] * L0[4], T12 = I[4] * L0[5];
                    const double T20 = I[2] * L0[0] + I[4] * L0[1] + I[5] * L0[3], T21 = I[4] * L0[2] + I[5] * L0[4], T22 = I[5] * L0[5];
                    double* gt = Mt + ((size_t)(set - 1) * g.it_l + wl) * 6;
                    // G = L_0^T T: G[r][c] = sum_k L0[k][r] T[k][c]
                    gt[0] = L0[0] * T00 + L0[1] * T10 + L0[3] * T20;
                    gt[1] = L0[0] * T01 + L0[1] * T11 + L0[3] * T21;
                    gt[2] = L0[0] * T02 + L0[1] * T12 + L0[3] * T22;
                    gt[3] = L0[2] * T11 + L0[4] * T21;
                    gt[4] = L0[2] * T12 + L0[4] * T22;
                    gt[5] = L0[5] * T22;
                }
            }
            const bool okall0 = ok0;
            // Y_0 into the tile (the first tile was zeroed at kernel start)
            const double t0 = Li0[0] * gv[0], t1 = Li0[1] * gv[0] + Li0[2] * gv[1], t2 = Li0[3] * gv[0] + Li0[4] * gv[1] + Li0[5] * gv[2];
            BA_STAMP(b, 2);
            if (p >= 0 && okall0 && wl < g.it_l) {
                for (int j = sub; j < nobs; j += SCH_SUBS) {
                    int cs = cs0;
                    float2 uvv = uv0;                          // round 0 is still in registers
                    if (j != sub) {
                        int jj = j + l; while (jj >= nobs) jj -= nobs;
                        cs = g.obs_cs[o0 + jj];
                        uvv = b.obs_uv[o0 + jj];
                    }
                    const int c = cs & 0xFFFF;
                    const int s = (cs >> 16) - 1;
                    if (s < 0) continue;
                    obs_eval<true>(prep + (size_t)c * PSTR, X, uvv, d, o);
                    const int pos = rank_in_mask(um0, um1, s);
                    double* dst = yt + (size_t)(3 * wl) * YT_STRIDE4 + 6 * pos;
#pragma unroll
                    for (int a = 0; a < 6; a++) {
                        const double w0 = o.w * (o.jc[a] * o.jp[0] + o.jc[6 + a] * o.jp[3]);
                        const double w1 = o.w * (o.jc[a] * o.jp[1] + o.jc[6 + a] * o.jp[4]);
                        const double w2 = o.w * (o.jc[a] * o.jp[2] + o.jc[6 + a] * o.jp[5]);
                        dst[a] = w0 * Li0[0];
                        dst[YT_STRIDE4 + a] = w0 * Li0[1] + w1 * Li0[2];
                        dst[2 * YT_STRIDE4 + a] = w0 * Li0[3] + w1 * Li0[4] + w2 * Li0[5];
                    }
                }
                if (sub == 0) {
                    double* dst = yt + (size_t)(3 * wl) * YT_STRIDE4 + 6 * ns;
                    dst[0] = t0; dst[YT_STRIDE4] = t1; dst[2 * YT_STRIDE4] = t2;
                }
            }
            __syncthreads();
            BA_STAMP(b, 4);
            epilogue();                                        // (its atomics leave in front of the scatter's)
            {
                const int nw = (int)(blockDim.x >> 6);
                double* const S0 = b.S + (size_t)((unsigned)item % (unsigned)b.srep) * b.s_rep_stride;
                const int nch = 3 * g.it_l / 4;
                // sets 0 .. 2 in one pass over the tile, sets 3 .. 4 (rounds that speculate deeper) in a second one
                if (st.nact == 2) syrk_scatter_sets<2, true, YT_STRIDE4>(yt, Mt, g.it_l, nch, ns, gslot, d.n, wave, nw, S0, rhs_rep);
                else syrk_scatter_sets<3, true, YT_STRIDE4>(yt, Mt, g.it_l, nch, ns, gslot, d.n, wave, nw, S0, rhs_rep);
                if (st.nact > 3) {
                    const double* Gt2 = Mt + (size_t)2 * g.it_l * 6;                     // G of sets 3, 4
                    double* const S3 = S0 + (size_t)3 * d.n * d.n;
                    double* const rhs3 = rhs_rep + (size_t)3 * d.n;
                    if (st.nact == 4) syrk_scatter_sets<1, false, YT_STRIDE4>(yt, Gt2, g.it_l, nch, ns, gslot, d.n, wave, nw, S3, rhs3);
                    else syrk_scatter_sets<2, false, YT_STRIDE4>(yt, Gt2, g.it_l, nch, ns, gslot, d.n, wave, nw, S3, rhs3);
                }
            }
            set_first = st.nact;                               // nothing left for the set-by-set loop
        }
    }
    for (int set = set_first; set < st.nact; set++) {
    const double radius = ba_set_radius(st, set);
    double* const S_set = b.S + (size_t)((unsigned)item % (unsigned)b.srep) * b.s_rep_stride + (size_t)set * d.n * d.n;
    double* const rhs_set = rhs_rep + (size_t)set * d.n;
    double Li[6] = {0, 0, 0, 0, 0, 0}, I[6] = {0, 0, 0, 0, 0, 0}, Lc[6] = {0, 0, 0, 0, 0, 0};
    bool ok = false;
    if (p >= 0) {
        double lam[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const double s2 = sp[k] * sp[k];
            lam[k] = clampd(s2 * Vd[k], opt.dmin, opt.dmax) / (radius * s2);
        }
        const double Vdm[6] = {V[0] + lam[0], V[1], V[2], V[3] + lam[1], V[4], V[5] + lam[2]};
        ok = chol3_inv(Vdm, Li, I, Lc);
        if (!ok) {
#pragma unroll
            for (int k = 0; k < 6; k++) { I[k] = 0.0; Li[k] = 0.0; Lc[k] = 0.0; }
            if (sub == 0) {
#pragma unroll
                for (int k = 0; k < BA_MAXSETS; k++) fail[k] = (k == set) ? 1.0 : fail[k];
            }
        }
        if (sub == 0) {
            double* lamp_set = b.lamp + ((size_t)set * d.P + p) * 3;
            double* vinv_set = b.Vinv + ((size_t)set * d.P + p) * 6;
#pragma unroll
            for (int k = 0; k < 3; k++) lamp_set[k] = lam[k];
#pragma unroll
            for (int k = 0; k < 6; k++) vinv_set[k] = I[k];
        }
    }
    // t = L^-1 g : the rhs row
    const double t0 = Li[0] * gv[0], t1 = Li[1] * gv[0] + Li[2] * gv[1], t2 = Li[3] * gv[0] + Li[4] * gv[1] + Li[5] * gv[2];

    BA_STAMP(b, 2);
    // ---- pass 2: Y into the LDS tile (compact rows), SYRK on the matrix cores, scatter
    if (ns > 0 && ns <= 21) {
        const bool big = ns > 10;
        const int nbatch = big ? 2 : 1, lb_n = big ? g.it_l / 2 : g.it_l;
        const int stride = big ? YT_STRIDE8 : YT_STRIDE4;
        const bool transform = !big && set > 0 && prev_all_ok;
        if (transform) {
            const int ncol = 3 * lb_n, nrow1 = 6 * ns + 1;     // rows incl. the rhs row
            if (p >= 0 && sub == 0) {                          // M = L_s^-1 L_{s-1} (lower), row-major 00 10 11 20 21 22
                double* mt = Mt + 6 * wl;
                mt[0] = Li[0] * Lprev[0];
                mt[1] = Li[1] * Lprev[0] + Li[2] * Lprev[1];
                mt[2] = Li[2] * Lprev[2];
                mt[3] = Li[3] * Lprev[0] + Li[4] * Lprev[1] + Li[5] * Lprev[3];
                mt[4] = Li[4] * Lprev[2] + Li[5] * Lprev[4];
                mt[5] = Li[5] * Lprev[5];
            }
            __syncthreads();                                   // previous SYRK has consumed the tile; Mt is written
            for (int idx = threadIdx.x; idx < lb_n * nrow1; idx += blockDim.x) {
                const int lbq = idx / nrow1, r = idx - lbq * nrow1;
                if (item * g.it_l + lbq >= d.P) continue;       // no landmark in this slot of the last item
                double* c0 = yt + (size_t)(3 * lbq) * YT_STRIDE4 + r;
                const double* mt = Mt + 6 * lbq;
                const double y0 = c0[0], y1 = c0[YT_STRIDE4], y2 = c0[2 * YT_STRIDE4];
                c0[0] = y0 * mt[0];
                c0[YT_STRIDE4] = y0 * mt[1] + y1 * mt[2];
                c0[2 * YT_STRIDE4] = y0 * mt[3] + y1 * mt[4] + y2 * mt[5];
            }
            prev_all_ok = __syncthreads_or((p >= 0 && !ok) ? 1 : 0) == 0;
            const int nw = (int)(blockDim.x >> 6);
            syrk_scatter<4, 3, YT_STRIDE4>(yt, ncol / 4, ns, gslot, d.n, wave, nw, S_set, rhs_set);
        } else
        for (int bt = 0; bt < nbatch; bt++) {
            const int ncol = 3 * lb_n;
            if (bt > 0 || set > 0) {                           // the very first tile was zeroed at kernel start
                __syncthreads();                               // previous batch fully consumed
                for (int i = threadIdx.x; i < ncol * stride; i += blockDim.x) yt[i] = 0.0;
                __syncthreads();
            }
            BA_STAMP(b, 3);
            const int lb = wl - bt * lb_n;
            if (p >= 0 && ok && lb >= 0 && lb < lb_n) {
                for (int j = sub; j < nobs; j += SCH_SUBS) {
                    int cs = cs0;
                    float2 uvv = uv0;                          // round 0 is still in registers
                    if (j != sub) {
                        int jj = j + l; while (jj >= nobs) jj -= nobs;
                        cs = g.obs_cs[o0 + jj];
                        uvv = b.obs_uv[o0 + jj];
                    }
                    const int c = cs & 0xFFFF;
                    const int s = (cs >> 16) - 1;
                    if (s < 0) continue;
                    obs_eval<true>(prep + (size_t)c * PSTR, X, uvv, d, o);
                    const int pos = rank_in_mask(um0, um1, s);
                    double* dst = yt + (size_t)(3 * lb) * stride + 6 * pos;
#pragma unroll
                    for (int a = 0; a < 6; a++) {
                        const double w0 = o.w * (o.jc[a] * o.jp[0] + o.jc[6 + a] * o.jp[3]);
                        const double w1 = o.w * (o.jc[a] * o.jp[1] + o.jc[6 + a] * o.jp[4]);
                        const double w2 = o.w * (o.jc[a] * o.jp[2] + o.jc[6 + a] * o.jp[5]);
                        // Y = W L^-T :  Y[a][dd] = sum_e W[a][e] Linv[dd][e]
                        dst[a] = w0 * Li[0];
                        dst[stride + a] = w0 * Li[1] + w1 * Li[2];
                        dst[2 * stride + a] = w0 * Li[3] + w1 * Li[4] + w2 * Li[5];
                    }
                }
                if (sub == 0) {
                    double* dst = yt + (size_t)(3 * lb) * stride + 6 * ns;
                    dst[0] = t0; dst[stride] = t1; dst[2 * stride] = t2;
                }
            }
            prev_all_ok = __syncthreads_or((p >= 0 && !ok) ? 1 : 0) == 0;      // (the barrier the SYRK needs anyway)
            BA_STAMP(b, 4);
            const int nw = (int)(blockDim.x >> 6);            // 8 (64 landmarks) or 5 (40 landmarks)
            if (big) {
                if (nw >= 8) syrk_scatter_plain<8, 5, YT_STRIDE8>(yt, ncol / 4, ns, gslot, d.n, wave, nw, S_set, rhs_set);
                else syrk_scatter_plain<8, 8, YT_STRIDE8>(yt, ncol / 4, ns, gslot, d.n, wave, nw, S_set, rhs_set);
            } else {
                syrk_scatter<4, 3, YT_STRIDE4>(yt, ncol / 4, ns, gslot, d.n, wave, nw, S_set, rhs_set);
            }
        }
    } else if (ns > 21) {
        // generic fallback: per-landmark f64 atomics (any covisibility pattern)
        if (p >= 0 && ok && sub == 0) {
            for (int oi = o0; oi < o0 + nobs; oi++) {
                const int si = b.slot[b.obs_cam[oi]];
                if (si < 0) continue;
                obs_eval<true>(prep + (size_t)b.obs_cam[oi] * PSTR, X, b.obs_uv[oi], d, o);
                double Y[18];
#pragma unroll
                for (int a = 0; a < 6; a++) {
                    const double w0 = o.w * (o.jc[a] * o.jp[0] + o.jc[6 + a] * o.jp[3]);
                    const double w1 = o.w * (o.jc[a] * o.jp[1] + o.jc[6 + a] * o.jp[4]);
                    const double w2 = o.w * (o.jc[a] * o.jp[2] + o.jc[6 + a] * o.jp[5]);
                    Y[a * 3 + 0] = w0 * I[0] + w1 * I[1] + w2 * I[2];
                    Y[a * 3 + 1] = w0 * I[1] + w1 * I[3] + w2 * I[4];
                    Y[a * 3 + 2] = w0 * I[2] + w1 * I[4] + w2 * I[5];
                    atomicAdd(&rhs_set[6 * si + a], -(Y[a * 3] * gv[0] + Y[a * 3 + 1] * gv[1] + Y[a * 3 + 2] * gv[2]));
                }
                ObsLin oj;
                for (int ojx = o0; ojx < o0 + nobs; ojx++) {
                    const int sj = b.slot[b.obs_cam[ojx]];
                    if (sj < si) continue;      // upper block triangle only
                    obs_eval<true>(prep + (size_t)b.obs_cam[ojx] * PSTR, X, b.obs_uv[ojx], d, oj);
                    double* Sblk = S_set + (size_t)(6 * si) * d.n + 6 * sj;
#pragma unroll
                    for (int e = 0; e < 6; e++) {
                        const double w0 = oj.w * (oj.jc[e] * oj.jp[0] + oj.jc[6 + e] * oj.jp[3]);
                        const double w1 = oj.w * (oj.jc[e] * oj.jp[1] + oj.jc[6 + e] * oj.jp[4]);
                        const double w2 = oj.w * (oj.jc[e] * oj.jp[2] + oj.jc[6 + e] * oj.jp[5]);
#pragma unroll
                        for (int a = 0; a < 6; a++)
                            if (sj > si || a <= e)
                                atomicAdd(&Sblk[(size_t)a * d.n + e], -(Y[a * 3] * w0 + Y[a * 3 + 1] * w1 + Y[a * 3 + 2] * w2));
                    }
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 6; k++) Lprev[k] = Lc[k];
    }   // sets
    BA_STAMP(b, 5);
    if (!epilogue_done) epilogue();
    BA_STAMP(b, 6);
    BA_STAMP_FLUSH(b, 8);
    if (ROUND) {
        // every atomic of this workgroup has been performed at the memory side (vmcnt counts them until they are acknowledged)
        // before it counts itself: K7's workgroups read the accumulators behind the counter with L1-bypassing loads
#if RS_STAMPS
        if (threadIdx.x == 0) {
            const unsigned long long te = wall_clock64();
            atomicMax(b.dbg + 38, te);        // latest end of an item's arithmetic
            const unsigned long long key = ((te - t_item0) << 32) | ((unsigned long long)item << 8) | (unsigned long long)ns;
            atomicMax(b.dbg + 26, key);       // slowest item: duration (10 ns ticks) | item | cameras in its union
            atomicMin(b.dbg + 27, key);       // fastest
            atomicMax(b.dbg + 28, t_item0);   // latest start
            atomicMin(b.dbg + 29, t_item0);   // earliest start
        }
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#if RS_STAMPS
        if (threadIdx.x == 0) atomicMax(b.dbg + 37, wall_clock64());        // latest drained item
#endif
        if (threadIdx.x == 0) __hip_atomic_fetch_add(b.dbg + BA_SDONE, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

