// ba_schur.hip — K5: linearisation + point-block Schur complement on the f64
// matrix cores (v_mfma_f64_16x16x4_f64), and the one-off landmark grouping
// that makes it possible.
//
// Replaces, per LM iteration, the residual/Jacobian evaluation
// (ReprojectionError, reference src/Optimization.cpp:21-72), the Huber
// corrector and Ceres' SchurEliminator inside ceres::Solve (:360).
//
// Idea.  S = U + Lambda - sum_p W_p V_p^-1 W_p^T is a sum of rank-3 updates.
// With V_p = L_p L_p^T and Y_p = W_p L_p^-T (6k x 3 for a landmark seen by k free
// cameras) the sum is a SYRK:  S -= Y Y^T  with Y = [Y_1 Y_2 ...].  Landmarks are
// sorted once per solve by (first, last) free-camera slot, so 16 consecutive
// landmarks ("an item") touch a small union of cameras (<= 10 for a local
// window).  One WAVE owns one item: it writes the 16 Y_p into a COMPACT row
// space (6 rows per camera of the union, + 1 row carrying L_p^-1 g_p) in LDS,
// then runs a dense 64 x 48 (x 64) SYRK on the matrix cores — K = 48 columns,
// no padding along K — and scatter-adds the small dense result into the global
// S / rhs once per item.  The extra row makes the reduced right-hand side
// sum_p W_p V_p^-1 g_p fall out of the same MFMAs.  Jacobians live only in
// registers; nothing per-observation is written to HBM.
//
// Item classes: union <= 10 cameras -> 4x4 tiles (NT = 4, 16 landmarks per
// SYRK); <= 21 cameras -> 8x8 tiles (NT = 8, two half-items of 8 landmarks to
// stay inside the same 30 KB LDS tile); larger unions fall back to per-landmark
// f64 atomics (correct for any covisibility, slow).
#include "ba_common.h"

#define IT_L 64                 // landmarks per item, upper bound (= per workgroup; 8 per wave); the actual
                                // count g.it_l is 40 when that puts every item on its own CU (P <= 256 * 40)
#define IT_L_SMALL 40
#define SCH_WAVES 8             // waves per workgroup (2 per SIMD: latency hiding; <= 2 SYRK tiles per wave)
#define SCH_SUBS 8              // lanes sharing one landmark (8 landmarks per wave)
#define YT_STRIDE4 81           // doubles per K-column, NT = 4: 64 rows + 17 (17 mod 32 keeps the two
                                // half-wave column groups of a ds_read_b64 on disjoint banks; 3*81*2 = 6 mod 32
                                // spreads the 16 producer lanes over 16 bank pairs)
#define YT_STRIDE8 145          // NT = 8: 128 rows + 17
#define YT_DOUBLES (192 * YT_STRIDE4)  // 15552 doubles = 124416 B per workgroup (NT=8: 96 cols * 145 = 13920)
#define SCH_PRE 3               // observation rounds prefetched per lane (covers 12 observations per landmark)
#define SCH_MAXC_LDS 64         // cameras staged in LDS when the window has at most this many

typedef __attribute__((ext_vector_type(4))) double d4;

// ------------------------------------------------------------ setup kernels
// Counting sort of the landmarks by (first, last) free-camera slot.  Histogram and cursors are
// kept in GRP_REP replicas (workgroup w uses replica w % GRP_REP) and the histogram is first
// accumulated in LDS: device-scope atomics on one 64-B line serialise (~12 ns each), and the
// ~170 live buckets of a local window share a dozen lines.
#define GRP_REP 8
#define GRP_LDS_BINS 4096

static __device__ __forceinline__ void ba_group_count_body(const BaDims& d, const BaBufs& b, const BaGroup& g)
{
    __shared__ int lh[GRP_LDS_BINS];
    const int nb = g.n_buckets + 1;
    const bool use_lds = nb <= GRP_LDS_BINS;
    if (use_lds) for (int i = threadIdx.x; i < nb; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    int* hist = g.hist + (size_t)(blockIdx.x & (GRP_REP - 1)) * nb;
    if (p < d.P) {
        uint64_t m0 = 0, m1 = 0;
        int first = 1 << 30, last = -1;
        for (int o = b.obs_ptr[p]; o < b.obs_ptr[p + 1]; o++) {
            const int c = b.obs_cam[o];
            const int s = b.slot[c];
            g.obs_cs[o] = c | ((s + 1) << 16);
            if (s < 0) continue;
            if (s < 64) m0 |= 1ull << s; else m1 |= 1ull << (s - 64);
            first = min(first, s);
            last = max(last, s);
        }
        g.mask[2 * (size_t)p] = m0;
        g.mask[2 * (size_t)p + 1] = m1;
        const int bk = last < 0 ? d.Cf * d.Cf : first * d.Cf + last;
        g.bucket[p] = bk;
        if (use_lds) atomicAdd(&lh[bk], 1); else atomicAdd(&hist[bk], 1);
    }
    __syncthreads();
    if (use_lds)
        for (int i = threadIdx.x; i < nb; i += blockDim.x) { const int v = lh[i]; if (v) atomicAdd(&hist[i], v); }
}

static __device__ __forceinline__ void ba_group_scan_body(const BaGroup& g)
{
    // exclusive scan over (bucket-major, replica-minor) of hist[rep][bucket] into cursor[rep][bucket]
    __shared__ int wsum[16];
    __shared__ int carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nb = g.n_buckets + 1, total = nb * GRP_REP;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < total; base += 1024) {
        const int i = base + threadIdx.x;
        const int bk = i / GRP_REP, rep = i % GRP_REP;
        const int v = i < total ? g.hist[(size_t)rep * nb + bk] : 0;
        int x = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(x, off, 64);
            if (lane >= off) x += t;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int pre = carry;
        for (int w = 0; w < wave; w++) pre += wsum[w];
        if (i < total) g.cursor[(size_t)rep * nb + bk] = pre + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = pre + x;
        __syncthreads();
    }
}


static __device__ __forceinline__ void ba_group_scatter_body(const BaDims& d, const BaBufs& b, const BaGroup& g)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= d.P) return;
    const int nb = g.n_buckets + 1;
    const int pos = atomicAdd(&g.cursor[(size_t)(blockIdx.x & (GRP_REP - 1)) * nb + g.bucket[p]], 1);
    g.sorted[pos] = p;
    const int o0 = b.obs_ptr[p];
    g.lm[pos] = make_int4(p, o0, b.obs_ptr[p + 1] - o0, 0);
}

// one wave per item: lanes = the item's 64 landmarks, 128-bit OR across the wave
static __device__ __forceinline__ void ba_group_items_body(const BaDims& d, const BaGroup& g)
{
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= g.n_items) return;
    const int q = t * g.it_l + (threadIdx.x & 63);
    uint64_t m0 = 0, m1 = 0;
    if ((int)(threadIdx.x & 63) < g.it_l && q < d.P) {
        const int p = g.sorted[q];
        m0 = g.mask[2 * (size_t)p];
        m1 = g.mask[2 * (size_t)p + 1];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        m0 |= (uint64_t)__shfl_xor((unsigned long long)m0, off, 64);
        m1 |= (uint64_t)__shfl_xor((unsigned long long)m1, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        g.item_mask[2 * (size_t)t] = m0;
        g.item_mask[2 * (size_t)t + 1] = m1;
    }
}

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
template <bool PREP_LDS>
static __device__ __forceinline__ void ba_schur_body(const BaDims& d, const BaBufs& b, const BaOpt& opt, const BaGroup& g, int it)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    BA_STAMP_DECL;
    __shared__ BaState st_sh;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    double* yt = lds;                                            // WG tile
    double* ulds = lds + YT_DOUBLES;                             // [SCH_UCAP][42], by rank in the item's camera union
    double* cprep = ulds + SCH_UCAP * 42;                        // [C][BA_PREP_LDS] camera blocks (PREP_LDS only)
    int* gslot = (int*)(cprep + (PREP_LDS ? (size_t)d.C * BA_PREP_LDS : 0));   // [24]
    const int nlds = SCH_UCAP * 42;
    // ---- everything that does not depend on the LM state goes out before the state barrier: the item's
    // camera mask, this lane's landmark record {landmark, first observation, count} (one 16-byte load
    // instead of the chain sorted -> obs_ptr), LDS zeroing
    const int item = blockIdx.x;
    const uint64_t um0 = g.item_mask[2 * (size_t)item], um1 = g.item_mask[2 * (size_t)item + 1];
    const int l = lane & 7, sub = lane >> 3;         // 8 landmarks per wave, 8 lanes each
    const int wl = 8 * wave + l;                     // landmark slot inside the item
    const int q = item * g.it_l + wl;
    const int4 lmq = q < d.P ? g.lm[q] : make_int4(-1, 0, 0, 0);
    const int p = lmq.x, o0 = lmq.y, nobs = lmq.z;
    for (int i = threadIdx.x; i < nlds; i += blockDim.x) ulds[i] = 0.0;
    const int yt_used = max(3 * g.it_l * YT_STRIDE4, 3 * (g.it_l / 2) * YT_STRIDE8);     // what an item of it_l landmarks can touch
    for (int i = threadIdx.x; i < yt_used; i += blockDim.x) yt[i] = 0.0;      // first batch's tile, under the load latency
    const BaState st = ba_state_for_iteration(b, opt, it, &st_sh);
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
    const size_t rep_off = (size_t)(blockIdx.x & (BA_UREP - 1)) * b.cam_stride;
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
    double Lprev[6] = {0, 0, 0, 0, 0, 0};
    bool prev_all_ok = false;
    double* Mt = (double*)(gslot + 32);                       // [it_l][6] behind the slot table
    for (int set = 0; set < st.nact; set++) {
    const double radius = ba_set_radius(st, set);
    double* const S_set = b.S + (size_t)(blockIdx.x % (unsigned)b.srep) * b.s_rep_stride + (size_t)set * d.n * d.n;
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
    cost = wave_sum(cost);
#pragma unroll
    for (int k = 0; k < BA_MAXSETS; k++) fail[k] = wave_sum(fail[k]);
    gmax = wave_max_nonneg(gmax);
    __shared__ double redw[SCH_WAVES][2 + BA_MAXSETS];
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
        const size_t slot = (size_t)(blockIdx.x & (BA_NSLOT - 1)) * BA_SLOT_STRIDE;
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
    BA_STAMP(b, 6);
    BA_STAMP_FLUSH(b, 8);
}

__global__ __launch_bounds__(64 * SCH_WAVES) void ba_schur_mfma(BaDims d, BaBufs b, BaOpt opt, BaGroup g, int it)
{
    ba_schur_body<true>(d, b, opt, g, it);
}

// windows of more than SCH_MAXC_LDS cameras (cfg 5: 100 key frames): camera blocks from global memory
__global__ __launch_bounds__(64 * SCH_WAVES) void ba_schur_mfma_big(BaDims d, BaBufs b, BaOpt opt, BaGroup g, int it)
{
    ba_schur_body<false>(d, b, opt, g, it);
}


// single-window and batched (blockIdx.z = window, arguments from the device array) entry points of the grouping kernels
__global__ __launch_bounds__(256) void ba_group_count(BaDims d, BaBufs b, BaGroup g) { ba_group_count_body(d, b, g); }
__global__ __launch_bounds__(1024) void ba_group_scan(BaGroup g) { ba_group_scan_body(g); }
__global__ __launch_bounds__(256) void ba_group_scatter(BaDims d, BaBufs b, BaGroup g) { ba_group_scatter_body(d, b, g); }

// Scatter with the scan inside (local windows: at most GRP_SCAN_LDS histogram entries): every workgroup scans the
// (bucket-major, replica-minor) histogram for itself in LDS — 2.6 k entries for 18 free cameras, one chunk per thread and
// one workgroup scan — and takes positions as base + atomicAdd on a cursor array that ba_init left at zero.  Replaces the
// one-workgroup scan launch between count and scatter (a launch gap + 4.8 us for 10 KB of work).
#define GRP_SCAN_LDS 4096
__global__ __launch_bounds__(256) void ba_group_scatter_scan(BaDims d, BaBufs b, BaGroup g)
{
    __shared__ int base[GRP_SCAN_LDS];
    const int nb = g.n_buckets + 1, total = nb * GRP_REP;
    const int per = (total + 255) / 256;                  // <= 16
    const int i0 = (int)threadIdx.x * per;
    int v[GRP_SCAN_LDS / 256];
    int sum = 0;
#pragma unroll
    for (int u = 0; u < GRP_SCAN_LDS / 256; u++) {
        const int i = i0 + u;
        v[u] = (u < per && i < total) ? g.hist[(size_t)(i % GRP_REP) * nb + i / GRP_REP] : 0;
        sum += v[u];
    }
    int tot;
    int run = rs_block_exclusive_scan(sum, &tot);
#pragma unroll
    for (int u = 0; u < GRP_SCAN_LDS / 256; u++) {
        const int i = i0 + u;
        if (u < per && i < total) base[i] = run;
        run += v[u];
    }
    __syncthreads();
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= d.P) return;
    const int rep = (int)(blockIdx.x & (GRP_REP - 1)), bk = g.bucket[p];
    const int pos = base[bk * GRP_REP + rep] + atomicAdd(&g.cursor[(size_t)rep * nb + bk], 1);
    g.sorted[pos] = p;
    const int o0 = b.obs_ptr[p];
    g.lm[pos] = make_int4(p, o0, b.obs_ptr[p + 1] - o0, 0);
}
__global__ __launch_bounds__(256) void ba_group_items(BaDims d, BaGroup g) { ba_group_items_body(d, g); }
__global__ __launch_bounds__(256) void ba_group_count_batch(const BaWin* w) { const BaWin& x = w[blockIdx.z]; if ((int)(blockIdx.x * 256) < x.d.P) ba_group_count_body(x.d, x.b, x.g); }
__global__ __launch_bounds__(1024) void ba_group_scan_batch(const BaWin* w) { ba_group_scan_body(w[blockIdx.z].g); }
__global__ __launch_bounds__(256) void ba_group_scatter_batch(const BaWin* w) { const BaWin& x = w[blockIdx.z]; ba_group_scatter_body(x.d, x.b, x.g); }
__global__ __launch_bounds__(256) void ba_group_items_batch(const BaWin* w) { const BaWin& x = w[blockIdx.z]; ba_group_items_body(x.d, x.g); }
__global__ __launch_bounds__(64 * SCH_WAVES) void ba_schur_mfma_batch(const BaWin* w, BaOpt opt, int it)
{
    const BaWin& x = w[blockIdx.z];
    if ((int)blockIdx.x >= x.g.n_items) return;
    const BaBufs b = ba_win_round(x, it, false);
    ba_schur_body<true>(x.d, b, opt, x.g, it);
}

// ------------------------------------------------------------------ host glue
size_t ba_group_bytes(int P, int Cf, int M)
{
    const size_t nb = ((size_t)Cf * Cf + 2) * GRP_REP;
    const size_t ni = ((size_t)P + IT_L_SMALL - 1) / IT_L_SMALL + 1;
    return 256 * 10 + sizeof(int32_t) * (2 * (size_t)P + 2 * nb + (size_t)M) + sizeof(int4) * (size_t)P +
           sizeof(uint64_t) * (2 * (size_t)P + 2 * ni);
}

// The whole grouping in ONE workgroup (a launch costs ~4 us on this GPU; a local window has ~10^4
// landmarks): histogram and cursors in LDS, landmarks strided over the 1024 threads.
#define GRP_SMALL_P 2048
__global__ __launch_bounds__(1024) void ba_group_small(BaDims d, BaBufs b, BaGroup g)
{
    __shared__ int lh[GRP_LDS_BINS];
    __shared__ int wsum[16];
    __shared__ int carry;
    const int nb = g.n_buckets + 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < nb; i += 1024) lh[i] = 0;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int p = threadIdx.x; p < d.P; p += 1024) {
        uint64_t m0 = 0, m1 = 0;
        int first = 1 << 30, last = -1;
        for (int o = b.obs_ptr[p]; o < b.obs_ptr[p + 1]; o++) {
            const int c = b.obs_cam[o];
            const int s = b.slot[c];
            g.obs_cs[o] = c | ((s + 1) << 16);
            if (s < 0) continue;
            if (s < 64) m0 |= 1ull << s; else m1 |= 1ull << (s - 64);
            first = min(first, s);
            last = max(last, s);
        }
        g.mask[2 * (size_t)p] = m0;
        g.mask[2 * (size_t)p + 1] = m1;
        const int bk = last < 0 ? d.Cf * d.Cf : first * d.Cf + last;
        g.bucket[p] = bk;
        atomicAdd(&lh[bk], 1);
    }
    __syncthreads();
    // exclusive scan of lh in place
    for (int base = 0; base < nb; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < nb ? lh[i] : 0;
        int x = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(x, off, 64);
            if (lane >= off) x += t;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int pre = carry;
        for (int w = 0; w < wave; w++) pre += wsum[w];
        if (i < nb) lh[i] = pre + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = pre + x;
        __syncthreads();
    }
    for (int p = threadIdx.x; p < d.P; p += 1024) {
        const int pos = atomicAdd(&lh[g.bucket[p]], 1);
        g.sorted[pos] = p;
        const int o0 = b.obs_ptr[p];
        g.lm[pos] = make_int4(p, o0, b.obs_ptr[p + 1] - o0, 0);
    }
    __syncthreads();
    for (int t = wave; t < g.n_items; t += 16) {
        const int q = t * g.it_l + lane;
        uint64_t m0 = 0, m1 = 0;
        if (lane < g.it_l && q < d.P) {
            const int p = g.sorted[q];
            m0 = g.mask[2 * (size_t)p];
            m1 = g.mask[2 * (size_t)p + 1];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            m0 |= (uint64_t)__shfl_xor((unsigned long long)m0, off, 64);
            m1 |= (uint64_t)__shfl_xor((unsigned long long)m1, off, 64);
        }
        if (lane == 0) {
            g.item_mask[2 * (size_t)t] = m0;
            g.item_mask[2 * (size_t)t + 1] = m1;
        }
    }
}

void ba_group_zero_range(const BaGroup& g, int32_t** ptr, int* count)
{
    *ptr = g.hist;          // the histogram and, behind it, the cursors (ba_group_scatter_scan counts from zero)
    *count = (int)(g.cursor - g.hist) + (g.n_buckets + 1) * GRP_REP;
}

static size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

void ba_group_carve(char* base, int P, int Cf, int M, BaGroup* g)
{
    const size_t nb = ((size_t)Cf * Cf + 2) * GRP_REP;
    const size_t ni_max = ((size_t)P + IT_L_SMALL - 1) / IT_L_SMALL + 1;
    g->it_l = P <= 256 * IT_L_SMALL ? IT_L_SMALL : IT_L;
    const size_t ni = P > 0 ? ((size_t)P + g->it_l - 1) / g->it_l : 1;     // an empty landmark shard keeps one (empty) item:
    size_t off = 0;                                                            // its workgroup runs the round's decision
    g->sorted = (int32_t*)(base + off); off += al256(sizeof(int32_t) * P);
    g->bucket = (int32_t*)(base + off); off += al256(sizeof(int32_t) * P);
    g->hist = (int32_t*)(base + off); off += al256(sizeof(int32_t) * nb);
    g->cursor = (int32_t*)(base + off); off += al256(sizeof(int32_t) * nb);
    g->mask = (uint64_t*)(base + off); off += al256(sizeof(uint64_t) * 2 * P);
    g->item_mask = (uint64_t*)(base + off); off += al256(sizeof(uint64_t) * 2 * ni_max);
    g->lm = (int4*)(base + off); off += al256(sizeof(int4) * P);
    g->obs_cs = (int32_t*)(base + off); off += al256(sizeof(int32_t) * M);
    g->n_items = (int)ni;
    g->n_buckets = Cf * Cf;           // + 1 bucket for landmarks without a free camera
}

int ba_launch_grouping(rs_context* ctx, const BaDims& d, const BaBufs& b, const BaGroup& g)
{
    hipStream_t s = ctx->stream;
    rs_prof_scope ps(ctx, "K5s_group_landmarks");
    if (d.P <= GRP_SMALL_P && g.n_buckets + 1 <= GRP_LDS_BINS) {
        hipLaunchKernelGGL(ba_group_small, dim3(1), dim3(1024), 0, s, d, b, g);
        return RS_OK;
    }
    // g.hist was zeroed by ba_init (ba_group_zero_range)
    const int pb = (d.P + 255) / 256;
    hipLaunchKernelGGL(ba_group_count, dim3(pb), dim3(256), 0, s, d, b, g);
    if ((g.n_buckets + 1) * GRP_REP <= GRP_SCAN_LDS) {
        hipLaunchKernelGGL(ba_group_scatter_scan, dim3(pb), dim3(256), 0, s, d, b, g);
    } else {
        hipLaunchKernelGGL(ba_group_scan, dim3(1), dim3(1024), 0, s, g);
        hipLaunchKernelGGL(ba_group_scatter, dim3(pb), dim3(256), 0, s, d, b, g);
    }
    hipLaunchKernelGGL(ba_group_items, dim3((g.n_items + 3) / 4), dim3(256), 0, s, d, g);
    return RS_OK;
}

size_t ba_schur_lds_bytes(int C, int Cf)
{
    (void)Cf;
    const size_t prep = C <= SCH_MAXC_LDS ? (size_t)C * BA_PREP_LDS : 0;
    return sizeof(double) * ((size_t)YT_DOUBLES + (size_t)SCH_UCAP * 42 + prep + 6 * IT_L) + sizeof(int) * 32;   // + Mt
}

int ba_prepare_schur(int C, int Cf)
{
    const void* fn = C <= SCH_MAXC_LDS ? (const void*)ba_schur_mfma : (const void*)ba_schur_mfma_big;
    return (int)rs_lds_attr(fn, ba_schur_lds_bytes(C, Cf));
}

void ba_launch_schur(hipStream_t s, const BaDims& d, const BaBufs& b, const BaOpt& opt, const BaGroup& g, int it)
{
    if (d.C <= SCH_MAXC_LDS)
        hipLaunchKernelGGL(ba_schur_mfma, dim3(g.n_items), dim3(8 * g.it_l), ba_schur_lds_bytes(d.C, d.Cf), s, d, b, opt, g, it);
    else
        hipLaunchKernelGGL(ba_schur_mfma_big, dim3(g.n_items), dim3(8 * g.it_l), ba_schur_lds_bytes(d.C, d.Cf), s, d, b, opt, g, it);
}

// ---- batched launches (one per kernel for B windows)
void ba_launch_grouping_batch(hipStream_t s, const BaWin* d_wins, int B, int max_P, int max_items)
{
    const int pb = (max_P + 255) / 256;
    hipLaunchKernelGGL(ba_group_count_batch, dim3(pb, 1, B), dim3(256), 0, s, d_wins);
    hipLaunchKernelGGL(ba_group_scan_batch, dim3(1, 1, B), dim3(1024), 0, s, d_wins);
    hipLaunchKernelGGL(ba_group_scatter_batch, dim3(pb, 1, B), dim3(256), 0, s, d_wins);
    hipLaunchKernelGGL(ba_group_items_batch, dim3((max_items + 3) / 4, 1, B), dim3(256), 0, s, d_wins);
}

int ba_prepare_schur_batch(size_t lds)
{
    return (int)rs_lds_attr((const void*)ba_schur_mfma_batch, lds);
}

void ba_launch_schur_batch(hipStream_t s, const BaWin* d_wins, int B, const BaOpt& opt, int it, int max_items, int it_l, size_t lds)
{
    hipLaunchKernelGGL(ba_schur_mfma_batch, dim3(max_items, 1, B), dim3(8 * it_l), lds, s, d_wins, opt, it);
}

void ba_group_set_items(BaGroup* g, int P, bool throughput)
{
    g->it_l = throughput ? IT_L : (P <= 256 * IT_L_SMALL ? IT_L_SMALL : IT_L);
    g->n_items = P > 0 ? (P + g->it_l - 1) / g->it_l : 1;
}
