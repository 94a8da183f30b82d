// ba_solve_big.hip — K7 for windows whose reduced camera system does not fit the LDS kernel
// (n = 6 Cf > 126; cfg 5: 98 free key frames, n = 588).  Same job as ba_solve.hip — solve
// (U + Lambda_c - sum Y Y^T) x = g for the camera step (reference src/Optimization.cpp:360, the Schur
// solve of ceres::Solve) — as a right-looking BLOCKED L D L^T on the matrix in global memory
// (L2-resident: 2.8 MB at n = 588), 48 columns (8 cameras) per block step, two launches per step:
//
//   prologue (1 WG)    fold the accumulator replicas, cost / gradient test, Jacobi scale, damping, assemble
//                      the full symmetric matrix in place, y = reduced right-hand side
//   per block J:
//     diag   (1 wave)  lane = row of the 48x48 diagonal block (+ one lane for the right-hand side), the
//                      row lives in 48 registers, columns are eliminated with v_readlane broadcasts (no LDS,
//                      no barriers); then M = L_JJ^-1 the same way.  Writes L_JJ, D_J, M_J, y's block.
//     update (grid)    one workgroup per block pair (bi >= bk > J): panels P_i = A_iJ M^T D^-1 (recomputed per
//                      workgroup from the 48x48 inverse instead of a separate triangular-solve launch), then
//                      A_ik -= P_i D P_k^T; the bk == J+1 workgroups also store P_i as the factor's panel
//   finish (1 WG)      block backward substitution, camera step, candidate cameras, step scalars
//
// ~2 + 2 NB + 1 launches (29 at n = 588): launch-bound (~4.4 us each) rather than flop-bound
// (68 MFLOP), 25.9 ms -> ~0.3 ms per solve against the single-workgroup global-memory Cholesky it replaces.
#include "ba_common.h"
#include "imu_dual.h"

#define BB 48                 // block size: 8 cameras
#define BBS 49                // LDS row stride of a block

struct BigBufs {
    double* Ls;      // [n][n] factor panels (strictly-lower L, row-major); diagonal blocks hold the unit-lower L_JJ
    double* M;       // [BB][BB] inverse of the current diagonal block's L
    double* dv;      // [n] D
    double* yf;      // [n] D^-1 L^-1 g
    int* fail;       // [1]
    double* sep;     // [2][WB * WB + WB] two-sided banded factorisation: each side's Schur update of the separator block + right-hand side
    double* Ad;      // [n][6] banded factorisation: the damped matrix's entries inside the cameras' own 6 x 6 blocks, (hi, lo) at [hi][lo % 6]
};

__device__ __forceinline__ double rl64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// ------------------------------------------------------------------ prologue
__global__ __launch_bounds__(1024) void ba_big_prologue(BaDims d, BaBufs b, BaOpt opt, BigBufs g, int band)
{
    const int n = d.n, tid = threadIdx.x, nt = blockDim.x;
    __shared__ BaState st;
    __shared__ double red[16];
    if (tid == 0) { st = *b.st; *g.fail = 0; }
    __syncthreads();
    if (st.done) return;
    for (int i = tid; i < BA_NSLOT * BA_SLOT_STRIDE; i += nt) b.pt_scal[i] = 0.0;     // K8 of this iteration accumulates here
    // fold the BA_UREP replicas of the camera-side accumulators into replica 0
    for (size_t i = tid; i < b.cam_stride; i += nt) {
        double v = 0.0;
        for (int r = 0; r < BA_UREP; r++) v += b.rhs[(size_t)r * b.cam_stride + i];
        // U | gc are only accumulated on fresh iterations (K5 skips its first pass after a rejected step)
        if ((int)i >= n) { if (st.fresh) b.Ukeep[i - n] = v; else v = b.Ukeep[i - n]; }
        b.rhs[i] = v;
    }
    __syncthreads();
    // (1) fresh linearisation: cost at x, Jacobi scaling of the camera blocks, gradient test
    if (st.fresh) {
        if (tid < 64) {
            const double c = slot_sum(b.scal, 0);
            if (tid == 0) {
                st.x_cost = c;
                if (st.iter == 0) st.initial_cost = st.x_cost;
            }
        }
        if (!st.have_scale)
            for (int i = tid; i < n; i += nt) {
                const double h = b.U[(i / 6) * 36 + (i % 6) * 7];
                b.sc[i] = opt.jacobi ? 1.0 / (1.0 + sqrt(h)) : 1.0;
            }
        double gm = 0.0;
        for (int i = tid; i < n; i += nt) gm = fmax(gm, fabs(b.gc[i]));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) gm = fmax(gm, __shfl_down(gm, off, 64));
        if ((tid & 63) == 0) red[tid >> 6] = gm;
        __syncthreads();
        double gslots = 0.0;
        if (tid < 64) gslots = slot_max_all(b);
        if (tid == 0) {
            double gg = gslots;
            for (int w = 0; w < (nt + 63) / 64; w++) gg = fmax(gg, red[w]);
            if (!isfinite(st.x_cost)) { st.done = 1; st.termination = RS_BA_FAILURE; }
            else if (gg <= opt.gtol) { st.done = 1; st.termination = RS_BA_CONVERGENCE_GRADIENT; }
        }
        __syncthreads();
    }
    if (tid < 64) { const double f = slot_sum(b.scal, 1); if (f > 0.0 && tid == 0) *g.fail = 1; }   // K5 saw a bad landmark block
    if (tid == 0) *b.st = st;
    if (st.done) return;
    __syncthreads();
    // (2) damping, right-hand side, and the full symmetric matrix in place (S is accumulated in its upper triangle)
    const double radius = st.radius;
    for (int i = tid; i < n; i += nt) {
        const double h = b.U[(i / 6) * 36 + (i % 6) * 7];
        const double s2 = b.sc[i] * b.sc[i];
        const double l = clampd(s2 * h, opt.dmin, opt.dmax) / (radius * s2);
        const double yy = b.gc[i] + b.rhs[i];   // lam aliases rhs: same thread, same index
        b.rhs[i] = l;                           // lam
        b.dc[i] = yy;                           // y
    }
    __syncthreads();
    // the damped matrix inside the cameras' own blocks (S + U + damping), for the banded factorisation's single-load fetch
    if (band)
        for (int idx = tid; idx < 6 * n; idx += nt) {
            const int hi = idx / 6, lo = (hi / 6) * 6 + idx % 6;
            if (lo > hi) continue;
            double v = b.S[(size_t)lo * n + hi] + b.U[(hi / 6) * 36 + (lo % 6) * 6 + (hi % 6)];
            if (hi == lo) v += b.rhs[hi];
            g.Ad[idx] = v;
        }
}

// The lower triangle (incl. the diagonal) of the damped reduced matrix, in place: S is accumulated in its upper
// triangle; lower(i, j) = upper(j, i) + U (same camera) + lambda (diagonal).  The strict upper triangle is left
// as it is (nothing below reads it meaningfully), so there is no read/write overlap between threads.
__global__ __launch_bounds__(256) void ba_big_assemble(BaDims d, BaBufs b)
{
    if (b.st->done) return;
    const int n = d.n;
    const double* lam = b.rhs;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (size_t)n * n; idx += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx / n), j = (int)(idx % n);
        if (i < j) continue;
        double v = b.S[(size_t)j * n + i];
        if (i / 6 == j / 6) v += b.U[(i / 6) * 36 + (j % 6) * 6 + (i % 6)];     // U is stored upper: (row j%6, col i%6), j <= i
        if (i == j) v += lam[i];
        b.S[idx] = v;
    }
}

// ------------------------------------------------------------------ diagonal block
// (rcp_nr: ba_common.h — a full IEEE division is 40 instructions on the critical path of every column)
__global__ __launch_bounds__(256) void ba_big_diag(BaDims d, BaBufs b, BigBufs g, int J)
{
    if (b.st->done) return;
    // The block lives in LDS — rows 0..w-1, identity padding up to row 47, and the right-hand side's entries as row 48 —
    // and is factored as three 16-column sub-blocks, each in three phases (a column-by-column loop over all 48 columns
    // with two workgroup barriers per column cost 1 us per column):
    //   A  wave 0 factors the 16x16 diagonal sub-block with its rows in REGISTERS (lane = row, 16 fully unrolled
    //      column steps, pivots and column entries broadcast with v_readlane: no LDS, no barriers, no branches);
    //   B  waves 1-3, lane = a row below the sub-block (block rows and the right-hand side row): forward substitution
    //      against the sub-block, T = R L^-T (16 steps, L read as LDS broadcasts), multipliers P = T D^-1;
    //      wave 0 meanwhile inverts the sub-block's unit-lower L (row r of L^-1 by lane r, in registers) for the
    //      trailing-update kernel;
    //   C  all waves: the rest of the block  A[r][q] -= sum_j T[r][j] P[q][j].
    // Padding rows / columns are the identity, so no phase needs to know the block's real width.
    __shared__ double Lm[(BB + 1) * BBS];
    __shared__ double Mm[BB * BBS];
    __shared__ double Tm[(BB + 1) * 17];
    __shared__ double rdl[16];
    const int n = d.n, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c0 = BB * J, w = min(BB, n - c0);
    {
        // all loads of the block go out before the first one is used
        constexpr int ROUNDS = ((BB + 1) * BB + 255) / 256;
        double v[ROUNDS];
#pragma unroll
        for (int it = 0; it < ROUNDS; it++) {
            const int idx = tid + 256 * it, r = idx / BB, k = idx % BB;
            v[it] = (r == k) ? 1.0 : 0.0;
            if (idx < (BB + 1) * BB && k < w) { if (r < w) v[it] = b.S[(size_t)(c0 + r) * n + c0 + k]; else if (r == BB) v[it] = b.dc[c0 + k]; }
        }
#pragma unroll
        for (int it = 0; it < ROUNDS; it++) {
            const int idx = tid + 256 * it;
            if (idx < (BB + 1) * BB) Lm[(idx / BB) * BBS + idx % BB] = v[it];
            if (idx < BB * BB) Mm[(idx / BB) * BBS + idx % BB] = (idx / BB == idx % BB) ? 1.0 : 0.0;
        }
    }
    __syncthreads();
    bool bad = false;
    for (int c = 0; c < w; c += 16) {
        int r = lane & 15;
        double a[16];
        if (wave == 0) {                                                    // ---- A
#pragma unroll
            for (int k = 0; k < 16; k++) a[k] = Lm[(c + r) * BBS + c + k];
            double my_rd = 1.0, my_piv = 1.0;
#pragma unroll
            for (int cc = 0; cc < 16; cc++) {
                asm volatile("" : "+v"(r));      // per-step lane masks: hoisted, the 3 x 16 compare results lived in SGPR pairs and spilled
                const double piv = rl64(a[cc], cc);
                bad = bad || !(piv > 0.0) || !isfinite(piv);
                const double rd = rcp_nr(piv);
                const double lc = a[cc] * rd;                               // l_{r,cc} for r > cc
#pragma unroll
                for (int k = cc + 1; k < 16; k++) a[k] -= lc * rl64(a[cc], k);     // lane k holds a_{k,cc} = l_{k,cc} d_cc
                a[cc] = r > cc ? lc : a[cc];
                my_rd = r == cc ? rd : my_rd;
                my_piv = r == cc ? piv : my_piv;
                __builtin_amdgcn_sched_barrier(0);                          // keep the next steps' v_readlanes (SGPR pairs) from piling up
            }
            if (lane < 16) {
#pragma unroll
                for (int k = 0; k < 16; k++) Lm[(c + r) * BBS + c + k] = a[k];      // multipliers below the diagonal, d on it
                rdl[r] = my_rd;
                if (c + r < w) g.dv[c0 + c + r] = my_piv;
            }
        }
        __syncthreads();
        const int below = max(0, w - c - 16);                               // block rows under the sub-block
        const int nrows = below + 1;                                        // + the right-hand side row
        if (wave > 0) {                                                     // ---- B
            const int pr = tid - 64;
            if (pr < nrows) {
                const int row = pr < below ? c + 16 + pr : BB;
                double x[16];
#pragma unroll
                for (int jj = 0; jj < 16; jj++) {
                    double sacc = Lm[row * BBS + c + jj];
#pragma unroll
                    for (int k = 0; k < jj; k++) sacc -= x[k] * Lm[(c + jj) * BBS + c + k];
                    x[jj] = sacc;
                }
#pragma unroll
                for (int jj = 0; jj < 16; jj++) {
                    Tm[pr * 17 + jj] = x[jj];
                    Lm[row * BBS + c + jj] = x[jj] * rdl[jj];
                }
            }
        } else if (lane < 16) {                                             // ---- row r of L_ss^-1, L_ss still in a[]'s image in LDS
            double m[16];
#pragma unroll
            for (int k = 0; k < 16; k++) m[k] = (k == r) ? 1.0 : 0.0;
#pragma unroll
            for (int jj = 14; jj >= 0; jj--) {
                asm volatile("" : "+v"(r));
                double sacc = 0.0;
#pragma unroll
                for (int k = jj + 1; k < 16; k++) sacc -= m[k] * Lm[(c + k) * BBS + c + jj];   // m[k] = 0 beyond the row's diagonal
                m[jj] = jj < r ? sacc : m[jj];
            }
#pragma unroll
            for (int k = 0; k < 16; k++) Mm[(c + r) * BBS + c + k] = m[k];
        }
        __syncthreads();
        for (int idx = tid; idx < nrows * below; idx += 256) {              // ---- C
            const int pr = idx / below, qq = idx % below;
            const int row = pr < below ? c + 16 + pr : BB;
            if (pr < below && qq > pr) continue;                            // lower triangle (and the whole rhs row)
            double acc = 0.0;
#pragma unroll
            for (int jj = 0; jj < 16; jj++) acc += Tm[pr * 17 + jj] * Lm[(c + 16 + qq) * BBS + c + jj];
            Lm[row * BBS + c + 16 + qq] -= acc;
        }
        __syncthreads();
    }
    for (int idx = tid; idx < w * w; idx += 256) {
        const int r = idx / w, k = idx % w;
        g.Ls[(size_t)(c0 + r) * n + c0 + k] = k < r ? Lm[r * BBS + k] : (k == r ? 1.0 : 0.0);
    }
    if (tid < w) g.yf[c0 + tid] = Lm[BB * BBS + tid];       // D^-1 L^-1 g of this block
    if (__any(bad) && lane == 0) *g.fail = 1;
    // M: the inverses of the three 16x16 diagonal sub-blocks of L; the trailing-update kernel does the rest of the
    // triangular solve as a block forward substitution on the matrix cores.  Identity outside the sub-blocks.
    for (int idx = tid; idx < BB * BB; idx += 256) g.M[idx] = Mm[(idx / BB) * BBS + idx % BB];
}

// ------------------------------------------------------------------ trailing update
typedef __attribute__((ext_vector_type(4))) double d4;

// out = X Z^T for 48 x K operands held [row][k] (row stride BBS) in LDS, K = 4 * nchunk, on the matrix cores:
// v_mfma_f64_16x16x4_f64 takes A[lane&15][lane>>4] and B[lane&15][lane>>4] of a 16x4 slice, and returns
// C[row = (lane>>4) + 4 reg][col = lane & 15].  The 9 output tiles are dealt round-robin to the 4 waves;
// `store(R, C, v)` is called for every element this lane owns.
// K loop of NTL tiles of one wave, straight-line per tile count so that the operands of chunk kc + 1 are requested
// before the MFMAs of chunk kc issue (inside `if (ti < ntile)` branches every MFMA waited for its own ds_read)
template <int NTL>
__device__ __forceinline__ void gemm_nt_tiles(const double* const* xa, const double* const* zb, int nchunk, d4* acc)
{
    double a0[NTL], b0[NTL], a1[NTL], b1[NTL];
#pragma unroll
    for (int ti = 0; ti < NTL; ti++) { a0[ti] = xa[ti][0]; b0[ti] = zb[ti][0]; }
    for (int kc = 0; kc < nchunk; kc += 2) {
        const int k1 = 4 * min(kc + 1, nchunk - 1), k2 = 4 * min(kc + 2, nchunk - 1);
#pragma unroll
        for (int ti = 0; ti < NTL; ti++) { a1[ti] = xa[ti][k1]; b1[ti] = zb[ti][k1]; }
#pragma unroll
        for (int ti = 0; ti < NTL; ti++) acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[ti], b0[ti], acc[ti], 0, 0, 0);
#pragma unroll
        for (int ti = 0; ti < NTL; ti++) { a0[ti] = xa[ti][k2]; b0[ti] = zb[ti][k2]; }
        if (kc + 1 < nchunk) {
#pragma unroll
            for (int ti = 0; ti < NTL; ti++) acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[ti], b1[ti], acc[ti], 0, 0, 0);
        }
    }
}

template <typename F>
__device__ __forceinline__ void gemm_nt_48(const double* X, const double* Z, int nchunk, F store)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    // a wave's (up to three) tiles advance together: three independent accumulator chains instead of one 12-deep
    // chain after the other (an f64 MFMA is ~64 cycles of latency)
    d4 acc[3];
    const double *xa[3], *zb[3];
    int ntile = 0;
#pragma unroll
    for (int ti = 0; ti < 3; ti++) {
        const int t = wave + 4 * ti;                 // wave 0 owns three tiles, the others two
        acc[ti] = d4{0.0, 0.0, 0.0, 0.0};
        const int tt = t < 9 ? t : 0;
        xa[ti] = X + (16 * (tt / 3) + lr) * BBS + lk;
        zb[ti] = Z + (16 * (tt % 3) + lr) * BBS + lk;
        if (t < 9) ntile = ti + 1;
    }
    if (nchunk > 0) {
        if (ntile == 3) gemm_nt_tiles<3>(xa, zb, nchunk, acc);       // wave-uniform
        else gemm_nt_tiles<2>(xa, zb, nchunk, acc);
    }
#pragma unroll
    for (int ti = 0; ti < 3; ti++) {
        const int t = wave + 4 * ti;
        if (t >= 9) break;
#pragma unroll
        for (int reg = 0; reg < 4; reg++) store(16 * (t / 3) + lk + 4 * reg, 16 * (t % 3) + lr, acc[ti][reg]);
    }
}

__global__ __launch_bounds__(256) void ba_big_update(BaDims d, BaBufs b, BigBufs g, int J)
{
    if (b.st->done) return;
    extern __shared__ __attribute__((aligned(16))) double ulds[];
    double *Ai = ulds, *Ak = Ai + BB * BBS, *Mm = Ak + BB * BBS, *Pi = Mm + BB * BBS, *Pk = Pi + BB * BBS, *Lj = Pk + BB * BBS,
           *Rt = Lj + BB * BBS, *dvl = Rt + 2 * BB * 17;
    const int n = d.n, tid = threadIdx.x;
    const int NBLK = (n + BB - 1) / BB;
    const int c0 = BB * J, w = min(BB, n - c0);
    // decode (bi, bk): bk in (J, NBLK), bi in [bk, NBLK]; bi == NBLK is the right-hand side "row block"
    int bk = J + 1, rem = (int)blockIdx.x;
    while (rem >= NBLK - bk + 1) { rem -= NBLK - bk + 1; bk++; }
    const int bi = bk + rem;
    const bool rhs = bi == NBLK;
    const int ri0 = BB * bi, hi = rhs ? 1 : min(BB, n - ri0);
    const int rk0 = BB * bk, hk = min(BB, n - rk0);
    {
        // 9 rounds x 4 operands, branch-free: every load goes to a clamped (valid) address and is masked afterwards, so
        // all 36 are in flight together (conditional loads came out as one exec-masked block each)
        double vm[9], vl[9], vi[9], vk[9];
        const int rimax = rhs ? n - 1 : ri0 + hi - 1;
#pragma unroll
        for (int it = 0; it < 9; it++) {
            const int idx = tid + 256 * it, r = idx / BB, k = idx % BB;
            const int kc = c0 + min(k, w - 1);
            vm[it] = g.M[idx];
            vl[it] = g.Ls[(size_t)(c0 + min(r, w - 1)) * n + kc];
            vi[it] = b.S[(size_t)min(ri0 + r, rimax) * n + kc];
            vk[it] = b.S[(size_t)(rk0 + min(r, hk - 1)) * n + kc];
        }
#pragma unroll
        for (int it = 0; it < 9; it++) {
            const int idx = tid + 256 * it, r = idx / BB, k = idx % BB;
            Mm[r * BBS + k] = vm[it];
            Lj[r * BBS + k] = (r < w && k < w) ? vl[it] : (r == k ? 1.0 : 0.0);
            Ai[r * BBS + k] = (k < w && !rhs && r < hi) ? vi[it] : 0.0;
            Ak[r * BBS + k] = (k < w && r < hk) ? vk[it] : 0.0;
        }
    }
    if (tid < BB) dvl[tid] = tid < w ? g.dv[c0 + tid] : 1.0;
    __syncthreads();
    const int nchunk = (w + 3) / 4;            // operands are zero beyond w
    // Panels P = A L_JJ^-T D^-1 by block forward substitution over the three 16-column sub-blocks (M holds the
    // inverses of L's 16x16 diagonal sub-blocks):   R_c = A_c - sum_{e<c} T_e L_ce^T,   T_c = R_c M_cc^T,   P_c = T_c / d.
    // Both panels (rows of block bi and of block bk) go through the stages together: 6 row tiles over 4 waves.
    // The right-hand side's panel is yf, already final.
    {
        const int lane = tid & 63, wave = tid >> 6, lr = lane & 15, lk = lane >> 4;
        const bool two = !rhs && bi != bk;
        for (int c = 0; c < 3; c++) {
            if (16 * c >= w) break;                                        // wave-uniform: nothing but padding left
#pragma unroll
            for (int ti = 0; ti < 2; ti++) {
                const int t = wave + 4 * ti;
                if (t >= 6 || (t >= 3 && !two)) break;
                double* X = t < 3 ? Ak : Ai;
                double* R = Rt + (t < 3 ? 0 : BB * 17);
                const int tr = t % 3;
                d4 acc = {0.0, 0.0, 0.0, 0.0};
                for (int kc = 0; kc < 4 * c; kc++)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X[(16 * tr + lr) * BBS + 4 * kc + lk], Lj[(16 * c + lr) * BBS + 4 * kc + lk], acc, 0, 0, 0);
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const int rr = 16 * tr + lk + 4 * reg;
                    R[rr * 17 + lr] = X[rr * BBS + 16 * c + lr] - acc[reg];
                }
            }
            __syncthreads();
#pragma unroll
            for (int ti = 0; ti < 2; ti++) {
                const int t = wave + 4 * ti;
                if (t >= 6 || (t >= 3 && !two)) break;
                double* X = t < 3 ? Ak : Ai;
                double* P = t < 3 ? Pk : Pi;
                const double* R = Rt + (t < 3 ? 0 : BB * 17);
                const int tr = t % 3;
                d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kc = 0; kc < 4; kc++)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(R[(16 * tr + lr) * 17 + 4 * kc + lk], Mm[(16 * c + lr) * BBS + 16 * c + 4 * kc + lk], acc, 0, 0, 0);
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const int rr = 16 * tr + lk + 4 * reg;
                    X[rr * BBS + 16 * c + lr] = acc[reg];                  // T_c, read by the later sub-blocks
                    P[rr * BBS + 16 * c + lr] = acc[reg] / dvl[16 * c + lr];
                }
            }
            __syncthreads();
        }
        for (int idx = tid; idx < BB * BB; idx += 256) {                   // sub-blocks that are pure padding
            const int r = idx / BB, k = idx % BB;
            if (k >= 16 * ((w + 15) / 16)) { Pk[r * BBS + k] = 0.0; Pi[r * BBS + k] = 0.0; }
        }
        if (rhs && tid < BB) Pi[tid] = tid < w ? g.yf[c0 + tid] : 0.0;     // row 0 of Pi
    }
    __syncthreads();
    const double* PI = (bi == bk) ? Pk : Pi;
    if (bk == J + 1 && !rhs) {                 // this workgroup also stores the panel of row block bi
        for (int idx = tid; idx < BB * BB; idx += 256) {
            const int r = idx / BB, k = idx % BB;
            if (r < hi && k < w) g.Ls[(size_t)(ri0 + r) * n + c0 + k] = PI[r * BBS + k];
        }
    }
    // scale one operand by D, then C -= (P_i D) P_k^T
    __syncthreads();
    double* PD = Ai;                            // A_iJ is no longer needed
    for (int idx = tid; idx < BB * BB; idx += 256) {
        const int r = idx / BB, k = idx % BB;
        PD[r * BBS + k] = (rhs && r > 0) ? 0.0 : PI[r * BBS + k] * dvl[k];
    }
    __syncthreads();
    // results first into registers, then ALL global loads, then the stores (a load-modify-store per element
    // inside the tile loop would be a dozen dependent L2 round trips)
    double upd[12];
    int ur[12], uq[12];
    int cnt = 0;
#pragma unroll
    for (int u = 0; u < 12; u++) { upd[u] = 0.0; ur[u] = -1; uq[u] = 0; }
    gemm_nt_48(PD, Pk, nchunk, [&](int r, int q, double v) {
        const bool ok = q < hk && (rhs ? r == 0 : r < hi);
        upd[cnt] = v; ur[cnt] = ok ? r : -1; uq[cnt] = q;
        cnt++;
    });
    double old[12];
#pragma unroll
    for (int u = 0; u < 12; u++) {
        const int r = max(ur[u], 0), q = min(uq[u], hk - 1);
        old[u] = rhs ? b.dc[rk0 + q] : b.S[(size_t)(ri0 + min(r, hi - 1)) * n + rk0 + q];
    }
#pragma unroll
    for (int u = 0; u < 12; u++) {
        if (ur[u] < 0) continue;
        if (rhs) b.dc[rk0 + uq[u]] = old[u] - upd[u];
        else b.S[(size_t)(ri0 + ur[u]) * n + rk0 + uq[u]] = old[u] - upd[u];
    }
}

// ------------------------------------------------------------------ banded factorisation (one launch)
// A window whose landmarks are each seen by key frames at most `span` apart gives a BLOCK-BANDED reduced matrix: S(i, j) = 0
// for cameras more than `span` slots apart (a local map: every landmark lives for a few key frames; cfg 5: span 9, i.e. a
// bandwidth of 59 columns of the 588).  This is the sparsity the reference's SPARSE_SCHUR solve exploits through Eigen's sparse
// Cholesky (src/Optimization.cpp:360).  With 64-column blocks and a bandwidth of at most 64 columns only ONE sub-diagonal
// block per block column is non-zero and L D L^T creates no fill outside the band, so a workgroup factors block column after
// block column on a window in LDS —
//     D = A[J][J] (64 x 64),  P = A[J+1][J],  the right-hand side's entries of block J      (the 129-row panel)
//     T = A[J+1][J+1],  the right-hand side's entries of block J+1,  and the NEXT step's P
// — per block: the 129-row panel [D; P; right-hand side] is factored in place in eight 8-column steps (band_block8, below), which
// leaves L_JJ, D_J, the multipliers L_{J+1,J} and D^-1 L^-1 g; then T -= L_P D L_P^T on the matrix cores, T becomes the next D by
// a pointer swap.  The factor's columns go to memory step by step (the backward substitution reads them); the blocks of the next
// step (original entries of S: nothing outside the band ever updates them) are fetched by the tile waves, a few values per step.
// ba_band_factor runs as ONE workgroup over all block columns, or as TWO that eliminate from both ends of the band towards a
// separator block (see there), which ba_band_sep factors.  Round 4 (8-column chain with roles per wave, 512 threads): 158 -> 124 us
// per LM step at n = 588 against round 3's four 16-column sub-blocks per block with ONE wave factoring the 16 x 16 diagonal
// sub-block and its inverse by v_readlane broadcasts (6 us of the 8.7 us per sub-block).
#define WB 64
#define WBS 65
#define WROWS (2 * WB + 1)
// barrier for LDS traffic only: __syncthreads() also waits for every global access in flight (vmcnt), which would serialise
// the stores of the factor with the factorisation
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifndef BAND_DIAG
#define BAND_DIAG 0           // timing diagnostics (wrong results): 1 no factor stores, 2 the whole fetch at the end of the block, 4 no tile updates
#endif
// ---- the panel of one block column, 8 columns per step, with the chain kept off the matrix cores (the form of ba_solve.hip's
// K7 on a window in LDS).  The first 8 waves of the workgroup have ROLES (512 threads: 256 registers per lane, where 1024
// threads leave 128 and the chain's 36-entry triangle + row state spills; waves w, w + 4 share a SIMD; f64 MFMA and f64 VALU
// share one datapath per SIMD, so the chain's SIMD carries no tile wave):
//   chain waves `dw` and `dw ^ 4`   lane = panel row (the 64 rows of D / of P); the row's entries in the step's 8 columns live in
//                     registers.  Per step: the 8 x 8 diagonal sub-block goes through a 64-double scratch, EVERY lane factors
//                     it (L D L^T in registers, no cross-lane traffic), solves its own row t = a L^-T, l = t D^-1, writes l in
//                     place and t for the tile waves, then applies the step's rank-8 update to its entries in the NEXT 8
//                     columns itself — the chain never waits for the trailing update
//   wave 5            the right-hand side row: the same factorisation, y's row solve, and its whole rank-8 update (two columns
//                     per lane) every step; then it takes tiles as a sixth tile wave
//   tile waves        (5, on the other three SIMDs) the rank-8 update of the panel's remaining columns on the matrix cores,
//                     two 16 x 16 tiles per LDS round trip (operands and old values together, two MFMAs per tile), finished
//                     before the chain reads the step after next's raw columns; and, in the window where they would wait for
//                     the chain's factorisation, whatever the caller hands them for the PREVIOUS step's finished columns
//                     (after_step: the factor's columns to memory, the next blocks' fetch)
// two workgroup barriers per step.  Measured and dropped: tiles resident in registers as in K7, a pivot wave
// that factors the next sub-block one step ahead (a single wave's 8 x 8 factorisation takes as long as the chain's whole step),
// the chain waves on two SIMDs.  Rows above the step's sub-block are finished and idle; identity padding needs no masks.
#define BAND_TS 10
#ifndef BAND_CHAIN_X
#define BAND_CHAIN_X 4          // the chain waves are dw and dw ^ BAND_CHAIN_X: 1 = on two SIMDs (each has the FP64 pipe to itself in the
#endif                          // factorisation), 4 = on one SIMD (no tile wave's MFMAs next to the chain)
#define BAND_NT 5               // tile waves (waves beyond the first eight of a larger workgroup only keep the barriers)
#if BAND_CHAIN_X == 1
#define BAND_RHS_WAVE 2
#define band_is_tile(wave) ((wave) >= 3 && (wave) < 8)
#define band_tile_index(wave) ((wave) - 3)                                   // 0 .. 4
#else
#define BAND_RHS_WAVE 5         // (SIMD 1: its factorisation runs while the tile waves of that SIMD wait for the step's multipliers)
#define band_is_tile(wave) ((wave) < 8 && ((wave) & 3) != 0 && (wave) != BAND_RHS_WAVE)
#define band_tile_index(wave) ((wave) < 4 ? (wave) - 1 : (wave) - 3)         // 0 .. 4
#endif
// live tiles of a step by lo = (c + 8) / 16: tile rows lo .. 7 x tile columns lo .. min(row, 3), the triangle of the D rows first, then
// the P rows' rectangle; entry = 4 * tile row + tile column
__device__ static const unsigned char BAND_TILES[4][26] = {
    {0, 4, 5, 8, 9, 10, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31},
    {5, 9, 10, 13, 14, 15, 17, 18, 19, 21, 22, 23, 25, 26, 27, 29, 30, 31, 0, 0, 0, 0, 0, 0, 0, 0},
    {10, 14, 15, 18, 19, 22, 23, 26, 27, 30, 31, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {15, 19, 23, 27, 31, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}};
struct BandWin {
    double* Dm;    // [WB][WBS] D = A[J][J], factored in place (strictly lower: multipliers)
    double* Pp;    // [WB][WBS] P = A[J+1][J], becomes the multipliers L_{J+1,J}      (unused without P rows)
    double* y;     // [2 WB] the right-hand side's entries of blocks J and J+1; block J's become D^-1 L^-1 g
    double* Tt;    // [2 WB][BAND_TS] t = l D of the current step, by panel row
    double* scr;   // [8][8] the step's diagonal sub-block as the chain publishes it
    double* L2;    // [8][8] multipliers of the NEXT sub-block's rows (16-byte aligned: read as double2)
    double* dvl;   // [WB] pivots
};
template <typename Step>
static __device__ __forceinline__ void band_block8(const BandWin& W, const int s_first, const int dw, const bool has_p, bool& bad, Step&& after_step,
                                                   unsigned long long* stamps = nullptr)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#if RS_STAMPS
    // thread 0 (a chain wave): [0] wait for the sub-block, [1] factor + row solve, [2] wait for the step's multipliers, [3] fix-up;
    // thread 64 (a tile wave): [4] the two waits (and the caller's work between them), [5] update
    unsigned long long tq_ = wall_clock64();
#define BLOCK_STAMP(i) do { if (stamps && (tid & ~64) == 0) { const unsigned long long t_ = wall_clock64(); if ((i) < 4 ? tid == 0 : tid == 64) atomicAdd(stamps + (i), t_ - tq_); tq_ = t_; } } while (0)
#else
#define BLOCK_STAMP(i) do { } while (0)
#endif
    const int lr = lane & 15, lk = lane >> 4;
    const bool chain_d = wave == dw, chain_p = has_p && wave == (dw ^ BAND_CHAIN_X), rhs = wave == BAND_RHS_WAVE;
    const bool chain = chain_d || chain_p, tile = band_is_tile(wave);
    const int tw = band_tile_index(wave);
    double* const rp = chain_p ? W.Pp + lane * WBS : W.Dm + lane * WBS;     // chain: this lane's panel row
    const int irow = chain_p ? WB + lane : lane;
    double cur[8], t[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { cur[k] = 0.0; t[k] = 0.0; }
    if (chain) {
#pragma unroll
        for (int k = 0; k < 8; k++) cur[k] = rp[8 * s_first + k];
    } else if (rhs) {
#pragma unroll
        for (int k = 0; k < 8; k++) cur[k] = W.y[8 * s_first + k];
    }
    for (int s = s_first; s < WB / 8; s++) {
        const int c = 8 * s;
        if (chain_d && (unsigned)(lane - c) < 8u) {
#pragma unroll
            for (int k = 0; k < 8; k++) W.scr[(lane - c) * 8 + k] = cur[k];
        }
        lds_barrier();                                                       // the diagonal sub-block is published
        BLOCK_STAMP(0);
        if (chain || rhs) {
            // L D L^T of the sub-block, right-looking, interleaved with the lane's own row solve t = a L^-T (forward substitution
            // with the unit-lower block; row q of L is final when column q is eliminated, and dead afterwards), l = t D^-1
            const int m = irow - c;                  // < 0: finished row, 0 .. 7: a row of the sub-block itself, >= 8: below it
            double L[8][8], F[8], pv[8];
#pragma unroll
            for (int a = 0; a < 8; a++)
#pragma unroll
                for (int e = 0; e <= a; e++) L[a][e] = W.scr[a * 8 + e];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                double sacc = cur[q];
#pragma unroll
                for (int e = 0; e < q; e++) sacc -= t[e] * L[q][e];
                t[q] = (chain && q > m) ? 0.0 : sacc;                        // (row m of the sub-block: L is lower triangular)
                const double piv = L[q][q];
                bad = bad || !(piv > 0.0) || !isfinite(piv);
                const double rd = rcp_nr(piv);
                pv[q] = piv;
                F[q] = t[q] * rd;
                double lc[8];
#pragma unroll
                for (int a = q + 1; a < 8; a++) lc[a] = L[a][q] * rd;
#pragma unroll
                for (int a = q + 1; a < 8; a++)
#pragma unroll
                    for (int e = q + 1; e <= a; e++) L[a][e] -= lc[a] * L[e][q];
#pragma unroll
                for (int a = q + 1; a < 8; a++) L[a][q] = lc[a];
            }
            // raw entries of the next 8 columns: the tile waves have finished the previous step's update
            double nxt[8];
#pragma unroll
            for (int k = 0; k < 8; k++) nxt[k] = 0.0;
            if (chain && s + 1 < WB / 8) {
#pragma unroll
                for (int k = 0; k < 8; k++) nxt[k] = rp[c + 8 + k];
            }
            if (chain) {
                if (m >= 0) {
                    // (a row of the sub-block writes zeros above its diagonal: the upper triangle is never read)
                    double2* tt2 = reinterpret_cast<double2*>(W.Tt + irow * BAND_TS);
#pragma unroll
                    for (int k = 0; k < 8; k++) rp[c + k] = F[k];
#pragma unroll
                    for (int k = 0; k < 4; k++) tt2[k] = make_double2(t[2 * k], t[2 * k + 1]);
                    if ((unsigned)(m - 8) < 8u) {
                        double2* l2w = reinterpret_cast<double2*>(W.L2 + (m - 8) * 8);
#pragma unroll
                        for (int k = 0; k < 4; k++) l2w[k] = make_double2(F[2 * k], F[2 * k + 1]);
                    }
                    if (m == 0) {
#pragma unroll
                        for (int k = 0; k < 8; k++) W.dvl[c + k] = pv[k];
                    }
                }
#pragma unroll
                for (int k = 0; k < 8; k++) cur[k] = nxt[k];
            } else if (lane == 0) {
#pragma unroll
                for (int k = 0; k < 8; k++) W.y[c + k] = F[k];
            }
        } else if (s > s_first) {
            // the tile waves would wait for the step's multipliers: the caller's work on the PREVIOUS step's finished columns (its
            // factor columns to memory, the next blocks' fetch) runs here, off the chain's critical path
            after_step(s - 1, c - 8);
        }
        BLOCK_STAMP(1);
        lds_barrier();                                                       // multipliers and t of the step are in LDS
        BLOCK_STAMP(tid == 0 ? 2 : 4);
        if (chain) {
            if (s + 1 < WB / 8) {
                // this step's update of the next 8 columns of the lane's row: a[c+8+j] -= sum_k t_k l_{c+8+j, k}
                const double2* l2 = reinterpret_cast<const double2*>(W.L2);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const double2 a01 = l2[4 * j], a23 = l2[4 * j + 1], a45 = l2[4 * j + 2], a67 = l2[4 * j + 3];
                    cur[j] -= (t[0] * a01.x + t[1] * a01.y) + (t[2] * a23.x + t[3] * a23.y) + ((t[4] * a45.x + t[5] * a45.y) + (t[6] * a67.x + t[7] * a67.y));
                }
            }
        } else if (rhs) {
            // the right-hand side's columns lane (block J) and WB + lane (block J+1)
            if (lane >= c + 8) {
                const double* l = W.Dm + lane * WBS + c;
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < 8; k++) acc += t[k] * l[k];
                W.y[lane] -= acc;
            }
            if (has_p) {
                const double* l = W.Pp + lane * WBS + c;
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < 8; k++) acc += t[k] * l[k];
                W.y[WB + lane] -= acc;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // (one wave: its LDS accesses are in order)
            if (s + 1 < WB / 8) {
#pragma unroll
                for (int k = 0; k < 8; k++) cur[k] = W.y[c + 8 + k];
            }
        }
        if ((tile || rhs) && !(BAND_DIAG & 4)) {                                 // (the right-hand side's wave joins as sixth tile wave)
            // tile q of the step's list (BAND_TILES; without P rows only its triangle) -> tile wave q % 6
            const int pdiff = has_p ? (int)(W.Pp - W.Dm) : 0;                   // (P rows: the same LDS allocation as D)
            const int lo = (c + 8) >> 4, nlo = 4 - lo, tri = (nlo * (nlo + 1)) >> 1, ntile = tri + (has_p ? 4 * nlo : 0);
            // two tiles per round: the operands and old values of both travel together (one LDS round trip), their MFMAs interleave;
            // entries outside the live lower triangle are rewritten as they are: nobody writes them in this phase
            constexpr int NT6 = BAND_NT + 1;
#pragma unroll 1
            for (int q = rhs ? BAND_NT : tw; q < ntile; q += 2 * NT6) {
                const bool two = q + NT6 < ntile;                                // (wave-uniform; else the second tile repeats the first, unwritten)
                int tr[2], tc[2], pr0[2], qq[2];
                double x0[2], x1[2], z0[2], z1[2], old[2][4];
                double* e0[2];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int code = BAND_TILES[lo][(u && two) ? q + NT6 : q];
                    tr[u] = code >> 2; tc[u] = code & 3;
                    const double* X = W.Tt + (16 * tr[u] + lr) * BAND_TS + lk;
                    const double* Z = W.Dm + (16 * tc[u] + lr) * WBS + c + lk;
                    x0[u] = X[0]; x1[u] = X[4]; z0[u] = Z[0]; z1[u] = Z[4];
                    pr0[u] = 16 * tr[u] + lk; qq[u] = 16 * tc[u] + lr;
                    e0[u] = W.Dm + (tr[u] < 4 ? pr0[u] * WBS : pdiff + (pr0[u] - WB) * WBS) + qq[u];
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) old[u][reg] = e0[u][4 * reg * WBS];
                }
                d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[0], z0[0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[1], z0[1], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[0], z1[0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[1], z1[1], acc1, 0, 0, 0);
                // a tile below the diagonal and wholly behind column c + 8 needs no masks (most are)
                if (tr[0] > tc[0] && 16 * tc[0] >= c + 8 && tr[1] > tc[1] && 16 * tc[1] >= c + 8) {
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) e0[0][4 * reg * WBS] = old[0][reg] - acc0[reg];
                    if (two) {
#pragma unroll
                        for (int reg = 0; reg < 4; reg++) e0[1][4 * reg * WBS] = old[1][reg] - acc1[reg];
                    }
                } else {
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) {
                        const int pr = pr0[0] + 4 * reg;
                        e0[0][4 * reg * WBS] = (pr >= c + 8 && qq[0] >= c + 8 && qq[0] <= pr) ? old[0][reg] - acc0[reg] : old[0][reg];
                    }
                    if (two) {
#pragma unroll
                        for (int reg = 0; reg < 4; reg++) {
                            const int pr = pr0[1] + 4 * reg;
                            e0[1][4 * reg * WBS] = (pr >= c + 8 && qq[1] >= c + 8 && qq[1] <= pr) ? old[1][reg] - acc1[reg] : old[1][reg];
                        }
                    }
                }
            }
        }
        BLOCK_STAMP(tid == 0 ? 3 : 5);
    }
    if (!(chain || rhs)) after_step(WB / 8 - 1, WB - 8);
#undef BLOCK_STAMP
}

// Two-sided form (`split`): the band is cut at a SEPARATOR block column Js = (NB - 1) / 2.  Workgroup 0 eliminates the block
// columns above it in order, workgroup 1 the ones below it in REVERSE order (the same algorithm on the matrix with rows and
// columns reversed, v -> 64 NB - 1 - v: a band stays a band; the identity padding of the last real block becomes the head of
// its first block and whole padded sub-blocks are skipped); the bandwidth (<= 64) keeps the two sides uncoupled.  Each side
// leaves its Schur update of the separator block — side 0: A_ss - update, side 1: - update (in its reversed order) — in
// g.sep; ba_band_sep adds them and factors the separator, ba_big_finish substitutes backwards on both sides in lock step.  The
// chain of dependent 8-column steps is 34 + 8 instead of 74 at n = 588.  Factor blocks are stored at the REAL positions of their
// (virtual) rows and columns, so side 1's land in the upper triangle of g.Ls.
// doubles of LDS: four block buffers, the side's right-hand side (np + WB), the t panel, scratch, next multipliers, pivots
static inline size_t band_factor_lds_doubles(int n) { const size_t np = (size_t)((n + WB - 1) / WB) * WB; return (size_t)4 * WB * WBS + np + WB + 2 * WB * BAND_TS + 64 + 64 + WB; }
#define BAND_THREADS 512
__global__ __launch_bounds__(BAND_THREADS) void ba_band_factor(BaDims d, BaBufs b, BigBufs g, int split)
{
    if (b.st->done) return;
    extern __shared__ __attribute__((aligned(16))) double wl[];
    const int n = d.n, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lk = lane >> 4;
    const int NB = (n + WB - 1) / WB, np = NB * WB;
    // the window: D = A[J][J] and T = A[J+1][J+1] alternate between two buffers (T becomes the next D by a pointer swap), P =
    // A[J+1][J] and the NEXT block's P likewise
    double* Dm = wl;                               // [WB][WBS]
    double* Tm = Dm + WB * WBS;                    // [WB][WBS]
    double* Pp = Tm + WB * WBS;                    // [WB][WBS]
    double* Pn = Pp + WB * WBS;                    // [WB][WBS] A[J+2][J+1], fetched during block J
    double* yv = Pn + WB * WBS;                    // [np + WB] the right-hand side in this side's order; becomes D^-1 L^-1 g
    double* Tt = yv + np + WB;                     // [2 WB][BAND_TS]
    double* scr = Tt + 2 * WB * BAND_TS;           // [8][8]
    double* L2 = scr + 64;                         // [8][8]
    double* dvl = L2 + 64;                         // [WB]
    const int rev = split ? (int)blockIdx.x : 0;
    const int Js = (NB - 1) / 2;
    const int ND = split ? (rev ? NB - 1 - Js : Js) : NB;     // block columns this workgroup eliminates
    const int pad = np - n;                                     // identity padding: the tail of the real order = the head of the reversed one
    // virtual index -> real index; entries outside the matrix are identity
    auto phi = [&](int v) { return rev ? np - 1 - v : v; };
    auto inside = [&](int v) { return (unsigned)phi(v) < (unsigned)n; };
    // A'(r, k), clamped to the matrix (masked by the caller): ONE load — S's upper triangle, or the prologue's sum S + U + damping
    // inside a camera's own block
    auto entry = [&](int r, int k) {
        const int i = min(max(phi(r), 0), n - 1), j = min(max(phi(k), 0), n - 1), hi = max(i, j), lo = min(i, j);
        const double* p = hi / 6 == lo / 6 ? g.Ad + hi * 6 + lo % 6 : b.S + (size_t)lo * n + hi;
        return *p;
    };
#if RS_STAMPS
    unsigned long long tq = wall_clock64(), acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define BAND_STAMP(i) do { if (tid == 0) { const unsigned long long t_ = wall_clock64(); acc_t[i] += t_ - tq; tq = t_; } } while (0)
#else
#define BAND_STAMP(i) do { } while (0)
#endif
    if (tid < WB) dvl[tid] = 1.0;
    // block 0: D and P from memory (identity / zero outside the matrix); the right-hand side of every block.  Side 1 starts
    // the separator's entries from zero: it contributes updates only.
    for (int idx = tid; idx < WB * WB; idx += BAND_THREADS) {
        const int r = idx / WB, k = idx % WB;
        Dm[r * WBS + k] = (inside(r) && inside(k)) ? entry(r, k) : (r == k ? 1.0 : 0.0);
        Pp[r * WBS + k] = (1 < NB && inside(WB + r) && inside(k)) ? entry(WB + r, k) : 0.0;
    }
    for (int v = tid; v < np + WB; v += BAND_THREADS)
        yv[v] = (v < np && inside(v) && !(split && rev && v >= WB * ND)) ? b.dc[phi(v)] : 0.0;
    const bool tile = band_is_tile(wave);
    constexpr int NTT = 64 * BAND_NT;                                        // threads of the tile waves
    const int tt = band_tile_index(wave) * 64 + lane;                        // 0 .. NTT - 1
    int dw = 0;
    lds_barrier();
    BAND_STAMP(0);
    bool bad = false;
    for (int J = 0; J < ND; J++) {
        const int c0 = WB * J, r1 = c0 + WB, r2 = r1 + WB;
        const bool has_p = J + 1 < NB;
        const int hp = has_p ? (rev ? WB : min(WB, n - r1)) : 0;            // rows of block J+1
        const bool tz = split && rev && J + 1 == ND;                        // T is the separator's block seen from side 1
        // the tile waves fetch this step's T = A[J+1][J+1] (needed by the trailing update at the end of the block; its buffer
        // was the previous block's D) and the NEXT block's P = A[J+2][J+1] — original entries of S: nothing outside the band
        // ever updates them — requested behind the first step's update, placed behind the third's
        // in chunks of two values per lane, each requested behind one step's update and placed behind the next one's
        constexpr int CH = 4;                                                  // values of a chunk
        double fv[CH] = {0.0, 0.0, 0.0, 0.0};
        int chunk = 0;
        bool pending = false;
        auto request = [&](int ch) {                                         // (clamped addresses; masks when the values are placed)
#pragma unroll
            for (int u = 0; u < CH; u++) {
                const int idx = min(tt + NTT * (CH * ch + u), 2 * WB * WB - 1);
                fv[u] = entry(((idx >> 12) ? r2 : r1) + ((idx >> 6) & 63), r1 + (idx & 63));
            }
        };
        auto place = [&](int ch) {
#pragma unroll
            for (int u = 0; u < CH; u++) {
                const int idx = tt + NTT * (CH * ch + u), which = idx >> 12, r = (idx >> 6) & 63, k = idx & 63;
                if (idx >= 2 * WB * WB) continue;
                const bool vk = has_p && inside(r1 + k);
                if (which) Pn[r * WBS + k] = (J + 2 < NB && inside(r2 + r) && vk) ? fv[u] : 0.0;
                else Tm[r * WBS + k] = tz ? 0.0 : ((has_p && inside(r1 + r) && vk) ? fv[u] : (r == k ? 1.0 : 0.0));
            }
        };
        constexpr int NCHUNK = (2 * WB * WB + CH * NTT - 1) / (CH * NTT);       // 7: requested behind steps 0 .. 6
        // the factor's columns c .. c+7 -> memory (the backward substitution reads them): unit-lower L_JJ and L_{J+1,J}; element
        // e = row * 8 + column of the step's 128 x 8 values, two per tile thread (row and base address are the block's)
        constexpr int NST = (2 * WB * 8 + NTT - 1) / NTT;                    // 4
        int st_src[NST], st_row[NST];                                        // (offsets: D rows into Dm, P rows into Pp)
        long long st_dst[NST];
#pragma unroll
        for (int u = 0; u < NST; u++) {
            const int e = tt + NTT * u, i = e >> 3, kk = e & 7;
            const bool isd = i < WB, ok = e < 2 * WB * 8 && (isd ? inside(c0 + i) : (i - WB < hp && inside(r1 + i - WB)));
            st_row[u] = ok ? (isd ? i : WB) : -1;                             // D row i: columns <= i; P rows: every column
            st_src[u] = (isd ? i : i - WB) * WBS + kk;
            st_dst[u] = (long long)phi(isd ? c0 + i : r1 + i - WB) * n + phi(c0 + kk);
        }
        auto store_cols = [&](int c) {
#pragma unroll
            for (int u = 0; u < NST; u++) {
                const int col = c + ((tt + NTT * u) & 7);
                if (st_row[u] < 0 || col > st_row[u] || !inside(c0 + col)) continue;
                const double v = st_row[u] == WB ? Pp[st_src[u] + c] : Dm[st_src[u] + c];
                g.Ls[st_dst[u] + (rev ? -c : c)] = col < st_row[u] ? v : 1.0;
            }
        };
        const int s_first = (rev && J == 0) ? (pad >> 3) : 0;               // whole sub-blocks of side 1's padding are identity already
        const BandWin W = {Dm, Pp, yv + c0, Tt, scr, L2, dvl};
        band_block8(W, s_first, dw, has_p, bad, [&](int, int c) {
            if (!tile) return;
            // (the fetched values first: waiting for them behind this step's stores would wait for the stores as well)
#if !(BAND_DIAG & 2)
            if (pending) place(chunk - 1);
#endif
#if !(BAND_DIAG & 1)
            store_cols(c);
#endif
#if !(BAND_DIAG & 2)
            pending = chunk < NCHUNK;
            if (pending) request(chunk++);
#endif
        }, RS_STAMPS ? (unsigned long long*)b.dbg + (blockIdx.x == 0 ? 0 : 8) : nullptr);
        if (tile) {                                                             // (a short first block of side 1: the rest, waiting)
            if (pending) place(chunk - 1);
            for (; chunk < NCHUNK; chunk++) { request(chunk); place(chunk); }
        }
        BAND_STAMP(1);
        // D^-1 L^-1 g and D of the block
        if (wave == BAND_RHS_WAVE && inside(c0 + lane)) { g.yf[phi(c0 + lane)] = yv[c0 + lane]; g.dv[phi(c0 + lane)] = dvl[lane]; }
        if (!has_p) break;                                                       // last block
        // ---- trailing update on the matrix cores: T -= (L_P D) L_P^T (lower tiles); the right-hand side is up to date
        lds_barrier();                                                           // (T's last values have just been placed)
        for (int q = wave; q < 10; q += BAND_THREADS / 64) {                     // ten lower tiles of the 4 x 4
            const int tr = q < 1 ? 0 : q < 3 ? 1 : q < 6 ? 2 : 3, tc = q - (tr * (tr + 1)) / 2;
            d4 acc = {0.0, 0.0, 0.0, 0.0};
            const double* X = Pp + (16 * tr + lr) * WBS + lk;
            const double* Z = Pp + (16 * tc + lr) * WBS + lk;
#pragma unroll 4
            for (int kc = 0; kc < WB / 4; kc++)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X[4 * kc] * dvl[4 * kc + lk], Z[4 * kc], acc, 0, 0, 0);
#pragma unroll
            for (int reg = 0; reg < 4; reg++) Tm[(16 * tr + lk + 4 * reg) * WBS + 16 * tc + lr] -= acc[reg];
        }
        lds_barrier();
        BAND_STAMP(2);
        if (split && J == ND - 1) {
            // the separator's block as this side leaves it (lower triangle, this side's order) and its right-hand side
            double* sp = g.sep + (size_t)rev * (WB * WB + WB);
            for (int idx = tid; idx < WB * WB; idx += BAND_THREADS) {
                const int r = idx / WB, k = idx % WB;
                if (k <= r) sp[idx] = Tm[r * WBS + k];
            }
            if (tid < WB) sp[WB * WB + tid] = yv[r1 + tid];
            break;
        }
        // ---- shift the window by swapping buffers: T becomes D, the fetched P the panel's; the chain waves swap their rows
        { double* t_ = Dm; Dm = Tm; Tm = t_; t_ = Pp; Pp = Pn; Pn = t_; }
        dw ^= BAND_CHAIN_X;
    }
    if (__any(bad) && lane == 0) *g.fail = 1;
#if RS_STAMPS
    if (tid == 0) for (int q = 0; q < 8; q++) b.dbg[(blockIdx.x == 0 ? 16 : 40) + q] += acc_t[q];     // side 0: 16.., side 1: 40..
#endif
}

// Two-sided form, between the two sides' launch and ba_big_finish: ONE workgroup of BAND_THREADS adds the two sides' Schur
// updates of the separator block, factors it with the same 8-column chain (band_block8 without P rows: in ba_big_finish, whose
// 1024 threads leave 128 registers per lane, the chain spills) and leaves L_s (row-major [WB][WB], lower) and D^-1 L^-1 y_s in
// g.sep where side 0's contribution was.
static inline size_t band_sep_lds_doubles() { return (size_t)WB * WBS + 2 * WB + WB * BAND_TS + 64 + 64 + WB; }
__global__ __launch_bounds__(BAND_THREADS) void ba_band_sep(BaDims d, BaBufs b, BigBufs g)
{
    if (b.st->done || *g.fail) return;
    extern __shared__ __attribute__((aligned(16))) double wl[];
    const int tid = threadIdx.x, lane = tid & 63;
    double* Pm = wl;                                               // [WB][WBS] the separator's block
    double* ys = Pm + WB * WBS;                                    // [2 WB] its right-hand side (second half unused)
    double* Tt = ys + 2 * WB;                                      // [WB][BAND_TS]
    double* scr = Tt + WB * BAND_TS;
    double* L2 = scr + 64;
    double* dvl = L2 + 64;
    // lower triangle = side 0's block + side 1's update (transposed back from its reversed order)
    for (int idx = tid; idx < WB * WB; idx += BAND_THREADS) {
        const int r = idx / WB, k = idx % WB;
        const int lo = min(r, k), hi = max(r, k);                  // symmetric image; only hi >= lo is stored by the sides
        Pm[r * WBS + k] = g.sep[hi * WB + lo] + g.sep[WB * WB + WB + (WB - 1 - lo) * WB + (WB - 1 - hi)];
    }
    if (tid < WB) {
        ys[tid] = g.sep[WB * WB + tid] + g.sep[WB * WB + WB + WB * WB + (WB - 1 - tid)];
        dvl[tid] = 1.0;
    }
    lds_barrier();
    bool bad = false;
    const BandWin Ws = {Pm, nullptr, ys, Tt, scr, L2, dvl};
    band_block8(Ws, 0, 0, false, bad, [](int, int) {});
    if (__any(bad) && lane == 0) *g.fail = 1;
    lds_barrier();
    for (int idx = tid; idx < WB * WB; idx += BAND_THREADS) g.sep[idx] = Pm[(idx / WB) * WBS + idx % WB];
    if (tid < WB) g.sep[WB * WB + tid] = ys[tid];
}

// the band's backward substitution L^T x = D^-1 L^-1 g: per 64-column block, from the last,
//   v = yf_J - L_{J+1,J}^T x_{J+1}  (64 x 64, all threads),   L_JJ^T x_J = v  (one wave, the block's columns in registers)
static __device__ __forceinline__ void band_backsub(const BigBufs& g, int n, double* y, double* Lb)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    double* part = Lb + WB * WBS;                   // [16][WB] partial sums
    for (int i = tid; i < n; i += nt) y[i] = g.yf[i];
    const int NBLK = (n + WB - 1) / WB;
    __syncthreads();
    for (int J = NBLK - 1; J >= 0; J--) {
        const int c0 = WB * J, w = min(WB, n - c0);
        const int r1 = c0 + WB, hp = max(0, min(WB, n - r1));
        // diagonal block -> Lb; partial sums of L_P^T x_{J+1}: thread (k = column, grp = 16 slices of 4 rows)
        for (int idx = tid; idx < WB * WB; idx += nt) {
            const int r = idx / WB, k = idx % WB;
            Lb[r * WBS + k] = (r < w && k < w) ? g.Ls[(size_t)(c0 + r) * n + c0 + k] : 0.0;
        }
        {
            const int k = tid & 63, grp = tid >> 6;            // nt = 1024: 16 groups
            double acc = 0.0;
            if (k < w) {
                double l[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { const int r = 4 * grp + u; l[u] = r < hp ? g.Ls[(size_t)(r1 + r) * n + c0 + k] : 0.0; }
#pragma unroll
                for (int u = 0; u < 4; u++) { const int r = 4 * grp + u; acc += r < hp ? l[u] * y[r1 + r] : 0.0; }
            }
            part[grp * WB + k] = acc;
        }
        __syncthreads();
        if (tid < 64) {
            double v = 0.0;
            if (tid < w) {
                v = y[c0 + tid];
                for (int q = 0; q < 16; q++) v -= part[q * WB + tid];
            }
            double l[WB];
#pragma unroll
            for (int t = 0; t < WB; t++) l[t] = Lb[t * WBS + tid];            // column `tid` of the unit-lower block (t > tid is used)
#pragma unroll
            for (int t = WB - 1; t >= 0; t--) {
                const double xt = rl64(v, t);
                v -= (tid < t) ? l[t] * xt : 0.0;
            }
            if (tid < w) y[c0 + tid] = v;
        }
        __syncthreads();
    }
}

// Two-sided form: the separator's block (both sides' updates added) is factored here, then x_s, then both sides substitute
// backwards away from the separator in lock step — side 0 in the real order, side 1 in its reversed order (roles: below).
//   W: LDS window (the separator's factor and D^-1 L^-1 y_s from ba_band_sep); afterwards the same memory
//   serves as two diagonal blocks + two multiplier blocks [2][2][WB][WBS] + partial sums [2][7][WB].
static __device__ __forceinline__ void band_sep_backsub(const BigBufs& g, int n, double* y, double* W, int* s_fail)
{
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const int NB = (n + WB - 1) / WB, np = NB * WB, Js = (NB - 1) / 2;
    double* Pm = W;                                                // [WB][WBS] the separator's block, then its right-hand side row
    double* ys = Pm + WB * WBS;
    const int s0 = WB * Js;                                        // first real index of the separator
    // the separator's factor and D^-1 L^-1 y_s, as ba_band_sep left them
    for (int idx = tid; idx < WB * WB; idx += nt) Pm[(idx / WB) * WBS + idx % WB] = g.sep[idx];
    if (tid < WB) ys[tid] = g.sep[WB * WB + tid];
    for (int i = tid; i < n; i += nt) y[i] = g.yf[i];
    lds_barrier();
    // x_s = L_s^-T (D^-1 L^-1 y_s): one wave, the block's columns in registers
    if (tid < 64) {
        double v = ys[tid];
        double l[WB];
#pragma unroll
        for (int t = 0; t < WB; t++) l[t] = Pm[t * WBS + tid];
#pragma unroll
        for (int t = WB - 1; t >= 0; t--) {
            const double xt = rl64(v, t);
            v -= (tid < t) ? l[t] * xt : 0.0;
        }
        y[s0 + tid] = v;
    }
    __syncthreads();
    // both sides, away from the separator, in two roles with the same barrier sequence: wave 0 / wave 1 (different SIMDs)
    // run the triangular solve of side 0 / side 1 (the block's columns read from LDS sixteen at a time: sixty-four of them
    // in registers would be the whole register file of a wave at this workgroup size); the other 14 waves form the products
    // L_P^T x, and request the factor's blocks of step s + 1 (diagonal block and multipliers, 2 x 32 KB per side) at the
    // start of step s so that they travel under its arithmetic — the multipliers reach LDS in the solve phase (their buffer
    // is free: the products of step s are done), the diagonal block after the barrier that ends it.
    double* Lb = W;                                                // [2][WB][WBS] diagonal blocks
    double* Lp = Lb + 2 * WB * WBS;                                // [2][WB][WBS] multipliers L_{J+1,J}
    double* part = Lp + 2 * WB * WBS;                              // [2][7][WB]
    const int ND0 = Js, ND1 = NB - 1 - Js, nsteps = max(ND0, ND1);
    auto phi = [&](int sd, int v) { return sd ? np - 1 - v : v; };
    if (wave < 2) {
        const int sd = wave, ND = sd ? ND1 : ND0;
        __syncthreads();                                           // step 1's blocks are in LDS
        for (int step = 1; step <= nsteps; step++) {
            const int Jw = ND - step;
            __syncthreads();                                       // products of this step
            if (Jw >= 0) {
                const int i = phi(sd, WB * Jw + lane);
                double v = 0.0;
                if (i < n) {
                    v = y[i];
                    for (int q = 0; q < 7; q++) v -= part[(sd * 7 + q) * WB + lane];
                }
#pragma unroll
                for (int tc = WB - 16; tc >= 0; tc -= 16) {
                    double l[16];
#pragma unroll
                    for (int t = 0; t < 16; t++) l[t] = Lb[(sd * WB + tc + t) * WBS + lane];
#pragma unroll
                    for (int t = 15; t >= 0; t--) {
                        const double xt = rl64(v, tc + t);
                        v -= (lane < tc + t) ? l[t] * xt : 0.0;
                    }
                }
                if (i < n) y[i] = v;
            }
            __syncthreads();                                       // x of this block
        }
    } else {
        const int t = tid - 128;                                   // 0 .. 895
        double vv[19];
        // entry e of the 2 x 2 x 4096 values of a step: side, which block (0 diagonal, 1 multipliers), row, column
        auto request = [&](int step) {                             // (clamped addresses; masked when placed)
#pragma unroll
            for (int u = 0; u < 19; u++) {
                const int e = min(t + 896 * u, 4 * WB * WB - 1), sd = e >> 13, which = (e >> 12) & 1, r = (e >> 6) & 63, k = e & 63;
                const int J = max((sd ? ND1 : ND0) - step, 0), c0 = WB * J;
                const int i = min(phi(sd, (which ? c0 + WB : c0) + r), n - 1), j = min(phi(sd, c0 + k), n - 1);
                vv[u] = g.Ls[(size_t)i * n + j];
            }
        };
        auto place = [&](int step, int which_now) {
#pragma unroll
            for (int u = 0; u < 19; u++) {
                const int e = t + 896 * u, sd = e >> 13, which = (e >> 12) & 1, r = (e >> 6) & 63, k = e & 63;
                const int J = (sd ? ND1 : ND0) - step, c0 = WB * J;
                if (e >= 4 * WB * WB || which != which_now || J < 0) continue;
                const bool vk = phi(sd, c0 + k) < n;
                if (which) Lp[(sd * WB + r) * WBS + k] = vk ? vv[u] : 0.0;         // (rows of block J + 1 are inside the matrix)
                else Lb[(sd * WB + r) * WBS + k] = (vk && phi(sd, c0 + r) < n) ? vv[u] : 0.0;
            }
        };
        request(1);
        place(1, 0);
        place(1, 1);
        __syncthreads();
        for (int step = 1; step <= nsteps; step++) {
            if (step < nsteps) request(step + 1);
            {
                const int sd = t / 448, q = (t % 448) >> 6, k = t & 63;         // 2 sides x 7 row groups x 64 columns
                const int J = (sd ? ND1 : ND0) - step, r1 = WB * J + WB;
                if (J >= 0) {
                    double acc = 0.0;
                    for (int r = 10 * q; r < min(10 * q + 10, WB); r++) acc += Lp[(sd * WB + r) * WBS + k] * y[phi(sd, r1 + r)];
                    part[(sd * 7 + q) * WB + k] = acc;
                }
            }
            __syncthreads();
            if (step < nsteps) place(step + 1, 1);
            __syncthreads();
            if (step < nsteps) place(step + 1, 0);
        }
    }
    __syncthreads();
}

// block backward substitution L^T x = D^-1 L^-1 g by one workgroup: y (LDS, [n]) becomes x; Lb is an LDS block buffer
static __device__ __forceinline__ void big_backsub(const BigBufs& g, int n, double* y, double* Lb)
{
    // 1024 threads.  Per 48-column block J, from the last: (1) L_JJ^T x_J = y_J by one wave with the block's columns in
    // registers (lane = column, 48 unrolled v_readlane + fma steps); (2) y[0 : c0] -= L[block J rows, 0 : c0]^T x_J with
    // four threads per column (12 rows each, their loads in flight together), partial sums combined through LDS.
    // Lb: [BB][BBS] diagonal block, then [4][256] partial sums.
    const int tid = threadIdx.x, nt = blockDim.x;
    double* part = Lb + BB * BBS;
    for (int i = tid; i < n; i += nt) y[i] = g.yf[i];
    const int NBLK = (n + BB - 1) / BB;
    // diagonal block J of the factor -> registers (all loads issued together), registers -> Lb (zero padded)
    auto fetch = [&](int J, double v[3]) {
        const int c0 = BB * J, w = min(BB, n - c0);
#pragma unroll
        for (int it = 0; it < 3; it++) {
            const int idx = tid + nt * it, r = idx / BB, k = idx % BB;
            v[it] = (J >= 0 && r < w && k < w) ? g.Ls[(size_t)(c0 + r) * n + c0 + k] : 0.0;
        }
    };
    auto stash = [&](const double v[3]) {
#pragma unroll
        for (int it = 0; it < 3; it++) {
            const int idx = tid + nt * it;
            if (idx < BB * BB) Lb[(idx / BB) * BBS + idx % BB] = v[it];
        }
    };
    double nxt[3];
    fetch(NBLK - 1, nxt);
    stash(nxt);
    __syncthreads();
    for (int J = NBLK - 1; J >= 0; J--) {
        const int c0 = BB * J, w = min(BB, n - c0);
        fetch(J - 1, nxt);                                  // the next block's loads fly during this block's work
        if (tid < 64) {
            const int col = min(tid, BB - 1);
            double l[BB];
#pragma unroll
            for (int t = 0; t < BB; t++) l[t] = Lb[t * BBS + col];          // column `col` of the block (only t > col is used)
            double v = tid < w ? y[c0 + tid] : 0.0;
#pragma unroll
            for (int t = BB - 1; t >= 0; t--) {
                const double xt = rl64(v, t);
                v -= (tid < t) ? l[t] * xt : 0.0;
            }
            if (tid < w) y[c0 + tid] = v;
        }
        __syncthreads();
        // y[0 : c0] -= L[block J rows, 0 : c0]^T x_J
        for (int k0 = 0; k0 < c0; k0 += 256) {
            const int k = k0 + (tid & 255), grp = tid >> 8;
            double acc = 0.0;
            if (k < c0) {
                double l[12];
#pragma unroll
                for (int u = 0; u < 12; u++) l[u] = g.Ls[(size_t)(c0 + min(12 * grp + u, w - 1)) * n + k];
#pragma unroll
                for (int u = 0; u < 12; u++) acc += (12 * grp + u < w) ? l[u] * y[c0 + 12 * grp + u] : 0.0;
            }
            part[grp * 256 + (tid & 255)] = acc;
            __syncthreads();
            if (tid < 256 && k < c0) y[k] -= (part[tid] + part[256 + tid]) + (part[512 + tid] + part[768 + tid]);
            __syncthreads();
        }
        if (J > 0) stash(nxt);                              // Lb is free: the triangular solve above was its last reader
        __syncthreads();
    }
}

// ------------------------------------------------------------------ finish
__global__ __launch_bounds__(1024) void ba_big_finish(BaDims d, BaBufs b, BaOpt opt, BigBufs g, int band)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int n = d.n, tid = threadIdx.x, nt = blockDim.x;
    double* y = sm;                   // [n] becomes x
    double* Lb = y + n;               // [BB][BBS] current diagonal block
    __shared__ BaState st;
    __shared__ int s_fail;
    __shared__ double red3[16][3];
    if (tid == 0) { st = *b.st; s_fail = *g.fail; }
    __syncthreads();
    if (st.done) return;
    if (s_fail) {
        if (tid == 0) { st.solver_failed = 1; *b.st = st; }
        return;
    }
    if (band == 2) band_sep_backsub(g, n, y, Lb, &s_fail);
    else if (band) band_backsub(g, n, y, Lb);
    else big_backsub(g, n, y, Lb);
    // delta_c = -x, candidate cameras, camera part of the step scalars (as ba_solve.hip (5))
    const double* lam = b.rhs;
    double mcc = 0.0, ssq = 0.0, xsq = 0.0;
    bool bad = false;
    const double* Xc = b.Xc + (size_t)st.cur * d.C * 6;
    double* Xn = b.Xc + (size_t)(st.cur ^ 1) * d.C * 6;
    for (int c = tid; c < d.C; c += nt) {
        const int s = b.slot[c];
        bool active = false;
        if (s >= 0)
            for (int k = 0; k < 6; k++) active = active || b.U[s * 36 + k * 7] > 0.0;
        for (int k = 0; k < 6; k++) {
            const double x = Xc[6 * c + k];
            if (s >= 0) {
                const double dlt = -y[6 * s + k];
                if (!isfinite(dlt)) bad = true;
                mcc += 0.5 * (dlt * dlt * lam[6 * s + k] - dlt * b.gc[6 * s + k]);
                const double xn = x + dlt;
                if (active) { ssq += (x - xn) * (x - xn); xsq += x * x; }
                Xn[6 * c + k] = xn;
                b.dc[6 * s + k] = dlt;
            } else {
                Xn[6 * c + k] = x;
            }
        }
        cam_prepare(Xn + 6 * c, b.prep + ((size_t)(st.cur ^ 1) * d.C + c) * BA_PREP);
    }
    mcc = wave_sum(mcc); ssq = wave_sum(ssq); xsq = wave_sum(xsq);
    if (__any(bad) && (tid & 63) == 0) s_fail = 1;
    if ((tid & 63) == 0) { red3[tid >> 6][0] = mcc; red3[tid >> 6][1] = ssq; red3[tid >> 6][2] = xsq; }
    __syncthreads();
    if (tid == 0) {
        double a0 = 0, a1 = 0, a2 = 0;
        for (int w = 0; w < nt / 64; w++) { a0 += red3[w][0]; a1 += red3[w][1]; a2 += red3[w][2]; }
        st.cam_scal[0] = a0; st.cam_scal[1] = a1; st.cam_scal[2] = a2;
        st.solver_failed = s_fail;
        *b.st = st;
    }
}

// ------------------------------------------------------------------ host glue
// the factorisation proper: per 48-column block a diagonal-block launch and a trailing-update launch.  Works on
// d.n, b.S (matrix, lower triangle), b.dc (right-hand side) and the BigBufs; nothing in it knows what the unknowns are.
static void big_launch_factor(hipStream_t s, const BaDims& d, const BaBufs& b, const BigBufs& g, size_t lds_upd)
{
    const int NBLK = (d.n + BB - 1) / BB;
    for (int J = 0; J < NBLK; J++) {
        hipLaunchKernelGGL(ba_big_diag, dim3(1), dim3(256), 0, s, d, b, g, J);
        const int nb = NBLK - J - 1;                       // trailing column blocks
        const int pairs = nb * (nb + 1) / 2 + nb;          // (bi >= bk) pairs + one right-hand-side pair per bk
        if (pairs > 0) hipLaunchKernelGGL(ba_big_update, dim3(pairs), dim3(256), lds_upd, s, d, b, g, J);
    }
}

static void big_carve(char* ws, size_t n, BigBufs* g)
{
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t off = 0;
    g->Ls = (double*)(ws + off); off += al(sizeof(double) * n * n);
    g->M = (double*)(ws + off); off += al(sizeof(double) * BB * BB);
    g->dv = (double*)(ws + off); off += al(sizeof(double) * n);
    g->yf = (double*)(ws + off); off += al(sizeof(double) * n);
    g->fail = (int*)(ws + off); off += 256;
    g->sep = (double*)(ws + off); off += al(sizeof(double) * 2 * (64 * 64 + 64));
    g->Ad = (double*)(ws + off);
}

size_t ba_big_bytes(int n)
{
    return sizeof(double) * ((size_t)n * n + BB * BB + 2 * (size_t)n + 2 * (64 * 64 + 64) + 6 * (size_t)n) + 256 * 8;
}

// largest camera span (slots) whose block band fits the one-launch factorisation: 6 span + 5 <= WB columns
int ba_band_max_span() { return (WB - 5) / 6; }

int ba_launch_reduced_solve_big(rs_context* ctx, const BaDims& d, const BaBufs& b, const BaOpt& opt, char* ws, int band)
{
    const size_t n = (size_t)d.n;
    BigBufs g;
    big_carve(ws, n, &g);
    hipStream_t s = ctx->stream;
    if (band == 2 && (d.n + WB - 1) / WB < 3) band = 1;                  // no room for a separator between two sides
    if (band && sizeof(double) * band_factor_lds_doubles(d.n) > 160 * 1024) band = 0;       // (n > 2300: the general blocked form)
    // y, diagonal block, backsub partial sums; two-sided band: y + the panel window of the separator's factorisation
    // (two-sided band: the panel window of the separator's factorisation, then two diagonal blocks + two multiplier blocks + partial sums)
    const size_t lds_fin = sizeof(double) * (n + (band == 2 ? (size_t)4 * WB * WBS + 1024 : (size_t)WB * WBS + 1024));
    if (lds_fin > 48 * 1024)
        RS_HIP(ctx, rs_lds_attr((const void*)ba_big_finish, lds_fin));
    const size_t lds_upd = sizeof(double) * (6 * BB * BBS + 2 * BB * 17 + BB);
    RS_HIP(ctx, rs_lds_attr((const void*)ba_big_update, lds_upd));
    hipLaunchKernelGGL(ba_big_prologue, dim3(1), dim3(1024), 0, s, d, b, opt, g, band);
    if (!band) hipLaunchKernelGGL(ba_big_assemble, dim3(256), dim3(256), 0, s, d, b);
    if (band) {
        const size_t lds_band = sizeof(double) * band_factor_lds_doubles(d.n);
        RS_HIP(ctx, rs_lds_attr((const void*)ba_band_factor, lds_band));
        {
            rs_prof_scope ps(ctx, "K7b_band_factor");
            hipLaunchKernelGGL(ba_band_factor, dim3(band == 2 ? 2 : 1), dim3(BAND_THREADS), lds_band, s, d, b, g, band == 2 ? 1 : 0);
        }
        if (band == 2) {
            rs_prof_scope ps(ctx, "K7c_band_separator");
            hipLaunchKernelGGL(ba_band_sep, dim3(1), dim3(BAND_THREADS), sizeof(double) * band_sep_lds_doubles(), s, d, b, g);
        }
    } else {
        big_launch_factor(s, d, b, g, lds_upd);
    }
    hipLaunchKernelGGL(ba_big_finish, dim3(1), dim3(1024), lds_fin, s, d, b, opt, g, band);
    return RS_OK;
}

// =============================================================================== inertial reduced solve
// bundle_adjust with IMU factor pairs (reference src/Optimization.cpp:317-346): the camera side of the problem has,
// besides the 6-unknown pose blocks, a velocity (3) and a bias (6) block per frame the factors touch.  The landmark
// side is untouched (K5 / K8 as in the vision-only solve; the factors involve no point), so only this reduced solve
// differs:  N = 6 Cf + 9 Ci unknowns,  A = [U + S_schur on the pose part] + J_imu^T J_imu + Lambda,  y = g_total,
// factorised by the same blocked L D L^T (diag / update launches above).
//   ba_imu_prologue (1 WG)  replica fold, linearisation of every factor (one thread per factor, dual numbers),
//                           total cost / gradient / Jacobi scale / damping, assembly of A and y
//   ba_imu_finish   (1 WG)  backward substitution, pose / velocity / bias step and candidates, step scalars, cost of
//                           the inertial blocks at the candidate (-> BaState::cam_scal[3], added to K8's cost)
// Ceres semantics restated in oracle/ba.c (extra residual blocks): Jacobi scale and LM diagonal per column from the
// TOTAL J^T J diagonal (reprojection + inertial), gradient tolerance on the total gradient, x-norm over all blocks.
__device__ __forceinline__ void imu_columns(const BaDims& d, const BaBufs& b, int i, int j, int col[IMU_NP])
{
    const int n6 = d.n, qi = b.imu.inert_slot[i], qj = b.imu.inert_slot[j], si = b.slot[i], sj = b.slot[j];
    for (int k = 0; k < 6; k++) { col[k] = 6 * si + k; col[9 + k] = n6 + 9 * qi + 3 + k; col[15 + k] = 6 * sj + k; }
    for (int k = 0; k < 3; k++) { col[6 + k] = n6 + 9 * qi + k; col[21 + k] = n6 + 9 * qj + k; }
}

__global__ __launch_bounds__(512) void ba_imu_prologue(BaDims d, BaBufs b, BaOpt opt, BigBufs g)
{
    const int n6 = d.n, N = b.imu.N, tid = threadIdx.x, nt = blockDim.x;
    __shared__ BaState st;
    __shared__ double red[16];
    __shared__ double s_cost;
    if (tid == 0) { st = *b.st; *g.fail = 0; s_cost = 0.0; }
    __syncthreads();
    if (st.done) return;
    for (int i = tid; i < BA_NSLOT * BA_SLOT_STRIDE; i += nt) b.pt_scal[i] = 0.0;     // K8 of this round accumulates here
    for (size_t i = tid; i < b.cam_stride; i += nt) {                                  // fold the accumulator replicas
        double v = 0.0;
        for (int r = 0; r < BA_UREP; r++) v += b.rhs[(size_t)r * b.cam_stride + i];
        if ((int)i >= n6) { if (st.fresh) b.Ukeep[i - n6] = v; else v = b.Ukeep[i - n6]; }
        b.rhs[i] = v;
    }
    for (size_t i = tid; i < (size_t)N * N; i += nt) b.imu.A[i] = 0.0;
    for (int i = tid; i < N; i += nt) b.imu.gtot[i] = 0.0;
    __syncthreads();
    // ---- linearise the factors at x: J^T J into the lower triangle of A, J^T r into gtot, 1/2 |r|^2 into the cost
    const double* Xc = b.Xc + (size_t)st.cur * d.C * 6;
    const double* Xv = b.imu.Xv + (size_t)st.cur * d.C * 9;
    // One 32-lane group per factor pair, lane k = local parameter k (24 of them): the functor runs on DualLane (value +
    // this lane's partial), so the lane ends up with its COLUMN of the whitened 9 x 24 Jacobian in nine registers.  J^T J:
    // lane k fetches the other columns with shuffles and adds its row of the 24 x 24 block.  (One thread per factor with
    // all 24 partials and ~2800 serial atomics took 1.4 ms per launch.)
    {
        const int grp = tid >> 5, lk = tid & 31;
        for (int fi = grp; fi < b.imu.n_fac; fi += nt >> 5) {
            const ImuFactorDev& F = b.imu.fac[fi];
            const int i = F.f.cam_i, j = F.f.cam_j;
            double r[9], jl[9], rw[6], is[2];
            imu_preintegration_lanes(F, b.imu.gravity, Xc + 6 * i, Xv + 9 * i, Xv + 9 * i + 3, Xc + 6 * j, Xv + 9 * j, r, jl);
            imu_bias_walk(F.f, Xv + 9 * i + 3, Xv + 9 * j + 3, rw, is);
            int col[IMU_NP];
            imu_columns(d, b, i, j, col);
            int mycol = 0;
#pragma unroll
            for (int k = 0; k < IMU_NP; k++) mycol = (lk == k) ? col[k] : mycol;
            double gk = 0.0;
#pragma unroll
            for (int a = 0; a < 9; a++) gk += jl[a] * r[a];
            if (lk < IMU_NP) atomicAdd(&b.imu.gtot[mycol], gk);
#pragma unroll
            for (int l = 0; l < IMU_NP; l++) {
                double h = 0.0;
#pragma unroll
                for (int a = 0; a < 9; a++) h += jl[a] * __shfl(jl[a], l, 32);
                if (lk < IMU_NP && mycol >= col[l]) atomicAdd(&b.imu.A[(size_t)mycol * N + col[l]], h);
            }
            if (lk < 6) {                                   // bias walk: -1/sigma on bias_i[a], +1/sigma on bias_j[a]
                const int a = lk;
                const double sg = is[a / 3], s2 = sg * sg;
                const int ci = n6 + 9 * b.imu.inert_slot[i] + 3 + a, cj = n6 + 9 * b.imu.inert_slot[j] + 3 + a;
                atomicAdd(&b.imu.gtot[ci], -sg * rw[a]);
                atomicAdd(&b.imu.gtot[cj], sg * rw[a]);
                atomicAdd(&b.imu.A[(size_t)ci * N + ci], s2);
                atomicAdd(&b.imu.A[(size_t)cj * N + cj], s2);
                atomicAdd(&b.imu.A[(size_t)max(ci, cj) * N + min(ci, cj)], -s2);
            }
            if (lk == 0) {
                double c = 0.0;
                for (int a = 0; a < 9; a++) c += 0.5 * r[a] * r[a];
                for (int a = 0; a < 6; a++) c += 0.5 * rw[a] * rw[a];
                atomicAdd(&s_cost, c);
            }
        }
    }
    __syncthreads();
    // ---- totals per column: diagonal of J^T J, gradient; Jacobi scale (first linearisation), damping, right-hand side
    double gm = 0.0;
    for (int i = tid; i < N; i += nt) {
        const double h = b.imu.A[(size_t)i * N + i] + (i < n6 ? b.U[(i / 6) * 36 + (i % 6) * 7] : 0.0);
        const double gi = b.imu.gtot[i] + (i < n6 ? b.gc[i] : 0.0);
        double sc = b.imu.sc[i];
        if (!st.have_scale) { sc = opt.jacobi ? 1.0 / (1.0 + sqrt(h)) : 1.0; b.imu.sc[i] = sc; }
        const double s2 = sc * sc;
        b.imu.lam[i] = clampd(s2 * h, opt.dmin, opt.dmax) / (st.radius * s2);
        b.imu.gtot[i] = gi;
        b.imu.yv[i] = gi + (i < n6 ? b.rhs[i] : 0.0);
        gm = fmax(gm, fabs(gi));
    }
    if (st.fresh) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) gm = fmax(gm, __shfl_down(gm, off, 64));
        if ((tid & 63) == 0) red[tid >> 6] = gm;
        __syncthreads();
        double cslots = 0.0, gslots = 0.0;
        if (tid < 64) { cslots = slot_sum(b.scal, 0); gslots = slot_max_all(b); }
        if (tid == 0) {
            st.x_cost = cslots + s_cost;
            if (st.iter == 0) st.initial_cost = st.x_cost;
            double gg = gslots;
            for (int w = 0; w < (nt + 63) / 64; w++) gg = fmax(gg, red[w]);
            if (!isfinite(st.x_cost)) { st.done = 1; st.termination = RS_BA_FAILURE; }
            else if (gg <= opt.gtol) { st.done = 1; st.termination = RS_BA_CONVERGENCE_GRADIENT; }
        }
    }
    __syncthreads();
    if (tid < 64) { const double f = slot_sum(b.scal, 1); if (f > 0.0 && tid == 0) *g.fail = 1; }   // K5 saw a bad landmark block
    if (tid == 0) *b.st = st;
    if (st.done) return;
    __syncthreads();
}

// the lower triangle of the damped matrix: inertial part (already there) + U + S_schur (pose part) + Lambda.  A grid of
// its own: one workgroup streaming the N x N matrix with a load in flight per thread took 130 us of the prologue.
__global__ __launch_bounds__(256) void ba_imu_assemble(BaDims d, BaBufs b)
{
    if (b.st->done) return;
    const int n6 = d.n, N = b.imu.N;
    const int i = blockIdx.x, tid = threadIdx.x;             // one row per workgroup
    for (int j = tid; j <= i; j += blockDim.x) {
        const size_t idx = (size_t)i * N + j;
        double v = b.imu.A[idx];
        if (i < n6) {                                       // j <= i < n6: both are pose columns
            v += b.S[(size_t)j * n6 + i];                   // K5 accumulates S in its upper triangle
            if (i / 6 == j / 6) v += b.U[(i / 6) * 36 + (j % 6) * 6 + (i % 6)];
        }
        if (i == j) v += b.imu.lam[i];
        b.imu.A[idx] = v;
    }
}

__global__ __launch_bounds__(1024) void ba_imu_finish(BaDims d, BaBufs b, BaOpt opt, BigBufs g)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int n6 = d.n, N = b.imu.N, tid = threadIdx.x, nt = blockDim.x;
    double* y = sm;                   // [N] becomes x
    double* Lb = y + N;               // [BB][BBS]
    __shared__ BaState st;
    __shared__ int s_fail;
    __shared__ double red3[16][3];
    __shared__ double s_cand;
    if (tid == 0) { st = *b.st; s_fail = *g.fail; s_cand = 0.0; }
    __syncthreads();
    if (st.done) return;
    if (s_fail) {
        if (tid == 0) { st.solver_failed = 1; *b.st = st; }
        return;
    }
    big_backsub(g, N, y, Lb);
    double mcc = 0.0, ssq = 0.0, xsq = 0.0;
    bool bad = false;
    const double* Xc = b.Xc + (size_t)st.cur * d.C * 6;
    double* Xn = b.Xc + (size_t)(st.cur ^ 1) * d.C * 6;
    const double* Xv = b.imu.Xv + (size_t)st.cur * d.C * 9;
    double* Xvn = b.imu.Xv + (size_t)(st.cur ^ 1) * d.C * 9;
    for (int c = tid; c < d.C; c += nt) {
        const int s = b.slot[c], q = b.imu.inert_slot[c];
        bool active = q >= 0;                               // a frame with an inertial block is in the problem
        if (s >= 0)
            for (int k = 0; k < 6; k++) active = active || b.U[s * 36 + k * 7] > 0.0;
        for (int k = 0; k < 6; k++) {
            const double x = Xc[6 * c + k];
            if (s >= 0) {
                const double dlt = -y[6 * s + k];
                if (!isfinite(dlt)) bad = true;
                mcc += 0.5 * (dlt * dlt * b.imu.lam[6 * s + k] - dlt * b.imu.gtot[6 * s + k]);
                const double xn = x + dlt;
                if (active) { ssq += (x - xn) * (x - xn); xsq += x * x; }
                Xn[6 * c + k] = xn;
                b.dc[6 * s + k] = dlt;                       // K8 back-substitutes the points with the pose step
            } else {
                Xn[6 * c + k] = x;
            }
        }
        for (int k = 0; k < 9; k++) {
            const double x = Xv[9 * c + k];
            if (q >= 0) {
                const int col = n6 + 9 * q + k;
                const double dlt = -y[col];
                if (!isfinite(dlt)) bad = true;
                mcc += 0.5 * (dlt * dlt * b.imu.lam[col] - dlt * b.imu.gtot[col]);
                const double xn = x + dlt;
                ssq += (x - xn) * (x - xn); xsq += x * x;
                Xvn[9 * c + k] = xn;
            } else {
                Xvn[9 * c + k] = x;
            }
        }
        cam_prepare(Xn + 6 * c, b.prep + ((size_t)(st.cur ^ 1) * d.C + c) * BA_PREP);
    }
    __syncthreads();                                         // candidates written (same workgroup reads them below)
    for (int fi = tid; fi < b.imu.n_fac; fi += nt) {        // cost of the inertial blocks at the candidate (values only)
        const ImuFactorDev& F = b.imu.fac[fi];
        const int i = F.f.cam_i, j = F.f.cam_j;
        double r[9], rw[6], is[2], c = 0.0;
        imu_preintegration(F, b.imu.gravity, Xn + 6 * i, Xvn + 9 * i, Xvn + 9 * i + 3, Xn + 6 * j, Xvn + 9 * j, r, nullptr);
        imu_bias_walk(F.f, Xvn + 9 * i + 3, Xvn + 9 * j + 3, rw, is);
        for (int a = 0; a < 9; a++) c += 0.5 * r[a] * r[a];
        for (int a = 0; a < 6; a++) c += 0.5 * rw[a] * rw[a];
        atomicAdd(&s_cand, c);
    }
    mcc = wave_sum(mcc); ssq = wave_sum(ssq); xsq = wave_sum(xsq);
    if (__any(bad) && (tid & 63) == 0) s_fail = 1;
    if ((tid & 63) == 0) { red3[tid >> 6][0] = mcc; red3[tid >> 6][1] = ssq; red3[tid >> 6][2] = xsq; }
    __syncthreads();
    if (tid == 0) {
        double a0 = 0, a1 = 0, a2 = 0;
        for (int w = 0; w < nt / 64; w++) { a0 += red3[w][0]; a1 += red3[w][1]; a2 += red3[w][2]; }
        st.cam_scal[0] = a0; st.cam_scal[1] = a1; st.cam_scal[2] = a2; st.cam_scal[3] = s_cand;
        st.solver_failed = s_fail;
        *b.st = st;
    }
}

size_t ba_inertial_bytes(int N, int n_fac, int C)
{
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    return ba_big_bytes(N) + al(sizeof(double) * (size_t)N * N) + 4 * al(sizeof(double) * ((size_t)N + 1)) +
           al(sizeof(double) * (size_t)(n_fac + 1) * 9 * IMU_NP) + al(sizeof(ImuFactorDev) * (size_t)(n_fac + 1)) +
           al(sizeof(int32_t) * (size_t)(C + 1)) + al(sizeof(double) * (BA_MAXSETS + 1) * 9 * (size_t)(C + 1)) + 256;      // Xv: one buffer per state slot
}

// carves the inertial buffers out of `ws` (after the BigBufs region); returns the device addresses the host uploads to
void ba_inertial_carve(char* ws, int N, int n_fac, int C, BaImu* imu, ImuFactorDev** d_fac, int32_t** d_inert)
{
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t off = ba_big_bytes(N);
    off = al(off);
    imu->A = (double*)(ws + off); off += al(sizeof(double) * (size_t)N * N);
    imu->yv = (double*)(ws + off); off += al(sizeof(double) * ((size_t)N + 1));
    imu->lam = (double*)(ws + off); off += al(sizeof(double) * ((size_t)N + 1));
    imu->sc = (double*)(ws + off); off += al(sizeof(double) * ((size_t)N + 1));
    imu->gtot = (double*)(ws + off); off += al(sizeof(double) * ((size_t)N + 1));
    imu->Jf = (double*)(ws + off); off += al(sizeof(double) * (size_t)(n_fac + 1) * 9 * IMU_NP);
    *d_fac = (ImuFactorDev*)(ws + off); off += al(sizeof(ImuFactorDev) * (size_t)(n_fac + 1));
    *d_inert = (int32_t*)(ws + off); off += al(sizeof(int32_t) * (size_t)(C + 1));
    imu->Xv = (double*)(ws + off);
    imu->fac = *d_fac;
    imu->inert_slot = *d_inert;
}

int ba_launch_reduced_solve_inertial(rs_context* ctx, const BaDims& d, const BaBufs& b, const BaOpt& opt, char* ws)
{
    const int N = b.imu.N;
    BigBufs g;
    big_carve(ws, (size_t)N, &g);
    hipStream_t s = ctx->stream;
    const size_t lds_fin = sizeof(double) * ((size_t)N + BB * BBS + 1024);
    if (lds_fin > 48 * 1024)
        RS_HIP(ctx, rs_lds_attr((const void*)ba_imu_finish, lds_fin));
    const size_t lds_upd = sizeof(double) * (6 * BB * BBS + 2 * BB * 17 + BB);
    RS_HIP(ctx, rs_lds_attr((const void*)ba_big_update, lds_upd));
    hipLaunchKernelGGL(ba_imu_prologue, dim3(1), dim3(512), 0, s, d, b, opt, g);
    hipLaunchKernelGGL(ba_imu_assemble, dim3(N), dim3(256), 0, s, d, b);
    // the factorisation runs on the N x N system: same kernels, their view of (n, matrix, right-hand side) swapped
    BaDims dN = d;
    dN.n = N;
    BaBufs bN = b;
    bN.S = b.imu.A;
    bN.dc = b.imu.yv;
    big_launch_factor(s, dN, bN, g, lds_upd);
    hipLaunchKernelGGL(ba_imu_finish, dim3(1), dim3(1024), lds_fin, s, d, b, opt, g);
    return RS_OK;
}
