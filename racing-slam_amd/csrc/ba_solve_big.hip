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

#define BB 48                 // block size: 8 cameras
#define BBS 49                // LDS row stride of a block

struct BigBufs {
    double* Ls;      // [n][n] factor panels (strictly-lower L, row-major); diagonal blocks hold the unit-lower L_JJ
    double* M;       // [BB][BB] inverse of the current diagonal block's L
    double* dv;      // [n] D
    double* yf;      // [n] D^-1 L^-1 g
    int* fail;       // [1]
};

__device__ __forceinline__ double rl64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// ------------------------------------------------------------------ prologue
__global__ __launch_bounds__(1024) void ba_big_prologue(BaDims d, BaBufs b, BaOpt opt, BigBufs g)
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
        if (tid < 64) gslots = slot_max_bits(b.gmax);
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
    const double* lam = b.rhs;
    for (int idx = tid; idx < n * n; idx += nt) {
        const int i = idx / n, j = idx % n;
        if (i > j) b.S[idx] = b.S[(size_t)j * n + i];
    }
    __syncthreads();
    for (int idx = tid; idx < n * n; idx += nt) {
        const int i = idx / n, j = idx % n;
        if (i / 6 != j / 6) continue;
        const int a = i % 6, e = j % 6;
        double v = b.S[idx] + ((a <= e) ? b.U[(i / 6) * 36 + a * 6 + e] : b.U[(i / 6) * 36 + e * 6 + a]);
        if (i == j) v += lam[i];
        b.S[idx] = v;
    }
}

// ------------------------------------------------------------------ diagonal block
__global__ __launch_bounds__(64) void ba_big_diag(BaDims d, BaBufs b, BigBufs g, int J)
{
    if (b.st->done) return;
    const int n = d.n, lane = threadIdx.x;
    const int c0 = BB * J, w = min(BB, n - c0);
    // lane < w: row c0 + lane of the block; lanes w..47: identity padding; lane 63: the right-hand side's entries
    // of this block (a row below all others)
    double a[BB], m[BB];
#pragma unroll
    for (int k = 0; k < BB; k++) {
        a[k] = 0.0;
        if (k < w) {
            if (lane < w) a[k] = b.S[(size_t)(c0 + lane) * n + c0 + k];
            else if (lane == 63) a[k] = b.dc[c0 + k];
        }
        if (k >= w && lane == k) a[k] = 1.0;      // pad: identity rows keep the pivots of unused columns at 1, everything finite
    }
    bool bad = false;
    double dpiv[BB];
#pragma unroll
    for (int c = 0; c < BB; c++) {
        const double piv = rl64(a[c], c);
        dpiv[c] = piv;
        if (c < w && (!(piv > 0.0) || !isfinite(piv))) bad = true;
        const double rd = 1.0 / piv;
        const double lc = a[c] * rd;                       // rows below c; the others compute unused values
#pragma unroll
        for (int k = c + 1; k < BB; k++) {
            const double akc = rl64(a[c], k);              // row k, column c, still l_kc * d_c
            a[k] -= lc * akc;
        }
        if (lane > c) a[c] = lc;
    }
    // M = L^-1 (unit lower), row `lane`: M L = I  =>  m[j] = -sum_{k > j} m[k] L[k][j]
#pragma unroll
    for (int j = 0; j < BB; j++) m[j] = (j == lane) ? 1.0 : 0.0;
#pragma unroll
    for (int j = BB - 2; j >= 0; j--) {
        double sacc = m[j];
#pragma unroll
        for (int k = j + 1; k < BB; k++) sacc -= m[k] * rl64(a[j], k);      // L[k][j] lives in row k's a[j]
        m[j] = lane > j ? sacc : m[j];
    }
    if (lane < w) {
#pragma unroll
        for (int k = 0; k < BB; k++) {
            if (k < w) g.Ls[(size_t)(c0 + lane) * n + c0 + k] = k < lane ? a[k] : (k == lane ? 1.0 : 0.0);
            g.M[lane * BB + k] = m[k];
        }
    } else if (lane < BB) {
#pragma unroll
        for (int k = 0; k < BB; k++) g.M[lane * BB + k] = (k == lane) ? 1.0 : 0.0;
    }
    if (lane == 63) {
#pragma unroll
        for (int k = 0; k < BB; k++) if (k < w) g.yf[c0 + k] = a[k];       // D^-1 L^-1 g of this block
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < BB; k++) if (k < w) g.dv[c0 + k] = dpiv[k];
    }
    if (__any(bad) && lane == 0) *g.fail = 1;
}

// ------------------------------------------------------------------ trailing update
// X Z^T for 48x48 operands held [row][k] in LDS: thread (ty, tx) owns outputs (ty + 16 a, tx + 16 b)
__device__ __forceinline__ void gemm_nt_48(const double* X, const double* Z, int w, double out[3][3])
{
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int c = 0; c < 3; c++) out[a][c] = 0.0;
    for (int k = 0; k < w; k++) {
        double x[3], z[3];
#pragma unroll
        for (int a = 0; a < 3; a++) { x[a] = X[(ty + 16 * a) * BBS + k]; z[a] = Z[(tx + 16 * a) * BBS + k]; }
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int c = 0; c < 3; c++) out[a][c] += x[a] * z[c];
    }
}

__global__ __launch_bounds__(256) void ba_big_update(BaDims d, BaBufs b, BigBufs g, int J)
{
    if (b.st->done) return;
    extern __shared__ __attribute__((aligned(16))) double ulds[];
    double *Ai = ulds, *Ak = Ai + BB * BBS, *Mm = Ak + BB * BBS, *Pi = Mm + BB * BBS, *Pk = Pi + BB * BBS, *dvl = Pk + BB * BBS;
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
    for (int idx = tid; idx < BB * BB; idx += 256) {
        const int r = idx / BB, k = idx % BB;
        Mm[r * BBS + k] = g.M[idx];
        double vi = 0.0, vk = 0.0;
        if (k < w) {
            if (!rhs && r < hi) vi = b.S[(size_t)(ri0 + r) * n + c0 + k];
            if (r < hk) vk = b.S[(size_t)(rk0 + r) * n + c0 + k];
        }
        Ai[r * BBS + k] = vi;
        Ak[r * BBS + k] = vk;
    }
    if (tid < BB) dvl[tid] = tid < w ? g.dv[c0 + tid] : 1.0;
    __syncthreads();
    const int ty = tid >> 4, tx = tid & 15;
    double acc[3][3];
    // P_k = A_kJ M^T D^-1  (and P_i likewise; the right-hand side's panel is yf, already final)
    gemm_nt_48(Ak, Mm, w, acc);
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int c = 0; c < 3; c++) Pk[(ty + 16 * a) * BBS + tx + 16 * c] = acc[a][c] / dvl[tx + 16 * c];
    if (rhs) {
        if (tid < BB) Pi[tid] = tid < w ? g.yf[c0 + tid] : 0.0;             // row 0 of Pi
    } else if (bi != bk) {
        gemm_nt_48(Ai, Mm, w, acc);
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int c = 0; c < 3; c++) Pi[(ty + 16 * a) * BBS + tx + 16 * c] = acc[a][c] / dvl[tx + 16 * c];
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
    gemm_nt_48(PD, Pk, w, acc);
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int r = ty + 16 * a, q = tx + 16 * c;
            if (q >= hk) continue;
            if (rhs) { if (r == 0) b.dc[rk0 + q] -= acc[a][c]; }
            else if (r < hi) b.S[(size_t)(ri0 + r) * n + rk0 + q] -= acc[a][c];
        }
}

// ------------------------------------------------------------------ finish
__global__ __launch_bounds__(1024) void ba_big_finish(BaDims d, BaBufs b, BaOpt opt, BigBufs g)
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
    for (int i = tid; i < n; i += nt) y[i] = g.yf[i];
    const int NBLK = (n + BB - 1) / BB;
    for (int J = NBLK - 1; J >= 0; J--) {
        const int c0 = BB * J, w = min(BB, n - c0);
        for (int idx = tid; idx < w * w; idx += nt) { const int r = idx / w, k = idx % w; Lb[r * BBS + k] = g.Ls[(size_t)(c0 + r) * n + c0 + k]; }
        __syncthreads();
        // L_JJ^T x_J = y_J: one wave, column by column from the last
        if (tid < 64) {
            double v = tid < w ? y[c0 + tid] : 0.0;
            for (int t = w - 1; t >= 0; t--) {
                const double xt = rl64(v, t);
                if (tid < t) v -= Lb[t * BBS + tid] * xt;
            }
            if (tid < w) y[c0 + tid] = v;
        }
        __syncthreads();
        // y[0 : c0] -= L[block J rows, 0 : c0]^T x_J
        for (int k = tid; k < c0; k += nt) {
            double acc = 0.0;
            for (int r = 0; r < w; r++) acc += g.Ls[(size_t)(c0 + r) * n + k] * y[c0 + r];
            y[k] -= acc;
        }
        __syncthreads();
    }
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
size_t ba_big_bytes(int n)
{
    return sizeof(double) * ((size_t)n * n + BB * BB + 2 * (size_t)n) + 256 * 5;
}

int ba_launch_reduced_solve_big(rs_context* ctx, const BaDims& d, const BaBufs& b, const BaOpt& opt, char* ws)
{
    const size_t n = (size_t)d.n;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    BigBufs g;
    size_t off = 0;
    g.Ls = (double*)(ws + off); off += al(sizeof(double) * n * n);
    g.M = (double*)(ws + off); off += al(sizeof(double) * BB * BB);
    g.dv = (double*)(ws + off); off += al(sizeof(double) * n);
    g.yf = (double*)(ws + off); off += al(sizeof(double) * n);
    g.fail = (int*)(ws + off);
    hipStream_t s = ctx->stream;
    const int NBLK = (d.n + BB - 1) / BB;
    const size_t lds_fin = sizeof(double) * (n + BB * BBS);
    if (lds_fin > 48 * 1024)
        RS_HIP(ctx, hipFuncSetAttribute((const void*)ba_big_finish, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fin));
    const size_t lds_upd = sizeof(double) * (5 * BB * BBS + BB);
    RS_HIP(ctx, hipFuncSetAttribute((const void*)ba_big_update, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_upd));
    hipLaunchKernelGGL(ba_big_prologue, dim3(1), dim3(1024), 0, s, d, b, opt, g);
    for (int J = 0; J < NBLK; J++) {
        hipLaunchKernelGGL(ba_big_diag, dim3(1), dim3(64), 0, s, d, b, g, J);
        const int nb = NBLK - J - 1;                       // trailing column blocks
        const int pairs = nb * (nb + 1) / 2 + nb;          // (bi >= bk) pairs + one right-hand-side pair per bk
        if (pairs > 0) hipLaunchKernelGGL(ba_big_update, dim3(pairs), dim3(256), lds_upd, s, d, b, g, J);
    }
    hipLaunchKernelGGL(ba_big_finish, dim3(1), dim3(1024), lds_fin, s, d, b, opt, g);
    return RS_OK;
}
