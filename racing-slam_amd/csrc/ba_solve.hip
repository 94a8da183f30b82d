// ba_solve.hip — K7: the reduced camera system of the local-window BA.
//
// Solves (U + Lambda_c - sum Y Y^T) y = g~ for the <= 21 free cameras of a
// local window (n = 6 Cf <= 128) in ONE workgroup with the matrix resident in
// LDS.  This is the step Ceres hands to a sparse Cholesky after Schur
// elimination (reference src/Optimization.cpp:360, SPARSE_SCHUR); at n ~ 108
// the matrix is dense and 93 KB, so it lives in the CU's 160 KB LDS.
//
// Algorithm: right-looking block Cholesky with 6x6 blocks (one camera per
// block column), 3 workgroup barriers per block column:
//   (a) wave 0 factors the 6x6 diagonal block, one lane per row, pivots
//       exchanged with v_readlane;
//   (b) one thread per row below solves its 1x6 panel row against L_JJ;
//   (c) a 16x16 thread grid applies the rank-6 update to the trailing lower
//       triangle, each thread keeping the 6-wide panel rows it needs in
//       registers.
// The right-hand side rides along as an extra matrix ROW (row n), so the
// forward substitution L y = g~ falls out of the factorisation; only the
// backward substitution runs afterwards (2 barriers per block).  Row stride is
// odd (n+1) so that column walks hit distinct LDS banks.
#include "ba_common.h"

#define K7_THREADS 256
#define K7_MAXB 8      // ceil(129 / 16)

__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xFFFFFFFFll), lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// 1/sqrt(x) from v_rsq_f64 + two Newton steps (quadratic convergence from ~2^-27)
__device__ __forceinline__ double fast_rsqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    y = y * (1.5 - h * y * y);
    y = y * (1.5 - h * y * y);
    return y;
}

__global__ __launch_bounds__(K7_THREADS) void ba_reduced_solve_lds(BaDims d, BaBufs b, BaOpt opt)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int n = d.n, tid = threadIdx.x, nt = K7_THREADS;
    const int LD = n + 1 + ((n & 1) ? 1 : 0);     // odd row stride; (n is a multiple of 6)
    double* A = sm;                                // (n+1) x LD, lower triangle + rhs row n
    double* lam = A + (size_t)(n + 1) * LD;        // [n] camera damping
    double* invd = lam + n;                        // [n] 1 / L_ii
    double* xs = invd + n;                         // [n] solution
    double* Minv = xs + n;                         // [n/6][6][6] inverses of the diagonal blocks of L
    __shared__ BaState st;
    __shared__ int s_fail;
    __shared__ double red[4];
    __shared__ double red3[4][3];
    if (tid == 0) { st = *b.st; s_fail = 0; }
    __syncthreads();
    if (st.done) return;

    // (1) fresh linearisation: cost at x, Jacobi scaling of the camera blocks, gradient test
    if (st.fresh) {
        if (tid == 0) {
            st.x_cost = b.scal[0];
            if (st.iter == 0) st.initial_cost = st.x_cost;
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
        if (tid == 0) {
            double g = __longlong_as_double((long long)*(unsigned long long*)b.gmax);
            for (int w = 0; w < nt / 64; w++) g = fmax(g, red[w]);
            if (!isfinite(st.x_cost)) { st.done = 1; st.termination = RS_BA_FAILURE; }
            else if (g <= opt.gtol) { st.done = 1; st.termination = RS_BA_CONVERGENCE_GRADIENT; }
        }
        __syncthreads();
        if (st.done) { if (tid == 0) *b.st = st; return; }
    }

    // (2) assemble the lower triangle: A[c][r] = S_upper[r][c] + U + Lambda_c;  A[n][k] = gc + rhs
    const double radius = st.radius;
    for (int i = tid; i < n; i += nt) {
        const double h = b.U[(i / 6) * 36 + (i % 6) * 7];
        const double s2 = b.sc[i] * b.sc[i];
        lam[i] = clampd(s2 * h, opt.dmin, opt.dmax) / (radius * s2);
        A[(size_t)n * LD + i] = b.gc[i] + b.rhs[i];
    }
    __syncthreads();
    for (int idx = tid; idx < n * n; idx += nt) {
        const int r = idx / n, c = idx - r * n;
        if (c < r) continue;
        double v = b.S[idx];                        // upper entry (r, c), coalesced along c
        if (r / 6 == c / 6) {
            v += b.U[(r / 6) * 36 + (r % 6) * 6 + (c % 6)];
            if (r == c) v += lam[r];
        }
        A[(size_t)c * LD + r] = v;
    }
    __syncthreads();

    // (3) block Cholesky, rhs as row n
    const int NB = n / 6;
    const int ty = tid >> 4, tx = tid & 15;
    for (int J = 0; J < NB; J++) {
        const int c0 = 6 * J;
        // (a) diagonal block, wave 0, lane a owns row a
        if (tid < 64) {
            const int a = tid < 6 ? tid : 5;
            double D[6];
#pragma unroll
            for (int e = 0; e < 6; e++) D[e] = (e <= a) ? A[(size_t)(c0 + a) * LD + c0 + e] : 0.0;
            bool bad = false;
#pragma unroll
            for (int c = 0; c < 6; c++) {
                const double piv = readlane_f64(D[c], c);
                if (!(piv > 0.0) || !isfinite(piv)) bad = true;
                const double rs = fast_rsqrt(piv);
                const double lac = D[c] * rs;          // lanes a >= c: L[a][c]; lane c: sqrt(piv)
                D[c] = lac;
                if (tid == c) invd[c0 + c] = rs;
#pragma unroll
                for (int e = c + 1; e < 6; e++) {
                    const double lec = readlane_f64(lac, e);
                    D[e] -= lac * lec;
                }
            }
            if (tid < 6) {
#pragma unroll
                for (int e = 0; e < 6; e++)
                    if (e <= a) A[(size_t)(c0 + a) * LD + c0 + e] = D[e];
            }
            if (bad && tid == 0) s_fail = 1;
        }
        __syncthreads();
        // (b') lanes 250..255 (never own a panel row for n <= 126): column e of L_JJ^-1
        if (tid >= 250) {
            const int e = tid - 250;
            double m[6];
#pragma unroll
            for (int a = 0; a < 6; a++) {
                double sacc = (a == e) ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < a; k++) sacc -= A[(size_t)(c0 + a) * LD + c0 + k] * ((k >= e) ? m[k] : 0.0);
                m[a] = (a >= e) ? sacc * invd[c0 + a] : 0.0;
                Minv[J * 36 + a * 6 + e] = m[a];
            }
        }
        // (b) panel rows i > c0+5 (incl. the rhs row n): row_i <- row_i * L_JJ^-T
        {
            double Lb[21], iv[6];
#pragma unroll
            for (int r = 0, q = 0; r < 6; r++) {
                iv[r] = invd[c0 + r];
#pragma unroll
                for (int e = 0; e < r; e++) Lb[q++] = A[(size_t)(c0 + r) * LD + c0 + e];
                q += 0;
            }
            for (int i = c0 + 6 + tid; i <= n; i += nt) {
                double* row = A + (size_t)i * LD + c0;
                double x[6];
#pragma unroll
                for (int r = 0, q = 0; r < 6; r++) {
                    double s = row[r];
#pragma unroll
                    for (int e = 0; e < r; e++) s -= Lb[q++] * x[e];
                    x[r] = s * iv[r];
                }
#pragma unroll
                for (int r = 0; r < 6; r++) row[r] = x[r];
            }
        }
        __syncthreads();
        // (c) trailing update A[i][k] -= sum_e A[i][c0+e] A[k][c0+e],  c0+6 <= k <= i <= n (k < n)
        {
            const int r0 = c0 + 6;
            if (r0 <= n) {
                double Li[K7_MAXB][6];
#pragma unroll
                for (int u = 0; u < K7_MAXB; u++) {
                    const int i = r0 + ty + 16 * u;
#pragma unroll
                    for (int e = 0; e < 6; e++) Li[u][e] = (i <= n) ? A[(size_t)i * LD + c0 + e] : 0.0;
                }
#pragma unroll
                for (int v = 0; v < K7_MAXB; v++) {
                    const int k = r0 + tx + 16 * v;
                    if (k >= n) continue;
                    double Lk[6];
#pragma unroll
                    for (int e = 0; e < 6; e++) Lk[e] = A[(size_t)k * LD + c0 + e];
#pragma unroll
                    for (int u = 0; u < K7_MAXB; u++) {
                        const int i = r0 + ty + 16 * u;
                        if (i > n || i < k) continue;
                        double s = 0.0;
#pragma unroll
                        for (int e = 0; e < 6; e++) s += Li[u][e] * Lk[e];
                        A[(size_t)i * LD + k] -= s;
                    }
                }
            }
        }
        __syncthreads();
    }
    if (b.scal[1] > 0.0 && tid == 0) s_fail = 1;
    __syncthreads();
    if (s_fail) {
        if (tid == 0) { st.solver_failed = 1; *b.st = st; }
        return;
    }
    // (4) backward substitution L^T x = y (y = row n), block column by block column
    double* y = A + (size_t)n * LD;
    for (int J = NB - 1; J >= 0; J--) {
        const int c0 = 6 * J;
        if (tid < 6) {      // x_J = L_JJ^-T y_J as a mat-vec with the stored inverse block
            double sacc = 0.0;
#pragma unroll
            for (int e = 0; e < 6; e++) sacc += Minv[J * 36 + e * 6 + tid] * y[c0 + e];
            xs[c0 + tid] = sacc;
        }
        __syncthreads();
        for (int i = tid; i < c0; i += nt) {
            double s = y[i];
#pragma unroll
            for (int e = 0; e < 6; e++) s -= A[(size_t)(c0 + e) * LD + i] * xs[c0 + e];
            y[i] = s;
        }
        __syncthreads();
    }
    // (5) delta_c = -x, candidate cameras, camera part of the step scalars
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
                const double dlt = -xs[6 * s + k];
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

size_t ba_reduced_solve_lds_bytes(int n)
{
    const int LD = n + 1 + ((n & 1) ? 1 : 0);
    return sizeof(double) * ((size_t)(n + 1) * LD + 3 * (size_t)n + 6 * (size_t)n + 8);
}

void ba_launch_reduced_solve_lds(hipStream_t s, const BaDims& d, const BaBufs& b, const BaOpt& opt)
{
    const size_t lds = ba_reduced_solve_lds_bytes(d.n);
    hipLaunchKernelGGL(ba_reduced_solve_lds, dim3(1), dim3(K7_THREADS), lds, s, d, b, opt);
}

int ba_prepare_reduced_solve_lds(int n)
{
    const size_t lds = ba_reduced_solve_lds_bytes(n);
    return (int)hipFuncSetAttribute((const void*)ba_reduced_solve_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}
