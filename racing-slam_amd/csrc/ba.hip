// ba.hip — local-window bundle adjustment (Levenberg-Marquardt with point-block
// Schur elimination) and pose-only refinement, all in f64, with the whole LM
// schedule resident on the device (no host round trip until the summary is
// read): host orchestration, K0 init, K10 finalize, the generic (any size)
// fallback kernels and refine_pose.  The fast paths live in ba_schur.hip (K5),
// ba_solve.hip (K7) and ba_update.hip (K8).
//
// Replaces optimization::bundle_adjust / refine_pose's ceres::Solve
// (reference src/Optimization.cpp:21-72,127-142,194-267,269-374).  The Ceres
// trust-region schedule restated here is documented in oracle/ba.c and
// SURVEY.md §8 a11; this file works in UNSCALED parameters, which is
// algebraically identical to Ceres' Jacobi-scaled solve:
//     (H + Lambda) delta = -g,   Lambda_i = clamp(s_i^2 H_ii, 1e-6, 1e32) / (radius s_i^2),
//     s_i = 1 / (1 + sqrt(H_ii at the first linearisation)),
//     model_cost_change = 1/2 (delta' Lambda delta - delta' g).
//
// Kernel chain per LM iteration (three launches, all unconditional; each kernel
// returns at once when state.done is set):
//   K5 linearise + Schur   applies the accept / reject decision of the PREVIOUS
//                          iteration first (every workgroup redundantly, from the
//                          previous state block and slot sums: no "decide" launch),
//                          then analytic Jacobians (left Jacobian of SO(3), never
//                          stored), Huber weights, V/gp, U/gc, damped V^-1, Schur
//                          products into S and the reduced rhs
//   [RCCL all-reduce of the accumulators when landmark-sharded]
//   K7 reduced solve       one workgroup: (U + Lambda_c - sum Y Y') y = rhs by block
//                          L D L', delta_c, candidate cameras
//   K8 back-substitution   delta_p, candidate points, model-cost terms, robust cost
//                          at the candidate; clears the accumulators for the next K5
//   [RCCL all-reduce of the step scalars]
// K10 applies the last decision and writes the result back.
#include <stdlib.h>
#include <atomic>

#include "ba_common.h"
#include "ba_backsub_body.h"
#include "ba_init_body.h"
#include "imu_dual.h"

// ---------------------------------------------------------------------- K0 (body: ba_init_body.h)
__global__ void ba_init(BaDims d, BaBufs b, BaOpt opt, const double* __restrict__ cams_in,
                        const double* __restrict__ pts_in, unsigned long long free_mask, int from_mask,
                        uint8_t* __restrict__ cam_free, int32_t* __restrict__ zero_i32, int zero_n)
{
    ba_init_body(d, b, opt, cams_in, pts_in, free_mask, from_mask, cam_free, zero_i32, zero_n);
}
// batched (blockIdx.z = window; windows of at most 64 cameras: the free-camera table is the mask)
__global__ void ba_init_batch(const BaWin* w, BaOpt opt)
{
    const BaWin& x = w[blockIdx.z];
    ba_init_body(x.d, x.b, opt, x.cams_in, x.pts_in, x.free_mask, 1, x.cam_free, x.zero_ptr, x.zero_n);
}

// ---------------------------------------------------------------------- K5
__global__ __launch_bounds__(BA_THREADS) void ba_linearize_schur(BaDims d, BaBufs b, BaOpt opt, int it)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];   // [Cf][42]: U(36) gc(6)
    __shared__ BaState st_sh;
    const BaState st = ba_state_for_iteration(b, opt, it, &st_sh);
    if (st.done) return;
    const int nlds = d.Cf * 42;
    for (int i = threadIdx.x; i < nlds; i += blockDim.x) lds[i] = 0.0;
    __syncthreads();

    const double* prep = b.prep + (size_t)st.cur * d.C * BA_PREP;
    const double* Xp = b.Xp + (size_t)st.cur * d.P * 3;
    const size_t rep_off = (size_t)(blockIdx.x & (BA_UREP - 1)) * b.cam_stride;
    double cost = 0.0, gmax = 0.0, fail = 0.0;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < d.P) {
        const double X[3] = {Xp[3 * (size_t)p], Xp[3 * (size_t)p + 1], Xp[3 * (size_t)p + 2]};
        const int o0 = b.obs_ptr[p], o1 = b.obs_ptr[p + 1];
        double V[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
        ObsLin o;
        for (int oi = o0; oi < o1; oi++) {
            const int c = b.obs_cam[oi];
            obs_eval<true>(prep + (size_t)c * BA_PREP, X, b.obs_uv[oi], d, o);
            cost += 0.5 * o.rho;
            const double w = o.w;
            V[0] += w * (o.jp[0] * o.jp[0] + o.jp[3] * o.jp[3]);
            V[1] += w * (o.jp[0] * o.jp[1] + o.jp[3] * o.jp[4]);
            V[2] += w * (o.jp[0] * o.jp[2] + o.jp[3] * o.jp[5]);
            V[3] += w * (o.jp[1] * o.jp[1] + o.jp[4] * o.jp[4]);
            V[4] += w * (o.jp[1] * o.jp[2] + o.jp[4] * o.jp[5]);
            V[5] += w * (o.jp[2] * o.jp[2] + o.jp[5] * o.jp[5]);
#pragma unroll
            for (int k = 0; k < 3; k++) g[k] += w * (o.jp[k] * o.r0 + o.jp[3 + k] * o.r1);
            const int s = b.slot[c];
            if (s >= 0) {
                double* u = lds + s * 42;
#pragma unroll
                for (int a = 0; a < 6; a++) {
#pragma unroll
                    for (int e = a; e < 6; e++) atomicAdd(&u[a * 6 + e], w * (o.jc[a] * o.jc[e] + o.jc[6 + a] * o.jc[6 + e]));
                    atomicAdd(&u[36 + a], w * (o.jc[a] * o.r0 + o.jc[6 + a] * o.r1));
                }
            }
        }
        gmax = fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2])));
        // Jacobi scaling (first linearisation) and LM damping of the point block
        double sp[3], lam[3];
        const double Vd[3] = {V[0], V[3], V[5]};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (!st.have_scale) {
                sp[k] = opt.jacobi ? 1.0 / (1.0 + sqrt(Vd[k])) : 1.0;
                b.sp[3 * (size_t)p + k] = sp[k];
            } else {
                sp[k] = b.sp[3 * (size_t)p + k];
            }
            const double s2 = sp[k] * sp[k];
            lam[k] = clampd(s2 * Vd[k], opt.dmin, opt.dmax) / (st.radius * s2);
            b.lamp[3 * (size_t)p + k] = lam[k];
            b.gp[3 * (size_t)p + k] = g[k];
        }
        double Vdm[6] = {V[0] + lam[0], V[1], V[2], V[3] + lam[1], V[4], V[5] + lam[2]}, I[6];
        const bool ok = inv3_psd(Vdm, I);
        if (!ok) { fail = 1.0; I[0] = I[1] = I[2] = I[3] = I[4] = I[5] = 0.0; }
#pragma unroll
        for (int k = 0; k < 6; k++) b.Vinv[6 * (size_t)p + k] = I[k];
        // Schur products: S[si,sj] -= Y_i W_j^T, rhs[si] -= Y_i g,  Y_i = W_i V^-1, W_i = w Jc_i^T Jp_i
        if (ok) {
            for (int oi = o0; oi < o1; oi++) {
                const int ci = b.obs_cam[oi];
                const int si = b.slot[ci];
                if (si < 0) continue;
                obs_eval<true>(prep + (size_t)ci * BA_PREP, X, b.obs_uv[oi], d, o);
                double Y[18];
#pragma unroll
                for (int a = 0; a < 6; a++) {
                    const double w0 = o.w * (o.jc[a] * o.jp[0] + o.jc[6 + a] * o.jp[3]);
                    const double w1 = o.w * (o.jc[a] * o.jp[1] + o.jc[6 + a] * o.jp[4]);
                    const double w2 = o.w * (o.jc[a] * o.jp[2] + o.jc[6 + a] * o.jp[5]);
                    Y[a * 3 + 0] = w0 * I[0] + w1 * I[1] + w2 * I[2];
                    Y[a * 3 + 1] = w0 * I[1] + w1 * I[3] + w2 * I[4];
                    Y[a * 3 + 2] = w0 * I[2] + w1 * I[4] + w2 * I[5];
                    atomicAdd(&b.rhs[rep_off + 6 * si + a], -(Y[a * 3] * g[0] + Y[a * 3 + 1] * g[1] + Y[a * 3 + 2] * g[2]));
                }
                ObsLin oj;
                for (int ojx = o0; ojx < o1; ojx++) {
                    const int cj = b.obs_cam[ojx];
                    const int sj = b.slot[cj];
                    if (sj < 0) continue;
                    obs_eval<true>(prep + (size_t)cj * BA_PREP, X, b.obs_uv[ojx], d, oj);
                    double* Sblk = b.S + (size_t)(6 * si) * d.n + 6 * sj;
#pragma unroll
                    for (int e = 0; e < 6; e++) {
                        const double w0 = oj.w * (oj.jc[e] * oj.jp[0] + oj.jc[6 + e] * oj.jp[3]);
                        const double w1 = oj.w * (oj.jc[e] * oj.jp[1] + oj.jc[6 + e] * oj.jp[4]);
                        const double w2 = oj.w * (oj.jc[e] * oj.jp[2] + oj.jc[6 + e] * oj.jp[5]);
#pragma unroll
                        for (int a = 0; a < 6; a++)
                            atomicAdd(&Sblk[(size_t)a * d.n + e], -(Y[a * 3] * w0 + Y[a * 3 + 1] * w1 + Y[a * 3 + 2] * w2));
                    }
                }
            }
        }
    }
    // block reductions
    cost = wave_sum(cost);
    fail = wave_sum(fail);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) gmax = fmax(gmax, __shfl_down(gmax, off, 64));
    if ((threadIdx.x & 63) == 0) {
        const size_t slot = (size_t)(blockIdx.x & (BA_NSLOT - 1)) * BA_SLOT_STRIDE;
        atomicAdd(&b.scal[slot], cost);
        if (fail > 0.0) atomicAdd(&b.scal[slot + 1], fail);
        atomic_max_nonneg(&b.gmax[slot], gmax);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nlds; i += blockDim.x) {
        const int s = i / 42, k = i % 42;
        const double v = lds[i];
        if (v != 0.0) {
            if (k < 36) atomicAdd(&b.U[rep_off + s * 36 + k], v);
            else atomicAdd(&b.gc[rep_off + 6 * s + (k - 36)], v);
        }
    }
}

// ---------------------------------------------------------------------- K7
// Single workgroup.  S lives in LDS when n <= BA_MAX_LDS_N, otherwise in place
// in the global accumulator.
__global__ __launch_bounds__(256) void ba_reduced_solve(BaDims d, BaBufs b, BaOpt opt, int use_lds)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int n = d.n, tid = threadIdx.x, nt = blockDim.x;
    __shared__ BaState st;
    __shared__ int s_fail;
    __shared__ double red[8];
    if (tid == 0) { st = *b.st; s_fail = 0; }
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
    double* S = use_lds ? sm : b.S;
    double* y = use_lds ? sm + (size_t)n * n : b.dc;    // rhs / solution vector
    double* lam = use_lds ? y + n : b.rhs;               // camera damping (rhs buffer is free once y is formed)

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
            double g = gslots;
            for (int w = 0; w < (nt + 63) / 64; w++) g = fmax(g, red[w]);
            if (!isfinite(st.x_cost)) { st.done = 1; st.termination = RS_BA_FAILURE; }
            else if (g <= opt.gtol) { st.done = 1; st.termination = RS_BA_CONVERGENCE_GRADIENT; }
        }
        __syncthreads();
        if (st.done) { if (tid == 0) *b.st = st; return; }
    }
    __syncthreads();

    // (2) assemble S = U + Lambda_c + (-sum Y W^T), y = gc + (-sum Y g)
    const double radius = st.radius;
    for (int i = tid; i < n; i += nt) {
        const double h = b.U[(i / 6) * 36 + (i % 6) * 7];
        const double s2 = b.sc[i] * b.sc[i];
        const double l = clampd(s2 * h, opt.dmin, opt.dmax) / (radius * s2);
        const double yy = b.gc[i] + b.rhs[i];   // lam may alias rhs (global path): same thread, same index
        lam[i] = l;
        y[i] = yy;
    }
    __syncthreads();
    // S is accumulated in its upper triangle.  Two phases because S may BE b.S (in place):
    // first mirror upper -> lower, then add the block diagonal and the damping.
    for (int idx = tid; idx < n * n; idx += nt) {
        const int i = idx / n, j = idx % n;
        if (i > j) S[idx] = b.S[(size_t)j * n + i];
        else if (S != b.S) S[idx] = b.S[idx];
    }
    __syncthreads();
    for (int idx = tid; idx < n * n; idx += nt) {
        const int i = idx / n, j = idx % n;
        if (i / 6 != j / 6) continue;
        const int a = i % 6, e = j % 6;
        double v = S[idx] + ((a <= e) ? b.U[(i / 6) * 36 + a * 6 + e] : b.U[(i / 6) * 36 + e * 6 + a]);
        if (i == j) v += lam[i];
        S[idx] = v;
    }
    __syncthreads();

    // (3) dense Cholesky S = L L^T (left-looking, column j by thread-per-row)
    for (int j = 0; j < n; j++) {
        for (int i = j + tid; i < n; i += nt) {
            double acc = S[(size_t)i * n + j];
            for (int k = 0; k < j; k++) acc -= S[(size_t)i * n + k] * S[(size_t)j * n + k];
            S[(size_t)i * n + j] = acc;   // unscaled column
        }
        __syncthreads();
        const double dj = S[(size_t)j * n + j];
        if (!(dj > 0.0) || !isfinite(dj)) { if (tid == 0) s_fail = 1; }
        const double inv = 1.0 / sqrt(dj);
        __syncthreads();
        for (int i = j + tid; i < n; i += nt) S[(size_t)i * n + j] = (i == j) ? sqrt(dj) : S[(size_t)i * n + j] * inv;
        __syncthreads();
    }
    if (tid < 64) { const double f = slot_sum(b.scal, 1); if (f > 0.0 && tid == 0) s_fail = 1; }
    __syncthreads();
    if (s_fail) {
        if (tid == 0) { st.solver_failed = 1; *b.st = st; }
        return;
    }
    // (4) forward / backward substitution, column oriented
    for (int j = 0; j < n; j++) {
        if (tid == 0) y[j] = y[j] / S[(size_t)j * n + j];
        __syncthreads();
        const double yj = y[j];
        for (int i = j + 1 + tid; i < n; i += nt) y[i] -= S[(size_t)i * n + j] * yj;
        __syncthreads();
    }
    for (int j = n - 1; j >= 0; j--) {
        if (tid == 0) y[j] = y[j] / S[(size_t)j * n + j];
        __syncthreads();
        const double yj = y[j];
        for (int i = tid; i < j; i += nt) y[i] -= S[(size_t)j * n + i] * yj;
        __syncthreads();
    }
    // (5) delta_c = -y, candidate cameras, camera part of the scalars
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
            double dlt = 0.0;
            if (s >= 0) {
                dlt = -y[6 * s + k];
                if (!isfinite(dlt)) bad = true;
                mcc += 0.5 * (dlt * dlt * lam[6 * s + k] - dlt * b.gc[6 * s + k]);
                const double xn = x + dlt;
                if (active) { ssq += (x - xn) * (x - xn); xsq += x * x; }
                Xn[6 * c + k] = xn;
            } else {
                Xn[6 * c + k] = x;
            }
        }
        cam_prepare(Xn + 6 * c, b.prep + ((size_t)(st.cur ^ 1) * d.C + c) * BA_PREP);
    }
    __syncthreads();   // all reads of y done before dc (may alias y) is rewritten
    for (int i = tid; i < n; i += nt) { const double v = -y[i]; b.dc[i] = v; }   // dc may alias y: same index
    mcc = wave_sum(mcc); ssq = wave_sum(ssq); xsq = wave_sum(xsq);
    if (__any(bad) && (tid & 63) == 0) s_fail = 1;
    __syncthreads();
    __shared__ double red3[4][3];
    if ((tid & 63) == 0) { red3[tid >> 6][0] = mcc; red3[tid >> 6][1] = ssq; red3[tid >> 6][2] = xsq; }
    __syncthreads();
    if (tid == 0) {
        double a0 = 0, a1 = 0, a2 = 0;
        for (int w = 0; w < (nt + 63) / 64; w++) { a0 += red3[w][0]; a1 += red3[w][1]; a2 += red3[w][2]; }
        st.cam_scal[0] = a0; st.cam_scal[1] = a1; st.cam_scal[2] = a2;
        st.solver_failed = s_fail;
        *b.st = st;
    }
}

// ---------------------------------------------------------------------- K8
__global__ __launch_bounds__(BA_THREADS) void ba_backsub_cost(BaDims d, BaBufs b)
{
    const BaState st = *b.st;
    if (st.done) return;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < b.acc_count; i += (size_t)gridDim.x * blockDim.x) b.acc[i] = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < BA_NSLOT * BA_SLOT_STRIDE; i += gridDim.x * blockDim.x) b.gmax[i] = 0.0;
    if (st.solver_failed) return;
    const double* prep = b.prep + (size_t)st.cur * d.C * BA_PREP;
    const double* prepn = b.prep + (size_t)(st.cur ^ 1) * d.C * BA_PREP;
    const double* Xp = b.Xp + (size_t)st.cur * d.P * 3;
    double* Xn = b.Xp + (size_t)(st.cur ^ 1) * d.P * 3;
    double cost = 0.0, mcc = 0.0, ssq = 0.0, xsq = 0.0;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < d.P) {
        const double X[3] = {Xp[3 * (size_t)p], Xp[3 * (size_t)p + 1], Xp[3 * (size_t)p + 2]};
        const int o0 = b.obs_ptr[p], o1 = b.obs_ptr[p + 1];
        double t[3] = {b.gp[3 * (size_t)p], b.gp[3 * (size_t)p + 1], b.gp[3 * (size_t)p + 2]};
        const double g[3] = {t[0], t[1], t[2]};
        ObsLin o;
        for (int oi = o0; oi < o1; oi++) {
            const int c = b.obs_cam[oi];
            const int s = b.slot[c];
            if (s < 0) continue;
            obs_eval<true>(prep + (size_t)c * BA_PREP, X, b.obs_uv[oi], d, o);
            double m0 = 0.0, m1 = 0.0;
#pragma unroll
            for (int a = 0; a < 6; a++) { const double dc = b.dc[6 * s + a]; m0 += o.jc[a] * dc; m1 += o.jc[6 + a] * dc; }
#pragma unroll
            for (int k = 0; k < 3; k++) t[k] += o.w * (o.jp[k] * m0 + o.jp[3 + k] * m1);   // W_i^T delta_c
        }
        const double* I = b.Vinv + 6 * (size_t)p;
        const double dp[3] = {-(I[0] * t[0] + I[1] * t[1] + I[2] * t[2]), -(I[1] * t[0] + I[3] * t[1] + I[4] * t[2]),
                              -(I[2] * t[0] + I[4] * t[1] + I[5] * t[2])};
        double Xc[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            Xc[k] = X[k] + dp[k];
            Xn[3 * (size_t)p + k] = Xc[k];
            mcc += 0.5 * (dp[k] * dp[k] * b.lamp[3 * (size_t)p + k] - dp[k] * g[k]);
            ssq += (X[k] - Xc[k]) * (X[k] - Xc[k]);
            xsq += X[k] * X[k];
        }
        for (int oi = o0; oi < o1; oi++) {
            obs_eval<false>(prepn + (size_t)b.obs_cam[oi] * BA_PREP, Xc, b.obs_uv[oi], d, o);
            cost += 0.5 * o.rho;
        }
    }
    cost = wave_sum(cost); mcc = wave_sum(mcc); ssq = wave_sum(ssq); xsq = wave_sum(xsq);
    if ((threadIdx.x & 63) == 0) {
        const size_t slot = (size_t)(blockIdx.x & (BA_NSLOT - 1)) * BA_SLOT_STRIDE;
        atomicAdd(&b.pt_scal[slot + 0], cost);
        atomicAdd(&b.pt_scal[slot + 1], mcc);
        atomicAdd(&b.pt_scal[slot + 2], ssq);
        atomicAdd(&b.pt_scal[slot + 3], xsq);
    }
}

// -------------------------------------------------------------------- finalize
#ifndef BA_FINALIZE_WGS
#define BA_FINALIZE_WGS 32
#endif
static __device__ __forceinline__ void ba_finalize_body(const BaDims& d, const BaBufs& b, const BaOpt& opt, int it, double* __restrict__ cams_out,
                                                        const uint8_t* __restrict__ cam_free, double* __restrict__ pts_out, BaState* host_st,
                                                        BaTrace* host_trace, double* __restrict__ host_cams, double* __restrict__ host_vb,
                                                        volatile int* host_done = nullptr, int32_t* __restrict__ zero_i32 = nullptr, int zero_n = 0)
{
    __shared__ int usable, cur;
    __shared__ BaState st_fin;
    // the decisions of the last round (every workgroup recomputes them; b.st_prev / b.pt_prev are immutable here)
    if (threadIdx.x < 64) ba_decide(b, opt, it, blockIdx.x == 0 ? b.trace : nullptr, &st_fin, false);
    __syncthreads();
    if (threadIdx.x == 0) {
        BaState st = st_fin;
        // after a successful step the cost at the new point is K5's value if it ran, else the candidate cost
        // a K8 workgroup of the fused launch never saw its K7 publish: the result is unusable — not because the solver failed
        // but because of scheduling (queue preemption, another process on the GPU, counter collection): the host re-runs the
        // solve as separate launches from the untouched inputs (ba_solve_impl)
        if (b.dbg[BA_HAND_ERR] != 0ull) { st.termination = RS_BA_FAILURE; st.hand_lost = 1; }
        const bool ok = st.termination != RS_BA_FAILURE && isfinite(st.x_cost) && st.x_cost <= st.initial_cost;
        usable = ok ? 1 : 0;
        cur = st.cur;
        st.usable = usable;
        if (blockIdx.x == 0) {
            *b.st = st;
            *host_st = st;            // the summary goes straight into pinned host memory: no copy launch after the solve
            if (!host_done) __threadfence_system();      // (with a completion flag, the ONE fence in front of the flag covers it)
        }
    }
    __syncthreads();
    if (blockIdx.x == 0) {            // the per-iteration record follows the summary (entries of earlier iterations were
        const int ne = st_fin.iter * (int)(sizeof(BaTrace) / sizeof(double));   // written by earlier launches, the last one above)
        const double* src = (const double*)b.trace;
        double* dst = (double*)host_trace;
        for (int i = threadIdx.x; i < ne; i += blockDim.x) dst[i] = src[i];
    }
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    const double* Xc = b.Xc + (size_t)cur * d.C * 6;
    const double* Xp = b.Xp + (size_t)cur * d.P * 3;
    // the grouping's histogram / cursors / span word go back to zero for the next solve's count launch (ba_init_count)
    for (int i = tid; i < zero_n; i += nth) zero_i32[i] = 0;
    // Everything the HOST waits for is written by workgroup 0 alone and fenced once: the summary and the trace (above), and
    // the cameras as the caller will see them in d_cameras, mirrored into pinned host memory (poses are host-owned objects in
    // the reference — Frame::set_pose, src/Optimization.cpp:363-368 — so the shim needs them there anyway).  Then it raises
    // the completion flag the host spins on (hipStreamSynchronize is a blocking wait whose wake-up costs tens of
    // microseconds).  The device-side results (d_cameras, d_points) are written by the whole grid and are STREAM-ordered:
    // the call may return while those copies are still running; whatever reads them on the context's stream — the next
    // library call, a staged download, a torch op — is ordered behind them.
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < d.C * 6; i += blockDim.x) host_cams[i] = (usable && cam_free[i / 6]) ? Xc[i] : cams_out[i];
        if (host_vb)        // inertial solve: velocity | bias of the accepted state (the host applies the write-back rule)
            for (int i = threadIdx.x; i < d.C * 9; i += blockDim.x) host_vb[i] = b.imu.Xv[(size_t)cur * d.C * 9 + i];
        if (host_done) {
            __threadfence_system();
            __syncthreads();
            if (threadIdx.x == 0) *host_done = it + 1;
        } else {
            __syncthreads();     // (host_cams reads cams_out before the grid rewrites it: this workgroup's share of that is below)
        }
    }
    if (usable) {
        // (workgroup 0 has read cams_out for the mirror above; the other workgroups only touch entries of FREE cameras, whose
        // mirror value comes from Xc, so no ordering between the workgroups is needed)
        for (int i = tid; i < d.C * 6; i += nth)
            if (cam_free[i / 6]) cams_out[i] = Xc[i];
        for (int i = tid; i < d.P * 3; i += nth) pts_out[i] = Xp[i];
    }
}

__global__ void ba_finalize(BaDims d, BaBufs b, BaOpt opt, int it, double* __restrict__ cams_out,
                            const uint8_t* __restrict__ cam_free, double* __restrict__ pts_out, BaState* host_st,
                            BaTrace* host_trace, double* __restrict__ host_cams, double* __restrict__ host_vb, volatile int* host_done,
                            int32_t* __restrict__ zero_i32, int zero_n)
{
    ba_finalize_body(d, b, opt, it, cams_out, cam_free, pts_out, host_st, host_trace, host_cams, host_vb, host_done, zero_i32, zero_n);
}
__global__ void ba_finalize_batch(const BaWin* w, BaOpt opt, int it)
{
    const BaWin& x = w[blockIdx.z];
    const BaBufs b = ba_win_round(x, it, true);
    ba_finalize_body(x.d, b, opt, it, x.cams_out, x.cam_free, x.pts_out, x.h_st, x.h_trace, x.h_cams, nullptr);
}

// Landmark-sharded solves: two facts every rank must agree on travel as MIN all-reduces of one 64-bit key each (the
// collective the sharded matcher already uses): the largest camera span of a landmark (decides between the banded and
// the general reduced solve: every rank must run the same factorisation on the all-reduced system) and "some rank lost a
// hand-off of its fused K7 + K8 launch" (every rank re-runs the solve, or none: their collectives must stay paired).
#define BA_KEY_SPAN 58
#define BA_KEY_LOST 59
__global__ void ba_span_key(const int32_t* __restrict__ maxspan, unsigned long long* __restrict__ key)
{
    *key = ~(unsigned long long)(unsigned)max(*maxspan, 0);            // min of ~span = ~(max span)
}
__global__ void ba_lost_key(const BaState* __restrict__ st, unsigned long long* __restrict__ key)
{
    *key = st->hand_lost ? 0ull : 1ull;                                // min = 0 as soon as one rank lost a hand-off
}

// ------------------------------------------------------------------ host side
extern "C" void rs_ba_default_options(rs_ba_options* o)
{
    if (!o) return;
    o->max_num_iterations = 10;
    o->huber_delta = sqrt(5.991);
    o->initial_trust_region_radius = 1e4;
    o->max_trust_region_radius = 1e16;
    o->min_trust_region_radius = 1e-32;
    o->min_relative_decrease = 1e-3;
    o->min_lm_diagonal = 1e-6;
    o->max_lm_diagonal = 1e32;
    o->function_tolerance = 1e-6;
    o->gradient_tolerance = 1e-10;
    o->parameter_tolerance = 1e-8;
    o->max_num_consecutive_invalid_steps = 5;
    o->jacobi_scaling = 1;
}

void rs_ba_cache_free(rs_context* ctx) { (void)ctx; }

extern "C" int rs_prof_counters(rs_context* ctx, uint64_t* h_out, int n)
{
    if (!ctx || !h_out || n < 0 || n > 64) return RS_ERR_INVALID;
    if (!ctx->ba_cache) { for (int i = 0; i < n; i++) h_out[i] = 0; return RS_OK; }
    RS_HIP(ctx, hipMemcpyAsync(h_out, ctx->ba_cache, sizeof(uint64_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    RS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RS_OK;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// the inertial residual blocks of one solve (null for a vision-only solve)
struct BaInertialArgs {
    double* h_velocity;            // [C][3] in/out
    double* h_bias;                // [C][6] in/out
    const rs_imu_factor* factors;
    int n_factors;
    const double* gravity;
};

// whitener of src/ImuFactor.cpp:10-17: L^-1 of the covariance's LLT, identity when it is not positive definite
static void imu_whitener(const double cov[81], double W[81])
{
    double L[81];
    memcpy(L, cov, sizeof L);
    bool ok = true;
    for (int j = 0; j < 9 && ok; j++) {
        double dd = L[j * 9 + j];
        for (int k = 0; k < j; k++) dd -= L[j * 9 + k] * L[j * 9 + k];
        if (!(dd > 0.0) || !std::isfinite(dd)) { ok = false; break; }
        dd = sqrt(dd);
        L[j * 9 + j] = dd;
        for (int i = j + 1; i < 9; i++) {
            double t = L[i * 9 + j];
            for (int k = 0; k < j; k++) t -= L[i * 9 + k] * L[j * 9 + k];
            L[i * 9 + j] = t / dd;
        }
    }
    memset(W, 0, sizeof(double) * 81);
    if (!ok) { for (int i = 0; i < 9; i++) W[i * 9 + i] = 1.0; return; }
    for (int c = 0; c < 9; c++)
        for (int i = 0; i < 9; i++) {
            double t = (i == c) ? 1.0 : 0.0;
            for (int k = 0; k < i; k++) t -= L[i * 9 + k] * W[k * 9 + c];
            W[i * 9 + c] = t / L[i * 9 + i];
        }
}

// Workspace layout of one window (all offsets multiples of 256 B) and the BaBufs pointers into it: shared by the single
// solve and by every window of a batched solve.
struct BaLayout {
    size_t Xc, Xp, prep, slot, sc, sp, Vinv, gp, lamp, Vc, Ukeep, acc, pts, dc, st, set, trace, dbg, fre, grp;
    size_t bytes;                       // end of the common part (callers may carve more behind it)
    size_t acc_count, cam_stride, pts_block;
};
static BaLayout ba_layout(const BaDims& d, int ns, int max_iter, size_t n_ranks, size_t grp_bytes, int srep = 1)
{
    BaLayout L;
    const size_t n = (size_t)d.n, C = (size_t)d.C, P = (size_t)d.P, nb = (size_t)ns + 1;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
    L.Xc = carve(sizeof(double) * nb * C * 6); L.Xp = carve(sizeof(double) * nb * P * 3);
    L.prep = carve(sizeof(double) * nb * C * BA_PREP); L.slot = carve(sizeof(int32_t) * C);
    L.sc = carve(sizeof(double) * (n + 1)); L.sp = carve(sizeof(double) * P * 3);
    L.Vinv = carve(sizeof(double) * ns * P * 6); L.gp = carve(sizeof(double) * P * 3);
    L.lamp = carve(sizeof(double) * ns * P * 3);
    L.Vc = carve(sizeof(double) * P * 6); L.Ukeep = carve(sizeof(double) * ((size_t)d.Cf * 36 + n + 1));
    L.cam_stride = (size_t)ns * n + (size_t)d.Cf * 36 + n;
    L.acc_count = (size_t)srep * ns * n * n + (size_t)BA_UREP * L.cam_stride + (1 + n_ranks) * (size_t)BA_NSLOT * BA_SLOT_STRIDE;
    L.acc = carve(sizeof(double) * L.acc_count);
    L.pts_block = (size_t)ns * BA_NSLOT * BA_SLOT_STRIDE;
    L.pts = carve(sizeof(double) * 2 * L.pts_block); L.dc = carve(sizeof(double) * ns * BA_DC_STRIDE(n));
    L.st = carve(sizeof(BaState) * 2);
    L.set = carve(sizeof(BaSetOut) * 2 * BA_MAXSETS);
    L.trace = carve(sizeof(BaTrace) * (size_t)(max_iter + 1));
    L.dbg = carve(sizeof(unsigned long long) * BA_DBG_WORDS);
    L.fre = carve(C);
    L.grp = carve(grp_bytes);
    L.bytes = off;
    return L;
}
// pointers of round parity 0 (the double-buffered blocks are re-pointed per round); rank = this rank's gradient-max block
static void ba_bind(BaBufs& b, char* ws, const BaLayout& L, const BaDims& d, int ns, int n_ranks, int rank, int srep = 1)
{
    const size_t n = (size_t)d.n;
    b.ns = ns;
    b.srep = srep; b.s_rep_stride = (size_t)ns * n * n;
    b.Xc = (double*)(ws + L.Xc); b.Xp = (double*)(ws + L.Xp); b.prep = (double*)(ws + L.prep);
    b.slot = (int32_t*)(ws + L.slot); b.sc = (double*)(ws + L.sc); b.sp = (double*)(ws + L.sp);
    b.Vinv = (double*)(ws + L.Vinv); b.gp = (double*)(ws + L.gp); b.lamp = (double*)(ws + L.lamp);
    b.Vc = (double*)(ws + L.Vc); b.Ukeep = (double*)(ws + L.Ukeep);
    b.acc = (double*)(ws + L.acc); b.acc_count = L.acc_count;
    b.S = b.acc; b.rhs = b.S + (size_t)srep * ns * n * n; b.U = b.rhs + (size_t)ns * n; b.gc = b.U + (size_t)d.Cf * 36;
    b.cam_stride = L.cam_stride; b.scal = b.rhs + (size_t)BA_UREP * L.cam_stride;
    b.gmax_all = b.scal + (size_t)BA_NSLOT * BA_SLOT_STRIDE; b.gmax_blocks = n_ranks; b.decided = 0;
    b.gmax = b.gmax_all + (size_t)rank * BA_NSLOT * BA_SLOT_STRIDE;
    b.pt_scal = (double*)(ws + L.pts); b.pt_prev = b.pt_scal; b.dc = (double*)(ws + L.dc);
    b.st = (BaState*)(ws + L.st); b.st_prev = b.st;
    b.trace = (BaTrace*)(ws + L.trace);
    b.set_out = (BaSetOut*)(ws + L.set); b.set_prev = b.set_out;
    b.dbg = (unsigned long long*)(ws + L.dbg);
    b.hand_timeout = BA_HAND_TIMEOUT_TICKS;
}

// Solves of this process that are between entry and return right now (any context, any thread).  A solve that has the
// device to itself — as far as the library can tell — runs K7 + K8 as one launch: its K8 workgroups hold a CU each while
// they wait for K7.  When other solves are in flight (several sessions on one GPU) those CUs are what the others need,
// and the two-launch form gives the higher aggregate: 8 sessions 3.95 k -> 4.70 k solves/s (tools/multi_session.py).
static std::atomic<int> g_ba_in_flight{0};
struct BaInFlight {
    int others;
    BaInFlight() : others(g_ba_in_flight.fetch_add(1)) {}
    ~BaInFlight() { g_ba_in_flight.fetch_sub(1); }
};

static int ba_solve_once(rs_context* ctx, int n_cameras, int n_points, int n_obs, double* d_cameras,
                         const uint8_t* h_cam_free, double* d_points, const int32_t* d_obs_ptr,
                         const int32_t* d_obs_cam, const float* d_obs_uv, const float h_intrinsics[4],
                         const rs_ba_options* options, rs_ba_summary* h_summary, const BaInertialArgs* in,
                         bool allow_fuse, bool* hand_lost)
{
    *hand_lost = false;
    if (!ctx || !h_summary) return RS_ERR_INVALID;
    const BaInFlight in_flight;
    memset(h_summary, 0, sizeof *h_summary);
    if (n_cameras < 0 || n_points < 0 || n_obs < 0) return rs_fail(ctx, RS_ERR_INVALID, "negative size");
    // A landmark shard may be EMPTY (more ranks than landmarks, or a rank whose range holds none): the rank still takes
    // part in every exchange step with zero contributions and solves the reduced system like the others.  Cameras are
    // replicated, so n_cameras == 0 is the same on every rank and ends the call everywhere.
    const bool sharded = rs_comm_active(ctx) && ctx->n_ranks > 1;
    if (n_cameras == 0 || ((n_points == 0 || n_obs == 0) && !sharded)) {   // nothing to optimise
        h_summary->usable = 0;
        h_summary->termination = RS_BA_FAILURE;
        return RS_OK;
    }
    if (!d_cameras || !h_cam_free || !d_obs_ptr || !h_intrinsics || (n_points > 0 && !d_points) || (n_obs > 0 && (!d_obs_cam || !d_obs_uv)))
        return rs_fail(ctx, RS_ERR_INVALID, "null pointer");
    rs_ba_options def;
    if (!options) { rs_ba_default_options(&def); options = &def; }
    RS_HIP(ctx, hipSetDevice(ctx->device));

    BaDims d;
    d.C = n_cameras; d.P = n_points; d.M = n_obs;
    std::vector<int32_t> slot(n_cameras);
    d.Cf = 0;
    for (int c = 0; c < n_cameras; c++) slot[c] = h_cam_free[c] ? d.Cf++ : -1;
    d.n = 6 * d.Cf;
    d.fx = h_intrinsics[0]; d.fy = h_intrinsics[1]; d.cx = h_intrinsics[2]; d.cy = h_intrinsics[3];
    d.huber_a = options->huber_delta;
    BaOpt opt;
    opt.max_iter = options->max_num_iterations; opt.max_invalid = options->max_num_consecutive_invalid_steps;
    opt.jacobi = options->jacobi_scaling; opt.r0 = options->initial_trust_region_radius;
    opt.rmax = options->max_trust_region_radius; opt.rmin = options->min_trust_region_radius;
    opt.min_rel = options->min_relative_decrease; opt.dmin = options->min_lm_diagonal;
    opt.dmax = options->max_lm_diagonal; opt.ftol = options->function_tolerance;
    opt.gtol = options->gradient_tolerance; opt.ptol = options->parameter_tolerance;
    if (opt.max_iter < 0 || opt.max_iter > 1000) return rs_fail(ctx, RS_ERR_INVALID, "max_num_iterations out of range");
    // beyond 128 free cameras K5 is the generic kernel with its U / gc partial sums in LDS: 42 doubles per free camera of
    // the 160 KB a workgroup may hold
    if ((size_t)d.Cf * 42 * sizeof(double) > 159 * 1024) return rs_fail(ctx, RS_ERR_UNSUPPORTED, "more than 484 free cameras");

    // which kernels: the MFMA Schur path + LDS reduced solve + LDS back-substitution form the fast path of a local
    // window; only that path evaluates speculative radii (ns > 1)
    const bool use_mfma = d.Cf >= 1 && d.Cf <= 128 && ba_schur_lds_bytes(d.C, d.Cf) <= 160 * 1024;
    const bool k8_lds = ba_backsub_lds_bytes(d.C, d.n) <= 64 * 1024;
    const bool solve_lds = d.n >= 6 && d.n <= BA_MAX_LDS_N;
    const bool solve_big = d.n > BA_MAX_LDS_N;
    // inertial frames: the cameras an IMU factor touches get a velocity (3) + bias (6) block
    std::vector<int32_t> inert(n_cameras, -1);
    int Ci = 0;
    if (in) {
        for (int f = 0; f < in->n_factors; f++) {
            const int i = in->factors[f].cam_i, j = in->factors[f].cam_j;
            if (i < 0 || j < 0 || i >= n_cameras || j >= n_cameras || i == j || !h_cam_free[i] || !h_cam_free[j])
                return rs_fail(ctx, RS_ERR_INVALID, "IMU factor %d must join two distinct optimised cameras", f);
        }
        std::vector<uint8_t> touched(n_cameras, 0);
        for (int f = 0; f < in->n_factors; f++) { touched[in->factors[f].cam_i] = 1; touched[in->factors[f].cam_j] = 1; }
        for (int c = 0; c < n_cameras; c++) if (touched[c]) inert[c] = Ci++;
    }
    const int N_in = d.n + 9 * Ci;
    // inertial solve on the local-window kernels (ba_imu.hip): the velocity / bias blocks are eliminated around the LDS
    // K7.  Needs the factors to join consecutive inertial cameras (what the reference builds: block-tridiagonal H_zz).
    bool imu_lds = in && use_mfma && k8_lds && solve_lds && Ci >= 2 && Ci <= ba_imu_lds_path_max_ci() && ctx->ba_imu_mode == 0;
    if (imu_lds)
        for (int f = 0; f < in->n_factors && imu_lds; f++) imu_lds = inert[in->factors[f].cam_j] == inert[in->factors[f].cam_i] + 1;
    int ns = 1;
    if (use_mfma && k8_lds && solve_lds && (!in || imu_lds)) {
        // (inertial solves keep three: their per-radius elimination kernels were sized and tested for that)
        ns = ctx->ba_sets > 0 ? ctx->ba_sets : (in ? BA_CALIBRATED_SETS : BA_DEFAULT_SETS);
        if (ns > BA_MAXSETS) ns = BA_MAXSETS;
        if (in && ns > BA_CALIBRATED_SETS) ns = BA_CALIBRATED_SETS;
        if (ns > opt.max_iter) ns = opt.max_iter > 0 ? opt.max_iter : 1;
    }
    const size_t C = (size_t)d.C;
    const size_t n_ranks = rs_comm_active(ctx) ? (size_t)ctx->n_ranks : 1;
    // replicas of S for K5's scatter: the plain local window only (MFMA K5 + LDS K7, one rank, vision only)
    int srep = 1;
    if (use_mfma && solve_lds && !in && !rs_comm_active(ctx)) srep = ctx->ba_s_replicas > 0 ? ctx->ba_s_replicas : BA_DEFAULT_SREP;
    const BaLayout L = ba_layout(d, ns, opt.max_iter, n_ranks, use_mfma ? ba_group_bytes(d.P, d.Cf, d.M) : 16, srep);
    const size_t o_grp = L.grp, o_free = L.fre, pts_block = L.pts_block;
    const size_t o_big = L.bytes;
    const size_t big_bytes = align_up(in ? ba_inertial_bytes(N_in, in->n_factors, d.C) : (solve_big ? ba_big_bytes(d.n) : 16), 256);
    const size_t o_zacc = o_big + big_bytes;
    const size_t ws_bytes = o_zacc + (imu_lds ? align_up(sizeof(double) * ba_imu_lds_total_doubles(Ci, d.n, ns), 256) : 0);
    void* wsv = nullptr;
    int rc = rs_workspace_quiet(ctx, ws_bytes, &wsv);      // (this solve keeps its own account of what it leaves behind, below)
    if (rc) return rc;
    char* ws = (char*)wsv;
    BaBufs b;
    memset(&b.imu, 0, sizeof b.imu);
    b.obs_ptr = d_obs_ptr; b.obs_cam = d_obs_cam; b.obs_uv = (const float2*)d_obs_uv;
    ba_bind(b, ws, L, d, ns, (int)n_ranks, rs_comm_active(ctx) ? ctx->rank : 0, srep);
    BaState* const st_base = b.st;
    double* const pts_base = b.pt_scal;
    BaSetOut* const set_base = b.set_out;
#if RS_STAMPS
    RS_HIP(ctx, hipMemsetAsync(b.dbg, 0, sizeof(unsigned long long) * BA_DBG_WORDS, ctx->stream));
#endif
    ctx->ba_cache = b.dbg;

    void* pin = nullptr;
    const size_t pin_prog = align_up(sizeof(BaState) + sizeof(int32_t) * C + C, 64);
    const size_t pin_trace = pin_prog + 64;
    const size_t pin_cams = pin_trace + sizeof(BaTrace) * (size_t)(opt.max_iter + 1);
    const size_t pin_vb = align_up(pin_cams + sizeof(double) * 6 * C, 64);
    rc = rs_pinned(ctx, pin_vb + (in ? sizeof(double) * 9 * C : 0), &pin);
    if (rc) return rc;
    double* h_vb = in ? (double*)((char*)pin + pin_vb) : nullptr;
    ctx->ba_cams = nullptr;
    ctx->ba_cams_n = 0;
    BaProgress* h_prog = (BaProgress*)((char*)pin + pin_prog);
    h_prog->round = 0; h_prog->done = 0; h_prog->iter = 0;
    b.prog = ns > 1 ? h_prog : nullptr;
    BaState* h_st = (BaState*)pin;
    int32_t* h_slot = (int32_t*)((char*)pin + sizeof(BaState));
    uint8_t* h_free = (uint8_t*)(h_slot + C);
    BaTrace* h_trace = (BaTrace*)((char*)pin + pin_trace);
    ctx->ba_trace_n = 0;
    uint8_t* d_cam_free = (uint8_t*)(ws + o_free);
    unsigned long long free_mask = 0;
    const int from_mask = C <= 64 ? 1 : 0;
    if (from_mask) {
        for (size_t c = 0; c < C; c++) if (h_cam_free[c]) free_mask |= 1ull << c;
    } else {
        memcpy(h_slot, slot.data(), sizeof(int32_t) * C);
        memcpy(h_free, h_cam_free, C);
        RS_HIP(ctx, hipMemcpyAsync(b.slot, h_slot, sizeof(int32_t) * C, hipMemcpyHostToDevice, ctx->stream));
        RS_HIP(ctx, hipMemcpyAsync(d_cam_free, h_free, C, hipMemcpyHostToDevice, ctx->stream));
    }

    if (solve_lds && ba_prepare_reduced_solve_lds(d.n) != 0) return rs_fail(ctx, RS_ERR_HIP, "LDS attribute (K7)");
    const size_t k5_lds = sizeof(double) * (size_t)d.Cf * 42;
    if (!use_mfma && k5_lds > 48 * 1024)
        RS_HIP(ctx, rs_lds_attr((const void*)ba_linearize_schur, k5_lds));
    if (use_mfma && ba_prepare_schur(d.C, d.Cf) != 0) return rs_fail(ctx, RS_ERR_HIP, "LDS attribute (K5)");
    BaGroup grp;
    memset(&grp, 0, sizeof grp);
    if (use_mfma) {
        ba_group_carve(ws + o_grp, d.P, d.Cf, d.M, &grp);
        if (ctx->ba_item) ba_group_set_items(&grp, d.P, true, ctx->ba_item);
    }
    b.obs_cs = use_mfma ? grp.obs_cs : nullptr;

    const int pblocks = d.P > 0 ? (d.P + BA_THREADS - 1) / BA_THREADS : 1;      // (an empty shard still runs the round's decision)
    int32_t* zero_ptr = nullptr;
    int zero_n = 0;
    if (use_mfma) ba_group_zero_range(grp, &zero_ptr, &zero_n);
    hipStream_t s = ctx->stream;
    std::vector<ImuFactorDev> fac_host;
    std::vector<double> xv_host;
    if (in) {
        ImuFactorDev* d_fac = nullptr;
        int32_t* d_inert = nullptr;
        ba_inertial_carve(ws + o_big, N_in, in->n_factors, d.C, &b.imu, &d_fac, &d_inert);
        b.imu.n_fac = in->n_factors; b.imu.Ci = Ci; b.imu.N = N_in;
        if (imu_lds) {
            b.imu.zacc = (double*)(ws + o_zacc);
            b.imu.zacc_n = (int)ba_imu_lds_zacc_doubles(Ci, d.n);
            RS_HIP(ctx, hipMemsetAsync(b.imu.zacc, 0, sizeof(double) * (size_t)b.imu.zacc_n, s));
        }
        for (int k = 0; k < 3; k++) b.imu.gravity[k] = in->gravity[k];
        fac_host.resize((size_t)in->n_factors);
        for (int f = 0; f < in->n_factors; f++) { fac_host[(size_t)f].f = in->factors[f]; imu_whitener(in->factors[f].covariance, fac_host[(size_t)f].W); }
        xv_host.resize(9 * C);
        for (size_t c = 0; c < C; c++) {
            for (int k = 0; k < 3; k++) xv_host[9 * c + k] = in->h_velocity[3 * c + k];
            for (int k = 0; k < 6; k++) xv_host[9 * c + 3 + k] = in->h_bias[6 * c + k];
        }
        // (pageable sources: the copies complete before the call returns — it synchronises below)
        RS_HIP(ctx, hipMemcpyAsync(d_fac, fac_host.data(), sizeof(ImuFactorDev) * fac_host.size(), hipMemcpyHostToDevice, s));
        RS_HIP(ctx, hipMemcpyAsync(d_inert, inert.data(), sizeof(int32_t) * C, hipMemcpyHostToDevice, s));
        RS_HIP(ctx, hipMemcpyAsync(b.imu.Xv, xv_host.data(), sizeof(double) * 9 * C, hipMemcpyHostToDevice, s));
        RS_HIP(ctx, hipMemcpyAsync(b.imu.Xv + 9 * C, xv_host.data(), sizeof(double) * 9 * C, hipMemcpyHostToDevice, s));
    }
    // Set-up launches.  Local windows: K0 and the grouping's count in one launch, its scatter (incl. the item masks) in a
    // second — two instead of four.  The count adds into a histogram that must be zero when the launch starts: the finalize
    // kernel of the previous solve leaves it so, and the context remembers that (grp_zero_ptr) unless the workspace was
    // reallocated or another caller asked for workspace bytes reaching into it since (ws_dirty_hi); otherwise one memset.
    const bool fused_setup = use_mfma && ba_setup_fusable(d, grp);
    if (fused_setup) {
        const bool known_zero = ctx->grp_zero_ptr == zero_ptr && ctx->grp_zero_n == zero_n &&
                                ctx->ws_dirty_hi <= (size_t)((char*)zero_ptr - ws);
        ctx->grp_zero_ptr = nullptr;                                  // (until this solve's finalize kernel is enqueued)
        if (!known_zero) RS_HIP(ctx, hipMemsetAsync(zero_ptr, 0, sizeof(int32_t) * (size_t)zero_n, s));
        ba_launch_setup_fused(ctx, d, b, opt, grp, (const double*)d_cameras, (const double*)d_points, free_mask, from_mask, d_cam_free);
    } else {
        ctx->grp_zero_ptr = nullptr;
        {
            rs_prof_scope ps(ctx, "K0_ba_init");
            hipLaunchKernelGGL(ba_init, dim3(64), dim3(256), 0, s, d, b, opt, (const double*)d_cameras, (const double*)d_points,
                               free_mask, from_mask, d_cam_free, zero_ptr, zero_n);
        }
        if (use_mfma) {
            rc = ba_launch_grouping(ctx, d, b, grp);
            if (rc) return rc;
        }
    }
    ctx->ws_dirty_hi = 0;
    // Blocked reduced solve: is S block-banded?  The grouping has just computed the largest camera span of a landmark; one
    // word travels to the host (the solve is milliseconds: the wait costs a few per cent of one round) and decides between
    // the one-launch banded factorisation and the general blocked one.  Sharded solves keep the general form: every rank
    // must run the same arithmetic on the all-reduced system.
    int band = 0;                     // 1: one workgroup, 2: two-sided (ba_solve_big.hip)
    if (solve_big && use_mfma && !in && ctx->ba_band_mode != 1) {
        if (rs_comm_active(ctx)) {
            // landmark shards: the span of the WHOLE window is the largest of the ranks' spans (round 4; every rank then runs
            // the same banded factorisation on the same all-reduced system, as it runs the same general one)
            volatile unsigned long long* h_key = (volatile unsigned long long*)((char*)pin + pin_prog + 32);
            *h_key = 0ull;
            hipLaunchKernelGGL(ba_span_key, dim3(1), dim3(1), 0, s, (const int32_t*)grp.maxspan, b.dbg + BA_KEY_SPAN);
            rc = rs_allreduce_min_u64(ctx, b.dbg + BA_KEY_SPAN, 1);
            if (rc) return rc;
            RS_HIP(ctx, hipMemcpyAsync((void*)h_key, b.dbg + BA_KEY_SPAN, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
            RS_HIP(ctx, hipStreamSynchronize(s));
            const unsigned long long span = ~*h_key;
            if (span <= (unsigned long long)ba_band_max_span()) band = ctx->ba_band_mode == 2 ? 1 : 2;
        } else {
            volatile int32_t* h_span = (volatile int32_t*)((char*)pin + pin_prog + 32);
            *h_span = -1;
            RS_HIP(ctx, hipMemcpyAsync((void*)h_span, grp.maxspan, sizeof(int32_t), hipMemcpyDeviceToHost, s));
            long spins = 0;
            while (*h_span < 0) {
                if ((++spins & 0x3FFFF) == 0 && hipStreamQuery(s) != hipErrorNotReady) break;
            }
            if (*h_span < 0) RS_HIP(ctx, hipStreamSynchronize(s));
            if (*h_span >= 0 && *h_span <= ba_band_max_span()) band = ctx->ba_band_mode == 2 ? 1 : 2;
        }
    }
    // One ROUND = K5 + K7 + K8 and evaluates the next `ns` LM iterations of the sequential loop (all of them only if
    // the first ns - 1 are rejected).  At least ceil(max_iter / ns) rounds are needed and at most max_iter; beyond the
    // minimum the host follows the state machine through the progress word the first kernel of every round publishes
    // in pinned memory: when round r starts with `iter` iterations done, at most max_iter - iter rounds (r included)
    // can still do work.  The host stays one round ahead of the GPU, so the stream never drains.
    // K7 + K8 as one launch (ba_solve.hip): the plain local window only — vision-only, one rank, both LDS kernels
    // and only while all of its workgroups are resident at once (one per CU: the launch carries K7's LDS): beyond that
    // K8's workgroups would run in several shifts behind the hand-off, and the launch of its own (many per CU) is faster
    b.hand_timeout = 100ull * (unsigned long long)ctx->ba_handoff_timeout_us;
    // The whole round as ONE launch (ba_round.hip: K5's item workgroups become K8's after they have counted themselves for
    // K7): the same conditions plus the MFMA K5 with its camera blocks in LDS, and again every workgroup resident at once.
    // (a landmark shard may fuse K7 + K8 too: the launch sits between the two exchange steps of the round, C1 in front of
    // it and C2 behind; whether a rank fuses is its own business — shard sizes differ — but a lost hand-off is agreed on by
    // all ranks below, so that every rank re-runs the solve or none does)
    const bool plain_window = allow_fuse && solve_lds && k8_lds && !in;
    // Measured (tools/round_stamps.py): 86 us per round against 43 + 45 as two launches — the round is a strict
    // chain (linearise -> solve -> back-substitute), so keeping the workgroups resident buys the boundary and little else.
    // It is therefore opt-in ("ba_fuse_mode" 3); the default stays K5, then K7 + K8 in one launch.
    const bool fuse_round = plain_window && !rs_comm_active(ctx) && use_mfma && ctx->ba_fuse_mode == 3 && ns <= BA_CALIBRATED_SETS &&
                            ba_round_eligible(d) && ba_round_workgroups(d, b, grp) <= ctx->n_cu;
    const bool fuse78 = !fuse_round && plain_window && d.P > 0 /* an empty landmark shard has no K8 workgroup to clear the accumulators */ &&
                        (ctx->ba_fuse_mode >= 2 || (ctx->ba_fuse_mode == 0 && in_flight.others == 0)) &&
                        ba_solve_backsub_workgroups(d, b, ctx->n_cu) <= ctx->n_cu;
    if (fuse_round && ba_prepare_round(d) != 0) return rs_fail(ctx, RS_ERR_HIP, "LDS attribute (round)");
    auto enqueue_round = [&](int it) -> int {
        // double-buffered state / step-scalar blocks: round `it` works on [it & 1] and reads [(it + 1) & 1]
        b.st = st_base + (it & 1); b.st_prev = st_base + ((it + 1) & 1);
        b.pt_scal = pts_base + (size_t)(it & 1) * pts_block;
        b.pt_prev = pts_base + (size_t)((it + 1) & 1) * pts_block;
        b.set_out = set_base + (size_t)(it & 1) * BA_MAXSETS; b.set_prev = set_base + (size_t)((it + 1) & 1) * BA_MAXSETS;
        if (fuse_round) {
            rs_prof_scope ps(ctx, "K578_ba_round");
            ba_launch_round(s, d, b, opt, grp, it);
            return RS_OK;
        }
        if (use_mfma) {
            rs_prof_scope ps(ctx, "K5_ba_schur_mfma");
            // more items than compute units: the round's decision once, in front, instead of in every item's prologue
            b.decided = grp.n_items > ctx->n_cu ? 1 : 0;
            if (b.decided) ba_launch_decide(s, b, opt, it);
            ba_launch_schur(s, d, b, opt, grp, it);
            b.decided = 0;
        } else {
            rs_prof_scope ps(ctx, "K5_ba_linearize_schur");
            hipLaunchKernelGGL(ba_linearize_schur, dim3(pblocks), dim3(BA_THREADS), k5_lds, s, d, b, opt, it);
        }
        if (rs_comm_active(ctx)) {
            rs_prof_scope ps(ctx, "C1_allreduce_system");
            // one SUM all-reduce: S | 8 x {rhs, U, gc} | cost / failure slots | every rank's gradient-max block
            int rc2 = rs_allreduce_f64(ctx, b.acc, b.acc_count, false);
            if (rc2) return rc2;
        }
        if (in && imu_lds) {
            { rs_prof_scope ps(ctx, "K6i_imu_eliminate"); ba_launch_imu_eliminate(s, d, b, opt); }
            { rs_prof_scope ps(ctx, "K7_ba_reduced_solve"); ba_launch_reduced_solve_lds(s, d, b, opt); }
            { rs_prof_scope ps(ctx, "K7i_imu_expand"); ba_launch_imu_expand(s, d, b, opt); }
        } else if (in) {
            rs_prof_scope ps(ctx, "K7_ba_reduced_solve_inertial");
            int rc2 = ba_launch_reduced_solve_inertial(ctx, d, b, opt, ws + o_big);
            if (rc2) return rc2;
        } else if (solve_lds && fuse78) {
            rs_prof_scope ps(ctx, "K78_ba_solve_backsub");
            ba_launch_solve_backsub(s, d, b, opt, ctx->n_cu);
        } else if (solve_lds) {
            rs_prof_scope ps(ctx, "K7_ba_reduced_solve");
            ba_launch_reduced_solve_lds(s, d, b, opt);
        } else if (solve_big) {
            rs_prof_scope ps(ctx, "K7_ba_reduced_solve_blocked");
            int rc2 = ba_launch_reduced_solve_big(ctx, d, b, opt, ws + o_big, band);
            if (rc2) return rc2;
        } else {
            rs_prof_scope ps(ctx, "K7_ba_reduced_solve_global");
            hipLaunchKernelGGL(ba_reduced_solve, dim3(1), dim3(256), 0, s, d, b, opt, 0);
        }
        if (fuse78) {
            // K8 ran inside the K7 launch
        } else if (k8_lds) {
            rs_prof_scope ps(ctx, "K8_ba_backsub_cost");
            ba_launch_backsub(s, d, b);
        } else {
            rs_prof_scope ps(ctx, "K8_ba_backsub_cost_global");
            hipLaunchKernelGGL(ba_backsub_cost, dim3(pblocks), dim3(BA_THREADS), 0, s, d, b);
        }
        if (rs_comm_active(ctx)) {
            rs_prof_scope ps(ctx, "C2_allreduce_cost");
            int rc2 = rs_allreduce_f64(ctx, b.pt_scal, pts_block, false);
            if (rc2) return rc2;
        }
        return RS_OK;
    };
    int rounds = 0;
    const int min_rounds = (opt.max_iter + ns - 1) / ns;
    for (; rounds < min_rounds; rounds++) {
        rc = enqueue_round(rounds);
        if (rc) return rc;
    }
    while (ns > 1 && rounds < opt.max_iter) {
        // wait until the GPU has started the last enqueued round (it then has a whole round of work in front of it)
        long spins = 0;
        while (h_prog->round < rounds) {
            if ((++spins & 0xFFFFF) == 0 && hipStreamQuery(s) != hipErrorNotReady) break;   // stream drained or failed
        }
        if (h_prog->round < rounds) break;                 // nothing left in flight: finalize reports the state
        const int it_seen = h_prog->iter, done_seen = h_prog->done;
        if (done_seen || opt.max_iter - it_seen <= 1) break;          // the round in flight is the last that can matter
        rc = enqueue_round(rounds);
        if (rc) return rc;
        rounds++;
    }
    {
        rs_prof_scope ps(ctx, "K10_ba_finalize");
        // the last decisions: round index `rounds` reads the blocks of round rounds - 1
        const int itf = rounds;
        b.st = st_base + (itf & 1); b.st_prev = st_base + ((itf + 1) & 1);
        b.pt_prev = pts_base + (size_t)((itf + 1) & 1) * pts_block;
        b.set_prev = set_base + (size_t)((itf + 1) & 1) * BA_MAXSETS;
        b.prog = nullptr;
        h_prog->pad = 0;
        hipLaunchKernelGGL(ba_finalize, dim3(BA_FINALIZE_WGS), dim3(256), 0, s, d, b, opt, itf, d_cameras, (const uint8_t*)d_cam_free, d_points, h_st, h_trace,
                           (double*)((char*)pin + pin_cams), h_vb, &h_prog->pad, fused_setup ? zero_ptr : nullptr, fused_setup ? zero_n : 0);
        RS_HIP(ctx, hipGetLastError());
        if (fused_setup) { ctx->grp_zero_ptr = zero_ptr; ctx->grp_zero_n = zero_n; }      // stream-ordered in front of the next solve
        // wait for the completion flag the last workgroup of ba_finalize raises in pinned memory (summary, trace, camera
        // mirror are then all there); fall back to the stream if it drains without the flag (a failed launch)
        long spins = 0;
        while (h_prog->pad != itf + 1) {
            if ((++spins & 0x3FFFF) == 0 && hipStreamQuery(s) != hipErrorNotReady) break;
        }
        if (h_prog->pad != itf + 1) RS_HIP(ctx, hipStreamSynchronize(s));
    }
    RS_HIP(ctx, hipGetLastError());
    h_summary->termination = h_st->termination;
    h_summary->iterations = h_st->iter;
    h_summary->successful_steps = h_st->successful;
    h_summary->usable = h_st->usable;
    h_summary->initial_cost = h_st->initial_cost;
    h_summary->final_cost = h_st->x_cost;
    h_summary->final_radius = h_st->radius;
    ctx->ba_trace = h_trace;            // stays valid until the next call that uses the pinned block
    ctx->ba_trace_n = h_st->iter;
    ctx->ba_stats[0] = h_st->n_rounds; ctx->ba_stats[1] = h_st->n_fresh; ctx->ba_stats[2] = h_st->n_sets;
    ctx->ba_stats[3] = rounds;
    *hand_lost = h_st->hand_lost != 0;
    if (rs_comm_active(ctx) && allow_fuse && solve_lds && k8_lds && !in) {
        // every rank gets here (the conditions above are the same on all of them: cameras are replicated), fused or not
        volatile unsigned long long* h_key = (volatile unsigned long long*)((char*)pin + pin_prog + 32);
        hipLaunchKernelGGL(ba_lost_key, dim3(1), dim3(1), 0, s, (const BaState*)b.st, b.dbg + BA_KEY_LOST);
        rc = rs_allreduce_min_u64(ctx, b.dbg + BA_KEY_LOST, 1);
        if (rc) return rc;
        RS_HIP(ctx, hipMemcpyAsync((void*)h_key, b.dbg + BA_KEY_LOST, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        RS_HIP(ctx, hipStreamSynchronize(s));
        *hand_lost = *h_key == 0ull;
    }
    ctx->ba_cams = (const double*)((char*)pin + pin_cams);
    ctx->ba_cams_n = n_cameras;
    if (in && h_st->usable)                                  // unpack_inertial for the optimised frames, src/Optimization.cpp:363-368
        for (size_t c = 0; c < C; c++) {
            if (!h_cam_free[c]) continue;
            for (int k = 0; k < 3; k++) in->h_velocity[3 * c + k] = h_vb[9 * c + k];
            for (int k = 0; k < 6; k++) in->h_bias[6 * c + k] = h_vb[9 * c + 3 + k];
        }
    return RS_OK;
}

// A lost hand-off inside the fused K7 + K8 launch (ba_backsub_body.h) says something about scheduling, not about the
// data: the inputs are untouched (nothing is written back from an unusable solve), so the solve runs once more as
// separate launches, which need no hand-off.  Counted in rs_ba_get_stats [4].
static int ba_solve_impl(rs_context* ctx, int n_cameras, int n_points, int n_obs, double* d_cameras,
                         const uint8_t* h_cam_free, double* d_points, const int32_t* d_obs_ptr,
                         const int32_t* d_obs_cam, const float* d_obs_uv, const float h_intrinsics[4],
                         const rs_ba_options* options, rs_ba_summary* h_summary, const BaInertialArgs* in)
{
    bool lost = false;
    int rc = ba_solve_once(ctx, n_cameras, n_points, n_obs, d_cameras, h_cam_free, d_points, d_obs_ptr, d_obs_cam, d_obs_uv,
                           h_intrinsics, options, h_summary, in, true, &lost);
    if (rc == RS_OK && lost) {
        ctx->ba_stats[4]++;
        rc = ba_solve_once(ctx, n_cameras, n_points, n_obs, d_cameras, h_cam_free, d_points, d_obs_ptr, d_obs_cam, d_obs_uv,
                           h_intrinsics, options, h_summary, in, false, &lost);
    }
    return rc;
}

extern "C" int rs_bundle_adjust(rs_context* ctx, int n_cameras, int n_points, int n_obs, double* d_cameras,
                                const uint8_t* h_cam_free, double* d_points, const int32_t* d_obs_ptr,
                                const int32_t* d_obs_cam, const float* d_obs_uv, const float h_intrinsics[4],
                                const rs_ba_options* options, rs_ba_summary* h_summary)
{
    return ba_solve_impl(ctx, n_cameras, n_points, n_obs, d_cameras, h_cam_free, d_points, d_obs_ptr, d_obs_cam, d_obs_uv,
                         h_intrinsics, options, h_summary, nullptr);
}

extern "C" int rs_bundle_adjust_inertial(rs_context* ctx, int n_cameras, int n_points, int n_obs, double* d_cameras,
                                         const uint8_t* h_cam_free, double* d_points, const int32_t* d_obs_ptr,
                                         const int32_t* d_obs_cam, const float* d_obs_uv, const float h_intrinsics[4],
                                         double* h_velocity, double* h_bias, const rs_imu_factor* h_factors, int n_factors,
                                         const double h_gravity[3], const rs_ba_options* options, rs_ba_summary* h_summary)
{
    if (n_factors < 0) return ctx ? rs_fail(ctx, RS_ERR_INVALID, "negative n_factors") : RS_ERR_INVALID;
    if (n_factors == 0)                                      // InertialInput::usable() false / no pair with >= 2 samples
        return ba_solve_impl(ctx, n_cameras, n_points, n_obs, d_cameras, h_cam_free, d_points, d_obs_ptr, d_obs_cam, d_obs_uv,
                             h_intrinsics, options, h_summary, nullptr);
    if (!h_velocity || !h_bias || !h_factors || !h_gravity) return ctx ? rs_fail(ctx, RS_ERR_INVALID, "null pointer") : RS_ERR_INVALID;
    const BaInertialArgs in{h_velocity, h_bias, h_factors, n_factors, h_gravity};
    return ba_solve_impl(ctx, n_cameras, n_points, n_obs, d_cameras, h_cam_free, d_points, d_obs_ptr, d_obs_cam, d_obs_uv,
                         h_intrinsics, options, h_summary, &in);
}

// ---------------------------------------------------------------- batch of independent windows
// Several sessions served by one GPU: the windows are independent problems, each a latency chain of small launches
// that fills a fraction of the chip, so they overlap on the device when they sit on different streams.  Lanes = child
// contexts (stream + workspace + pinned block each); one host thread per lane walks its share of the windows with the
// ordinary solve (incl. the round-following logic), which keeps every lane's stream fed.
#include <thread>

// Grid mode: the B windows run as ONE launch sequence, blockIdx.z = window (BaWin, ba_common.h).  Every kernel of the
// local-window fast path has a batched entry point that takes its per-window arguments from a device array; a round's
// K5 is then B x items workgroups (instead of 157-250), K7 B x sets workgroups (instead of <= 3), and a launch is paid
// once per round, not once per window and round.  Eligible: windows of <= 64 cameras on the MFMA / LDS path (what a
// local window is).  Returns 1 when the batch is not eligible (the caller falls back to the lanes), 0 when it ran.
static int ba_solve_batch_grid(rs_context* ctx, int B, const rs_ba_problem* Q, const rs_ba_options* options, rs_ba_summary* out, int* rc_out)
{
    *rc_out = RS_OK;
    rs_ba_options def;
    if (!options) { rs_ba_default_options(&def); options = &def; }
    BaOpt opt;
    opt.max_iter = options->max_num_iterations; opt.max_invalid = options->max_num_consecutive_invalid_steps;
    opt.jacobi = options->jacobi_scaling; opt.r0 = options->initial_trust_region_radius;
    opt.rmax = options->max_trust_region_radius; opt.rmin = options->min_trust_region_radius;
    opt.min_rel = options->min_relative_decrease; opt.dmin = options->min_lm_diagonal;
    opt.dmax = options->max_lm_diagonal; opt.ftol = options->function_tolerance;
    opt.gtol = options->gradient_tolerance; opt.ptol = options->parameter_tolerance;
    if (opt.max_iter < 1 || opt.max_iter > 1000) return 1;
    int ns = ctx->ba_sets > 0 ? ctx->ba_sets : BA_CALIBRATED_SETS;      // (throughput mode: extra radii are CU time other windows want)
    if (ns > BA_CALIBRATED_SETS) ns = BA_CALIBRATED_SETS;
    if (ns > opt.max_iter) ns = opt.max_iter;
    std::vector<BaWin> wins((size_t)B);
    std::vector<size_t> ws_off((size_t)B), pin_off((size_t)B);
    size_t ws_total = align_up(sizeof(BaWin) * (size_t)B, 256), pin_total = 0;
    int max_P = 0, max_items = 0, max_n = 0, max_C = 0;
    size_t k5_lds = 0, k8_lds = 0;
    std::vector<BaLayout> lay((size_t)B);
    for (int i = 0; i < B; i++) {
        const rs_ba_problem& q = Q[i];
        if (q.n_cameras <= 0 || q.n_points <= 0 || q.n_obs <= 0 || q.n_cameras > 64) return 1;
        if (!q.d_cameras || !q.h_cam_free || !q.d_points || !q.d_obs_ptr || !q.d_obs_cam || !q.d_obs_uv) return 1;
        BaWin& w = wins[(size_t)i];
        memset(&w, 0, sizeof w);
        BaDims& d = w.d;
        d.C = q.n_cameras; d.P = q.n_points; d.M = q.n_obs;
        d.Cf = 0;
        unsigned long long mask = 0;
        for (int c = 0; c < d.C; c++) if (q.h_cam_free[c]) { mask |= 1ull << c; d.Cf++; }
        d.n = 6 * d.Cf;
        d.fx = q.intrinsics[0]; d.fy = q.intrinsics[1]; d.cx = q.intrinsics[2]; d.cy = q.intrinsics[3];
        d.huber_a = options->huber_delta;
        const bool use_mfma = d.Cf >= 1 && ba_schur_lds_bytes(d.C, d.Cf) <= 160 * 1024;
        if (!use_mfma || ba_backsub_lds_bytes(d.C, d.n) > 64 * 1024 || d.n < 6 || d.n > BA_MAX_LDS_N) return 1;
        w.free_mask = mask;
        const size_t C = (size_t)d.C;
        lay[(size_t)i] = ba_layout(d, ns, opt.max_iter, 1, ba_group_bytes(d.P, d.Cf, d.M));   // the single solve's layout, one copy per window
        const BaLayout& L = lay[(size_t)i];
        ws_off[(size_t)i] = ws_total;
        ws_total += L.bytes;
        pin_off[(size_t)i] = pin_total;
        pin_total += align_up(sizeof(BaState), 64) + 64 + align_up(sizeof(BaTrace) * (size_t)(opt.max_iter + 1), 64) + align_up(sizeof(double) * 6 * C, 64);
        max_P = std::max(max_P, d.P); max_n = std::max(max_n, d.n); max_C = std::max(max_C, d.C);
        k8_lds = std::max(k8_lds, ba_backsub_lds_bytes(d.C, d.n));
    }
    void* wsv = nullptr;
    int rc = rs_workspace(ctx, ws_total, &wsv);
    if (rc) { *rc_out = rc; return 0; }
    char* ws = (char*)wsv;
    void* pinv = nullptr;
    const size_t pin_wins = align_up(pin_total, 256);
    rc = rs_pinned(ctx, pin_wins + sizeof(BaWin) * (size_t)B, &pinv);
    if (rc) { *rc_out = rc; return 0; }
    char* pin = (char*)pinv;
    for (int i = 0; i < B; i++) {
        const rs_ba_problem& q = Q[i];
        BaWin& w = wins[(size_t)i];
        const BaLayout& L = lay[(size_t)i];
        char* base = ws + ws_off[(size_t)i];
        const BaDims& d = w.d;
        BaBufs& b = w.b;
        b.obs_ptr = q.d_obs_ptr; b.obs_cam = q.d_obs_cam; b.obs_uv = (const float2*)q.d_obs_uv;
        ba_bind(b, base, L, d, ns, 1, 0);
        ba_group_carve(base + L.grp, d.P, d.Cf, d.M, &w.g);
        ba_group_set_items(&w.g, d.P, true, ctx->ba_batch_item);
        b.obs_cs = w.g.obs_cs;
        max_items = std::max(max_items, w.g.n_items);
        k5_lds = std::max(k5_lds, ba_schur_lds_bytes(d.C, d.Cf, w.g.it_l, ns));
        w.st_base = b.st; w.pts_base = b.pt_scal; w.set_base = b.set_out; w.pts_block = L.pts_block;
        char* hp = pin + pin_off[(size_t)i];
        w.h_st = (BaState*)hp;
        w.prog = (BaProgress*)(hp + align_up(sizeof(BaState), 64));
        w.h_trace = (BaTrace*)(hp + align_up(sizeof(BaState), 64) + 64);
        w.h_cams = (double*)((char*)w.h_trace + align_up(sizeof(BaTrace) * (size_t)(opt.max_iter + 1), 64));
        w.prog->round = 0; w.prog->done = 0; w.prog->iter = 0;
        b.prog = w.prog;
        w.cams_in = q.d_cameras; w.pts_in = q.d_points; w.cams_out = q.d_cameras; w.pts_out = q.d_points;
        w.cam_free = (uint8_t*)(base + L.fre);
        ba_group_zero_range(w.g, &w.zero_ptr, &w.zero_n);
    }
    if (ba_prepare_reduced_solve_lds_batch(max_n) != 0 || ba_prepare_schur_batch(k5_lds) != 0) { *rc_out = rs_fail(ctx, RS_ERR_HIP, "LDS attribute (batch)"); return 0; }
    hipStream_t s = ctx->stream;
    BaWin* h_wins = (BaWin*)(pin + pin_wins);
    memcpy(h_wins, wins.data(), sizeof(BaWin) * (size_t)B);
    const BaWin* d_wins = (const BaWin*)ws;
    if (hipMemcpyAsync(ws, h_wins, sizeof(BaWin) * (size_t)B, hipMemcpyHostToDevice, s) != hipSuccess) { *rc_out = rs_fail(ctx, RS_ERR_HIP, "window table upload"); return 0; }
    {
        rs_prof_scope ps(ctx, "K0_ba_init");
        hipLaunchKernelGGL(ba_init_batch, dim3(32, 1, B), dim3(256), 0, s, d_wins, opt);
    }
    {
        rs_prof_scope ps(ctx, "K5s_group_landmarks");
        ba_launch_grouping_batch(s, d_wins, B, max_P, max_items);
    }
    auto enqueue_round = [&](int it) {
        { rs_prof_scope ps(ctx, "K5_ba_schur_mfma"); ba_launch_decide_batch(s, d_wins, B, opt, it); ba_launch_schur_batch(s, d_wins, B, opt, it, max_items, wins[0].g.it_l, k5_lds); }
        { rs_prof_scope ps(ctx, "K7_ba_reduced_solve"); ba_launch_reduced_solve_lds_batch(s, d_wins, B, opt, it, ns, max_n); }
        { rs_prof_scope ps(ctx, "K8_ba_backsub_cost"); ba_launch_backsub_batch(s, d_wins, B, it, ns, max_P, k8_lds); }
    };
    int rounds = 0;
    const int min_rounds = (opt.max_iter + ns - 1) / ns;
    for (; rounds < min_rounds; rounds++) enqueue_round(rounds);
    while (ns > 1 && rounds < opt.max_iter) {
        // as in the single solve, for the slowest window: wait until every window has started the last enqueued round
        long spins = 0;
        bool drained = false;
        for (int i = 0; i < B && !drained; i++)
            while (wins[(size_t)i].prog->round < rounds) {
                if ((++spins & 0xFFFFF) == 0 && hipStreamQuery(s) != hipErrorNotReady) { drained = true; break; }
            }
        if (drained) break;
        bool more = false;
        for (int i = 0; i < B; i++) {
            const BaProgress* pr = wins[(size_t)i].prog;
            if (!pr->done && opt.max_iter - pr->iter > 1) more = true;
        }
        if (!more) break;
        enqueue_round(rounds);
        rounds++;
    }
    {
        rs_prof_scope ps(ctx, "K10_ba_finalize");
        hipLaunchKernelGGL(ba_finalize_batch, dim3(16, 1, B), dim3(256), 0, s, d_wins, opt, rounds);
    }
    if (hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) { *rc_out = rs_fail(ctx, RS_ERR_HIP, "batched bundle adjustment"); return 0; }
    for (int i = 0; i < B; i++) {
        const BaState* h = wins[(size_t)i].h_st;
        rs_ba_summary& o = out[i];
        o.termination = h->termination; o.iterations = h->iter; o.successful_steps = h->successful; o.usable = h->usable;
        o.initial_cost = h->initial_cost; o.final_cost = h->x_cost; o.final_radius = h->radius;
    }
    ctx->ba_trace = nullptr; ctx->ba_trace_n = 0; ctx->ba_cams = nullptr; ctx->ba_cams_n = 0;
    ctx->ba_stats[0] = ctx->ba_stats[1] = ctx->ba_stats[2] = 0; ctx->ba_stats[3] = rounds;
    return 0;
}

extern "C" int rs_bundle_adjust_batch(rs_context* ctx, int n_problems, const rs_ba_problem* h_problems,
                                      const rs_ba_options* options, rs_ba_summary* h_summaries)
{
    if (!ctx || n_problems < 0 || (n_problems > 0 && (!h_problems || !h_summaries))) return RS_ERR_INVALID;
    if (n_problems == 0) return RS_OK;
    if (rs_comm_active(ctx)) return rs_fail(ctx, RS_ERR_UNSUPPORTED, "batch of windows on a landmark-sharded context");
    RS_HIP(ctx, hipSetDevice(ctx->device));
    for (int i = 0; i < n_problems; i++) memset(&h_summaries[i], 0, sizeof(rs_ba_summary));
    if (ctx->ba_batch_mode == 0 && n_problems > 1) {
        int rc = RS_OK;
        if (ba_solve_batch_grid(ctx, n_problems, h_problems, options, h_summaries, &rc) == 0) return rc;
    }
    const int lanes = n_problems < RS_BA_BATCH_LANES ? n_problems : RS_BA_BATCH_LANES;
    while ((int)ctx->batch_lanes.size() < lanes) {
        rs_context* c = nullptr;
        int rc = rs_context_create(ctx->device, &c);
        if (rc) return rs_fail(ctx, rc, "cannot create batch lane");
        hipStream_t st = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { rs_context_destroy(c); return rs_fail(ctx, RS_ERR_HIP, "hipStreamCreate"); }
        c->stream = st;
        c->ba_sets = ctx->ba_sets;
        ctx->batch_lanes.push_back(c);
        ctx->batch_streams.push_back(st);
    }
    // the inputs were produced on the parent's stream: the lanes start after it
    RS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<int> status((size_t)lanes, RS_OK);
    auto work = [&](int lane) {
        (void)hipSetDevice(ctx->device);
        rs_context* c = ctx->batch_lanes[(size_t)lane];
        c->ba_sets = ctx->ba_sets;
        for (int i = lane; i < n_problems; i += lanes) {
            const rs_ba_problem& q = h_problems[i];
            const int rc = rs_bundle_adjust(c, q.n_cameras, q.n_points, q.n_obs, q.d_cameras, q.h_cam_free, q.d_points, q.d_obs_ptr,
                                            q.d_obs_cam, q.d_obs_uv, q.intrinsics, options, &h_summaries[i]);
            if (rc && !status[(size_t)lane]) status[(size_t)lane] = rc;
        }
    };
    std::vector<std::thread> threads;
    for (int l = 1; l < lanes; l++) threads.emplace_back(work, l);
    work(0);
    for (auto& t : threads) t.join();
    // The lanes' finalize kernels raise their host flags while the device-side copies into d_cameras / d_points may still be
    // running on the LANE streams: the parent's stream (what the caller reads the results on, and what
    // rs_context_synchronize(ctx) covers) waits for every lane stream here, whatever the windows' status.
    const int wrc = rs_context_wait_for(ctx, ctx->batch_lanes.data(), lanes);
    for (int l = 0; l < lanes; l++)
        if (status[(size_t)l]) return rs_fail(ctx, status[(size_t)l], "window on lane %d failed: %s", l, rs_last_error(ctx->batch_lanes[(size_t)l]));
    return wrc;
}

static_assert(sizeof(BaTrace) == sizeof(rs_ba_iteration), "BaTrace mirrors rs_ba_iteration");

extern "C" int rs_ba_get_trace(rs_context* ctx, rs_ba_iteration* h_out, int capacity, int* h_count)
{
    if (!ctx || !h_count || capacity < 0 || (capacity > 0 && !h_out)) return RS_ERR_INVALID;
    const int n = ctx->ba_trace ? ctx->ba_trace_n : 0;
    const int m = n < capacity ? n : capacity;
    if (m > 0) memcpy(h_out, ctx->ba_trace, sizeof(rs_ba_iteration) * (size_t)m);
    *h_count = n;
    return RS_OK;
}

extern "C" int rs_ba_get_cameras(rs_context* ctx, double* h_cameras, int n_cameras)
{
    if (!ctx || !h_cameras || n_cameras < 0) return RS_ERR_INVALID;
    if (!ctx->ba_cams || n_cameras != ctx->ba_cams_n) return rs_fail(ctx, RS_ERR_INVALID, "no bundle adjustment result of %d cameras on this context", n_cameras);
    memcpy(h_cameras, ctx->ba_cams, sizeof(double) * 6 * (size_t)n_cameras);
    return RS_OK;
}

extern "C" int rs_ba_get_stats(rs_context* ctx, int h_out[8])
{
    if (!ctx || !h_out) return RS_ERR_INVALID;
    for (int i = 0; i < 8; i++) h_out[i] = ctx->ba_stats[i];
    return RS_OK;
}

// ------------------------------------------------------------- refine_pose
// optimization::refine_pose (src/Optimization.cpp:194-267), vision-only: one
// camera block, points constant.  The whole LM loop runs inside ONE launch of
// a single workgroup (6 unknowns; <= 2000 residual pairs): per iteration a
// block reduction of the 6x6 normal equations, a register Cholesky on lane 0
// and a second reduction for the candidate cost.
#define RP_THREADS 512         // 2000 observations: four per thread; 256 VGPRs per thread keep the 28 accumulators in registers

__device__ __forceinline__ void block_sum(double* vals, int count, double* scratch /*[RP_THREADS / 64 + 1][32]*/)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = 0; k < count; k++) {
        const double v = wave_sum_lane63(vals[k]);
        if (lane == 63) scratch[wave * 32 + k] = v;
    }
    __syncthreads();
    if ((int)threadIdx.x < count) {                 // thread k folds the waves' partial sums of value k (fixed order)
        double t = 0.0;
        for (int w = 0; w < RP_THREADS / 64; w++) t += scratch[w * 32 + threadIdx.x];
        scratch[(RP_THREADS / 64) * 32 + threadIdx.x] = t;
    }
    __syncthreads();
    for (int k = 0; k < count; k++) vals[k] = scratch[(RP_THREADS / 64) * 32 + k];
    __syncthreads();
}

__global__ __launch_bounds__(RP_THREADS) void ba_refine_pose(BaDims d, BaOpt opt, const double* __restrict__ pts,
                                                            const float2* __restrict__ uv, int n,
                                                            double* __restrict__ cam_io, BaState* __restrict__ st_out, volatile int* host_done)
{
    __shared__ double x[6], xn[6], prep[BA_PREP], prepn[BA_PREP], scratch[(RP_THREADS / 64 + 1) * 32];
    __shared__ BaState st;
    __shared__ double sc[6];
    const int tid = threadIdx.x;
    if (tid == 0) {
        for (int k = 0; k < 6; k++) x[k] = cam_io[k];
        cam_prepare(x, prep);
        st.radius = opt.r0; st.decrease_factor = 2.0; st.x_cost = 0.0; st.initial_cost = 0.0;
        st.iter = 0; st.successful = 0; st.invalid_steps = 0; st.done = 0; st.termination = 0; st.cur = 0;
        st.have_scale = 0; st.solver_failed = 0; st.fresh = 1; st.usable = 0; st.consec_accepts = 0; st.nact = 1;
        st.n_rounds = 0; st.n_fresh = 0; st.n_sets = 0; st.hand_lost = 0;
    }
    __syncthreads();
    double acc[28];
    while (true) {
        // linearise at x (recomputed after a rejected step too: same values)
        for (int k = 0; k < 28; k++) acc[k] = 0.0;
        ObsLin o;
        for (int i = tid; i < n; i += RP_THREADS) {
            const double X[3] = {pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2]};
            obs_eval<true>(prep, X, uv[i], d, o);
            int q = 0;
#pragma unroll
            for (int a = 0; a < 6; a++) {
#pragma unroll
                for (int e = a; e < 6; e++) acc[q++] += o.w * (o.jc[a] * o.jc[e] + o.jc[6 + a] * o.jc[6 + e]);
            }
#pragma unroll
            for (int a = 0; a < 6; a++) acc[21 + a] += o.w * (o.jc[a] * o.r0 + o.jc[6 + a] * o.r1);
            acc[27] += 0.5 * o.rho;
        }
        block_sum(acc, 28, scratch);
        if (tid == 0) {
            double H[6][6], g[6], lam[6], dlt[6];
            int q = 0;
            for (int a = 0; a < 6; a++)
                for (int e = a; e < 6; e++) { H[a][e] = acc[q]; H[e][a] = acc[q]; q++; }
            for (int a = 0; a < 6; a++) g[a] = acc[21 + a];
            if (st.fresh) {
                st.x_cost = acc[27];
                if (st.iter == 0) st.initial_cost = st.x_cost;
                if (!st.have_scale)
                    for (int a = 0; a < 6; a++) sc[a] = opt.jacobi ? 1.0 / (1.0 + sqrt(H[a][a])) : 1.0;
                double gm = 0.0;
                for (int a = 0; a < 6; a++) gm = fmax(gm, fabs(g[a]));
                if (!isfinite(st.x_cost)) { st.done = 1; st.termination = RS_BA_FAILURE; }
                else if (gm <= opt.gtol) { st.done = 1; st.termination = RS_BA_CONVERGENCE_GRADIENT; }
            }
            if (!st.done && st.iter >= opt.max_iter) { st.done = 1; st.termination = RS_BA_NO_CONVERGENCE; }
            if (!st.done) {
                bool fail = false;
                for (int a = 0; a < 6; a++) {
                    const double s2 = sc[a] * sc[a];
                    lam[a] = clampd(s2 * H[a][a], opt.dmin, opt.dmax) / (st.radius * s2);
                    H[a][a] += lam[a];
                }
                // Cholesky 6x6 (one reciprocal per column instead of a division per entry: this is one lane's serial code)
                double rdiag[6];
                for (int j = 0; j < 6 && !fail; j++) {
                    double dj = H[j][j];
                    for (int k = 0; k < j; k++) dj -= H[j][k] * H[j][k];
                    if (!(dj > 0.0) || !isfinite(dj)) { fail = true; break; }
                    dj = sqrt(dj);
                    H[j][j] = dj;
                    rdiag[j] = 1.0 / dj;
                    for (int i = j + 1; i < 6; i++) {
                        double s = H[i][j];
                        for (int k = 0; k < j; k++) s -= H[i][k] * H[j][k];
                        H[i][j] = s * rdiag[j];
                    }
                }
                if (!fail) {
                    for (int i = 0; i < 6; i++) {
                        double s = g[i];
                        for (int k = 0; k < i; k++) s -= H[i][k] * dlt[k];
                        dlt[i] = s * rdiag[i];
                    }
                    for (int i = 5; i >= 0; i--) {
                        double s = dlt[i];
                        for (int k = i + 1; k < 6; k++) s -= H[k][i] * dlt[k];
                        dlt[i] = s * rdiag[i];
                    }
                    double mcc = 0.0, ssq = 0.0, xsq = 0.0;
                    for (int a = 0; a < 6; a++) {
                        dlt[a] = -dlt[a];
                        if (!isfinite(dlt[a])) fail = true;
                        mcc += 0.5 * (dlt[a] * dlt[a] * lam[a] - dlt[a] * g[a]);
                        xn[a] = x[a] + dlt[a];
                        ssq += (x[a] - xn[a]) * (x[a] - xn[a]);
                        xsq += x[a] * x[a];
                    }
                    st.cam_scal[0] = mcc; st.cam_scal[1] = ssq; st.cam_scal[2] = xsq;
                    cam_prepare(xn, prepn);
                }
                st.solver_failed = fail ? 1 : 0;
            }
        }
        __syncthreads();
        if (st.done) break;
        double cc[1] = {0.0};
        if (!st.solver_failed) {
            for (int i = tid; i < n; i += RP_THREADS) {
                const double X[3] = {pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2]};
                obs_eval<false>(prepn, X, uv[i], d, o);
                cc[0] += 0.5 * o.rho;
            }
        }
        block_sum(cc, 1, scratch);
        if (tid == 0) {
            st.iter++;
            const double cand = cc[0], mcc = st.cam_scal[0];
            st.fresh = 0;
            if (st.solver_failed || !(mcc > 0.0)) {
                if (++st.invalid_steps >= opt.max_invalid) { st.done = 1; st.termination = RS_BA_FAILURE; }
                else { st.radius /= st.decrease_factor; st.decrease_factor *= 2.0; }
            } else {
                st.invalid_steps = 0;
                const double step_norm = sqrt(st.cam_scal[1]), x_norm = sqrt(st.cam_scal[2]);
                if (step_norm <= opt.ptol * (x_norm + opt.ptol)) { st.done = 1; st.termination = RS_BA_CONVERGENCE_PARAMETER; }
                else if (fabs(st.x_cost - cand) <= opt.ftol * st.x_cost) { st.done = 1; st.termination = RS_BA_CONVERGENCE_FUNCTION; }
                else {
                    const double rel = (st.x_cost - cand) / mcc;
                    if (rel > opt.min_rel && isfinite(cand)) {
                        for (int a = 0; a < 6; a++) x[a] = xn[a];
                        for (int a = 0; a < BA_PREP; a++) prep[a] = prepn[a];
                        st.successful++;
                        const double t = 2.0 * rel - 1.0;
                        st.radius = fmin(opt.rmax, st.radius / fmax(1.0 / 3.0, 1.0 - t * t * t));
                        st.decrease_factor = 2.0;
                        st.fresh = 1;
                        st.x_cost = cand;
                    } else {
                        st.radius /= st.decrease_factor;
                        st.decrease_factor *= 2.0;
                        if (st.radius < opt.rmin) { st.done = 1; st.termination = RS_BA_CONVERGENCE_RADIUS; }
                    }
                }
            }
            st.solver_failed = 0;
            st.have_scale = 1;
        }
        __syncthreads();
        if (st.done) break;
    }
    if (tid == 0) {
        st.usable = (st.termination != RS_BA_FAILURE && isfinite(st.x_cost) && st.x_cost <= st.initial_cost) ? 1 : 0;
        if (st.usable)
            for (int k = 0; k < 6; k++) cam_io[k] = x[k];
        *st_out = st;
        __threadfence_system();           // cam_io / st_out are pinned host memory
        *host_done = 1;                   // the host spins on this instead of synchronising the stream
        __threadfence_system();
    }
}

extern "C" int rs_refine_pose(rs_context* ctx, double h_camera[6], const double* d_points, const float* d_uv, int n,
                              const float h_intrinsics[4], const rs_ba_options* options, rs_ba_summary* h_summary)
{
    if (!ctx || !h_summary || !h_camera) return RS_ERR_INVALID;
    memset(h_summary, 0, sizeof *h_summary);
    if (n < 0) return rs_fail(ctx, RS_ERR_INVALID, "negative n");
    if (n == 0) return RS_OK;   // "nothing to constrain", src/Optimization.cpp:227-229
    if (!d_points || !d_uv || !h_intrinsics) return rs_fail(ctx, RS_ERR_INVALID, "null pointer");
    rs_ba_options def;
    if (!options) { rs_ba_default_options(&def); options = &def; }
    RS_HIP(ctx, hipSetDevice(ctx->device));
    BaDims d;
    d.C = 1; d.Cf = 1; d.P = n; d.M = n; d.n = 6;
    d.fx = h_intrinsics[0]; d.fy = h_intrinsics[1]; d.cx = h_intrinsics[2]; d.cy = h_intrinsics[3];
    d.huber_a = options->huber_delta;
    BaOpt opt;
    opt.max_iter = options->max_num_iterations; opt.max_invalid = options->max_num_consecutive_invalid_steps;
    opt.jacobi = options->jacobi_scaling; opt.r0 = options->initial_trust_region_radius;
    opt.rmax = options->max_trust_region_radius; opt.rmin = options->min_trust_region_radius;
    opt.min_rel = options->min_relative_decrease; opt.dmin = options->min_lm_diagonal;
    opt.dmax = options->max_lm_diagonal; opt.ftol = options->function_tolerance;
    opt.gtol = options->gradient_tolerance; opt.ptol = options->parameter_tolerance;
    void* wsv = nullptr;
    int rc = rs_workspace(ctx, 1024, &wsv);
    if (rc) return rc;
    (void)wsv;
    void* pin = nullptr;
    rc = rs_pinned(ctx, 512, &pin);
    if (rc) return rc;
    ctx->ba_trace_n = 0;                 // the pinned block is reused: the last BA's record is gone
    ctx->ba_cams = nullptr;
    ctx->ba_cams_n = 0;
    double* h_cam = (double*)pin;
    BaState* h_st = (BaState*)((char*)pin + 256);
    memcpy(h_cam, h_camera, 6 * sizeof(double));
    hipStream_t s = ctx->stream;
    {
        // the kernel reads the camera from and writes camera + state to the PINNED block itself: no copy launches around
        // a 40 us kernel (three hipMemcpyAsync cost more than the solve)
        rs_prof_scope ps(ctx, "K11_refine_pose");
        volatile int* h_done = (volatile int*)((char*)pin + 448);
        *h_done = 0;
        hipLaunchKernelGGL(ba_refine_pose, dim3(1), dim3(RP_THREADS), 0, s, d, opt, d_points, (const float2*)d_uv, n, h_cam, h_st, h_done);
        RS_HIP(ctx, hipGetLastError());
        long spins = 0;
        while (*h_done != 1)
            if ((++spins & 0x3FFFF) == 0 && hipStreamQuery(s) != hipErrorNotReady) break;
        if (*h_done != 1) RS_HIP(ctx, hipStreamSynchronize(s));
    }
    RS_HIP(ctx, hipGetLastError());
    if (h_st->usable) memcpy(h_camera, h_cam, 6 * sizeof(double));
    h_summary->termination = h_st->termination;
    h_summary->iterations = h_st->iter;
    h_summary->successful_steps = h_st->successful;
    h_summary->usable = h_st->usable;
    h_summary->initial_cost = h_st->initial_cost;
    h_summary->final_cost = h_st->x_cost;
    h_summary->final_radius = h_st->radius;
    return RS_OK;
}

// ------------------------------------------------- refine_pose with an InertialConstraint
// optimization::refine_pose (src/Optimization.cpp:194-267) with a RotationPrior (:252-258: 3 more residuals on the
// pose block) or an InertialDelta (:237-251: the 9-residual preintegration block with the previous frame constant and
// this frame's velocity as a second free block -> 9 unknowns).  Same single-launch LM as ba_refine_pose; thread 0
// evaluates the one extra block with dual numbers and solves the nu x nu system (nu = 6 or 9).
struct RpInertial {
    int kind;                 // 1 rotation prior, 2 inertial delta
    double predicted[9], sigma;
    double prev_pose[6], prev_vel[3], prev_bias[6], gravity[3];
    ImuFactorDev fac;
};

__global__ __launch_bounds__(RP_THREADS) void ba_refine_pose_inertial(BaDims d, BaOpt opt, const double* __restrict__ pts,
                                                                     const float2* __restrict__ uv, int n,
                                                                     const RpInertial* __restrict__ ext,
                                                                     double* __restrict__ cam_io /*[9]: pose, velocity*/,
                                                                     BaState* __restrict__ st_out, volatile int* host_done)
{
    __shared__ double x[9], xn[9], prep[BA_PREP], prepn[BA_PREP], scratch[(RP_THREADS / 64 + 1) * 32];
    __shared__ double s_extr[9], s_extJ[9][IMU_NP];
    __shared__ BaState st;
    __shared__ double sc[9];
    const int tid = threadIdx.x;
    const int nu = ext->kind == 2 ? 9 : 6;
    if (tid == 0) {
        for (int k = 0; k < 9; k++) x[k] = cam_io[k];
        cam_prepare(x, prep);
        st.radius = opt.r0; st.decrease_factor = 2.0; st.x_cost = 0.0; st.initial_cost = 0.0;
        st.iter = 0; st.successful = 0; st.invalid_steps = 0; st.done = 0; st.termination = 0; st.cur = 0;
        st.have_scale = 0; st.solver_failed = 0; st.fresh = 1; st.usable = 0; st.consec_accepts = 0; st.nact = 1;
        st.n_rounds = 0; st.n_fresh = 0; st.n_sets = 0; st.hand_lost = 0;
    }
    __syncthreads();
    double acc[28];
    while (true) {
        for (int k = 0; k < 28; k++) acc[k] = 0.0;
        ObsLin o;
        for (int i = tid; i < n; i += RP_THREADS) {
            const double X[3] = {pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2]};
            obs_eval<true>(prep, X, uv[i], d, o);
            int q = 0;
#pragma unroll
            for (int a = 0; a < 6; a++) {
#pragma unroll
                for (int e = a; e < 6; e++) acc[q++] += o.w * (o.jc[a] * o.jc[e] + o.jc[6 + a] * o.jc[6 + e]);
            }
#pragma unroll
            for (int a = 0; a < 6; a++) acc[21 + a] += o.w * (o.jc[a] * o.r0 + o.jc[6 + a] * o.r1);
            acc[27] += 0.5 * o.rho;
        }
        // the extra residual block at x: the first 32 lanes evaluate it with one partial per lane (imu_dual.h) and
        // leave residual + Jacobian in LDS for the solving thread
        if (tid < 32) {
            if (ext->kind == 1) {
                double r[3], jl[3];
                imu_rotation_prior_lanes(ext->predicted, ext->sigma, x, r, jl);
                for (int a = 0; a < 3; a++) { if (tid < 3) s_extJ[a][tid] = jl[a]; if (tid == 0) s_extr[a] = r[a]; }
            } else {
                double r[9], jl[9];
                imu_preintegration_lanes(ext->fac, ext->gravity, ext->prev_pose, ext->prev_vel, ext->prev_bias, x, x + 6, r, jl);
                for (int a = 0; a < 9; a++) { if (tid < IMU_NP) s_extJ[a][tid] = jl[a]; if (tid == 0) s_extr[a] = r[a]; }
            }
        }
        block_sum(acc, 28, scratch);
        if (tid == 0) {
            double H[9][9], gv[9], lam[9], dlt[9];
            for (int a = 0; a < 9; a++) { gv[a] = 0.0; for (int e = 0; e < 9; e++) H[a][e] = 0.0; }
            int q = 0;
            for (int a = 0; a < 6; a++)
                for (int e = a; e < 6; e++) { H[a][e] = acc[q]; H[e][a] = acc[q]; q++; }
            for (int a = 0; a < 6; a++) gv[a] = acc[21 + a];
            double cost = acc[27];
            if (ext->kind == 1) {
                for (int a = 0; a < 3; a++) {
                    cost += 0.5 * s_extr[a] * s_extr[a];
                    for (int k = 0; k < 3; k++) { gv[k] += s_extJ[a][k] * s_extr[a]; for (int l = 0; l < 3; l++) H[k][l] += s_extJ[a][k] * s_extJ[a][l]; }
                }
            } else {
                for (int a = 0; a < 9; a++) {
                    cost += 0.5 * s_extr[a] * s_extr[a];
                    for (int k = 0; k < 9; k++) {            // local parameters 15..23 = pose_j (6), velocity_j (3)
                        const double jk = s_extJ[a][15 + k];
                        gv[k] += jk * s_extr[a];
                        for (int l = 0; l < 9; l++) H[k][l] += jk * s_extJ[a][15 + l];
                    }
                }
            }
            if (st.fresh) {
                st.x_cost = cost;
                if (st.iter == 0) st.initial_cost = st.x_cost;
                if (!st.have_scale)
                    for (int a = 0; a < nu; a++) sc[a] = opt.jacobi ? 1.0 / (1.0 + sqrt(H[a][a])) : 1.0;
                double gm = 0.0;
                for (int a = 0; a < nu; a++) gm = fmax(gm, fabs(gv[a]));
                if (!isfinite(st.x_cost)) { st.done = 1; st.termination = RS_BA_FAILURE; }
                else if (gm <= opt.gtol) { st.done = 1; st.termination = RS_BA_CONVERGENCE_GRADIENT; }
            }
            if (!st.done && st.iter >= opt.max_iter) { st.done = 1; st.termination = RS_BA_NO_CONVERGENCE; }
            if (!st.done) {
                bool fail = false;
                for (int a = 0; a < nu; a++) {
                    const double s2 = sc[a] * sc[a];
                    lam[a] = clampd(s2 * H[a][a], opt.dmin, opt.dmax) / (st.radius * s2);
                    H[a][a] += lam[a];
                }
                for (int j = 0; j < nu && !fail; j++) {     // Cholesky nu x nu
                    double dj = H[j][j];
                    for (int k = 0; k < j; k++) dj -= H[j][k] * H[j][k];
                    if (!(dj > 0.0) || !isfinite(dj)) { fail = true; break; }
                    dj = sqrt(dj);
                    H[j][j] = dj;
                    for (int i = j + 1; i < nu; i++) {
                        double t = H[i][j];
                        for (int k = 0; k < j; k++) t -= H[i][k] * H[j][k];
                        H[i][j] = t / dj;
                    }
                }
                if (!fail) {
                    for (int i = 0; i < nu; i++) {
                        double t = gv[i];
                        for (int k = 0; k < i; k++) t -= H[i][k] * dlt[k];
                        dlt[i] = t / H[i][i];
                    }
                    for (int i = nu - 1; i >= 0; i--) {
                        double t = dlt[i];
                        for (int k = i + 1; k < nu; k++) t -= H[k][i] * dlt[k];
                        dlt[i] = t / H[i][i];
                    }
                    double mcc = 0.0, ssq = 0.0, xsq = 0.0;
                    for (int a = 0; a < 9; a++) xn[a] = x[a];
                    for (int a = 0; a < nu; a++) {
                        dlt[a] = -dlt[a];
                        if (!isfinite(dlt[a])) fail = true;
                        mcc += 0.5 * (dlt[a] * dlt[a] * lam[a] - dlt[a] * gv[a]);
                        xn[a] = x[a] + dlt[a];
                        ssq += (x[a] - xn[a]) * (x[a] - xn[a]);
                        xsq += x[a] * x[a];
                    }
                    st.cam_scal[0] = mcc; st.cam_scal[1] = ssq; st.cam_scal[2] = xsq;
                    cam_prepare(xn, prepn);
                    // the extra block at the candidate
                    double ce = 0.0;
                    if (!fail) {
                        if (ext->kind == 1) {
                            double r[3];
                            imu_rotation_prior(ext->predicted, ext->sigma, xn, r, nullptr);
                            for (int a = 0; a < 3; a++) ce += 0.5 * r[a] * r[a];
                        } else {
                            double r[9];
                            imu_preintegration(ext->fac, ext->gravity, ext->prev_pose, ext->prev_vel, ext->prev_bias, xn, xn + 6, r, nullptr);
                            for (int a = 0; a < 9; a++) ce += 0.5 * r[a] * r[a];
                        }
                    }
                    st.cam_scal[3] = ce;
                }
                st.solver_failed = fail ? 1 : 0;
            }
        }
        __syncthreads();
        if (st.done) break;
        double cc[1] = {0.0};
        if (!st.solver_failed) {
            for (int i = tid; i < n; i += RP_THREADS) {
                const double X[3] = {pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2]};
                obs_eval<false>(prepn, X, uv[i], d, o);
                cc[0] += 0.5 * o.rho;
            }
        }
        block_sum(cc, 1, scratch);
        if (tid == 0) {
            st.iter++;
            const double cand = cc[0] + st.cam_scal[3], mcc = st.cam_scal[0];
            st.fresh = 0;
            if (st.solver_failed || !(mcc > 0.0)) {
                if (++st.invalid_steps >= opt.max_invalid) { st.done = 1; st.termination = RS_BA_FAILURE; }
                else { st.radius /= st.decrease_factor; st.decrease_factor *= 2.0; }
            } else {
                st.invalid_steps = 0;
                const double step_norm = sqrt(st.cam_scal[1]), x_norm = sqrt(st.cam_scal[2]);
                if (step_norm <= opt.ptol * (x_norm + opt.ptol)) { st.done = 1; st.termination = RS_BA_CONVERGENCE_PARAMETER; }
                else if (fabs(st.x_cost - cand) <= opt.ftol * st.x_cost) { st.done = 1; st.termination = RS_BA_CONVERGENCE_FUNCTION; }
                else {
                    const double rel = (st.x_cost - cand) / mcc;
                    if (rel > opt.min_rel && isfinite(cand)) {
                        for (int a = 0; a < 9; a++) x[a] = xn[a];
                        for (int a = 0; a < BA_PREP; a++) prep[a] = prepn[a];
                        st.successful++;
                        const double t = 2.0 * rel - 1.0;
                        st.radius = fmin(opt.rmax, st.radius / fmax(1.0 / 3.0, 1.0 - t * t * t));
                        st.decrease_factor = 2.0;
                        st.fresh = 1;
                        st.x_cost = cand;
                    } else {
                        st.radius /= st.decrease_factor;
                        st.decrease_factor *= 2.0;
                        if (st.radius < opt.rmin) { st.done = 1; st.termination = RS_BA_CONVERGENCE_RADIUS; }
                    }
                }
            }
            st.solver_failed = 0;
            st.have_scale = 1;
        }
        __syncthreads();
        if (st.done) break;
    }
    if (tid == 0) {
        st.usable = (st.termination != RS_BA_FAILURE && isfinite(st.x_cost) && st.x_cost <= st.initial_cost) ? 1 : 0;
        if (st.usable)
            for (int k = 0; k < 9; k++) cam_io[k] = x[k];
        *st_out = st;
        __threadfence_system();           // cam_io / st_out are pinned host memory
        *host_done = 1;                   // the host spins on this instead of synchronising the stream
        __threadfence_system();
    }
}

extern "C" int rs_refine_pose_inertial(rs_context* ctx, double h_camera[6], const double* d_points, const float* d_uv, int n,
                                       const float h_intrinsics[4], int kind, const double h_predicted[9], double sigma_radians,
                                       const double h_prev_pose[6], const double h_prev_velocity[3], const double h_prev_bias[6],
                                       const rs_imu_factor* h_delta, const double h_gravity[3], double h_velocity[3],
                                       const rs_ba_options* options, rs_ba_summary* h_summary)
{
    if (!ctx || !h_summary || !h_camera) return RS_ERR_INVALID;
    if (kind < 0 || kind > 2) return rs_fail(ctx, RS_ERR_INVALID, "kind must be 0, 1 or 2");
    // RotationPrior::enabled / InertialDelta::enabled (src/Optimization.h:50-53,60-63): a disabled constraint is no constraint
    if (kind == 1 && (!h_predicted || !(sigma_radians > 0.0))) kind = 0;
    if (kind == 2 && (!h_delta || !(h_delta->duration > 0.0))) kind = 0;
    if (kind == 0) return rs_refine_pose(ctx, h_camera, d_points, d_uv, n, h_intrinsics, options, h_summary);
    memset(h_summary, 0, sizeof *h_summary);
    if (n < 0) return rs_fail(ctx, RS_ERR_INVALID, "negative n");
    if (n == 0) return RS_OK;   // "nothing to constrain", src/Optimization.cpp:227-229 (checked before the inertial block is added)
    if (!d_points || !d_uv || !h_intrinsics) return rs_fail(ctx, RS_ERR_INVALID, "null pointer");
    if (kind == 2 && (!h_prev_pose || !h_prev_velocity || !h_prev_bias || !h_gravity || !h_velocity))
        return rs_fail(ctx, RS_ERR_INVALID, "null pointer (inertial delta)");
    rs_ba_options def;
    if (!options) { rs_ba_default_options(&def); options = &def; }
    RS_HIP(ctx, hipSetDevice(ctx->device));
    BaDims d;
    d.C = 1; d.Cf = 1; d.P = n; d.M = n; d.n = 6;
    d.fx = h_intrinsics[0]; d.fy = h_intrinsics[1]; d.cx = h_intrinsics[2]; d.cy = h_intrinsics[3];
    d.huber_a = options->huber_delta;
    BaOpt opt;
    opt.max_iter = options->max_num_iterations; opt.max_invalid = options->max_num_consecutive_invalid_steps;
    opt.jacobi = options->jacobi_scaling; opt.r0 = options->initial_trust_region_radius;
    opt.rmax = options->max_trust_region_radius; opt.rmin = options->min_trust_region_radius;
    opt.min_rel = options->min_relative_decrease; opt.dmin = options->min_lm_diagonal;
    opt.dmax = options->max_lm_diagonal; opt.ftol = options->function_tolerance;
    opt.gtol = options->gradient_tolerance; opt.ptol = options->parameter_tolerance;
    const size_t ext_off = 512, ws_bytes = ext_off + ((sizeof(RpInertial) + 255) & ~(size_t)255);
    void* wsv = nullptr;
    int rc = rs_workspace(ctx, ws_bytes, &wsv);
    if (rc) return rc;
    RpInertial* d_ext = (RpInertial*)((char*)wsv + ext_off);
    void* pin = nullptr;
    rc = rs_pinned(ctx, 1024 + sizeof(RpInertial), &pin);
    if (rc) return rc;
    ctx->ba_trace_n = 0;
    ctx->ba_cams = nullptr;
    ctx->ba_cams_n = 0;
    double* h_cam = (double*)pin;
    BaState* h_st = (BaState*)((char*)pin + 256);
    RpInertial* h_ext = (RpInertial*)((char*)pin + 1024);
    memset(h_ext, 0, sizeof *h_ext);
    h_ext->kind = kind;
    if (kind == 1) {
        memcpy(h_ext->predicted, h_predicted, sizeof h_ext->predicted);
        h_ext->sigma = sigma_radians;
    } else {
        memcpy(h_ext->prev_pose, h_prev_pose, sizeof h_ext->prev_pose);
        memcpy(h_ext->prev_vel, h_prev_velocity, sizeof h_ext->prev_vel);
        memcpy(h_ext->prev_bias, h_prev_bias, sizeof h_ext->prev_bias);
        memcpy(h_ext->gravity, h_gravity, sizeof h_ext->gravity);
        h_ext->fac.f = *h_delta;
        imu_whitener(h_delta->covariance, h_ext->fac.W);
    }
    memcpy(h_cam, h_camera, 6 * sizeof(double));
    for (int k = 0; k < 3; k++) h_cam[6 + k] = (kind == 2) ? h_velocity[k] : 0.0;
    hipStream_t s = ctx->stream;
    RS_HIP(ctx, hipMemcpyAsync(d_ext, h_ext, sizeof(RpInertial), hipMemcpyHostToDevice, s));   // read in the inner loops: device memory
    {
        rs_prof_scope ps(ctx, "K11_refine_pose_inertial");
        volatile int* h_done = (volatile int*)((char*)pin + 448);
        *h_done = 0;
        hipLaunchKernelGGL(ba_refine_pose_inertial, dim3(1), dim3(RP_THREADS), 0, s, d, opt, d_points, (const float2*)d_uv, n,
                           (const RpInertial*)d_ext, h_cam, h_st, h_done);                    // camera / state: the pinned block itself
        RS_HIP(ctx, hipGetLastError());
        long spins = 0;
        while (*h_done != 1)
            if ((++spins & 0x3FFFF) == 0 && hipStreamQuery(s) != hipErrorNotReady) break;
        if (*h_done != 1) RS_HIP(ctx, hipStreamSynchronize(s));
    }
    RS_HIP(ctx, hipGetLastError());
    if (h_st->usable) {
        memcpy(h_camera, h_cam, 6 * sizeof(double));
        if (kind == 2) memcpy(h_velocity, h_cam + 6, 3 * sizeof(double));    // unpack_inertial, :263-265
    }
    h_summary->termination = h_st->termination;
    h_summary->iterations = h_st->iter;
    h_summary->successful_steps = h_st->successful;
    h_summary->usable = h_st->usable;
    h_summary->initial_cost = h_st->initial_cost;
    h_summary->final_cost = h_st->x_cost;
    h_summary->final_radius = h_st->radius;
    return RS_OK;
}
