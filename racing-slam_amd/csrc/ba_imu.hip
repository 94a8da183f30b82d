// ba_imu.hip — inertial bundle adjustment on the LOCAL-WINDOW kernels (K5 MFMA Schur, LDS K7, K8): the velocity / bias
// blocks of the IMU factors (reference src/Optimization.cpp:317-346, src/ImuFactor.cpp:19-118) are eliminated in one
// workgroup between K5 and K7, so that K7 still solves the 6 Cf x 6 Cf pose system it was built for.
//
//   unknowns      poses p (6 per optimised camera), z_c = (velocity 3 | bias 6) per inertial camera, landmarks
//   K5            eliminates the landmarks as always:  S_pp (vision), rhs, U, gc
//   K6i (here)    linearises the IMU factor pairs (one 32-lane group per pair, one dual partial per lane, imu_dual.h):
//                   H_pp  -> U (same camera) / S (pose_i x pose_j),  g_p -> gc
//                   H_zz  block tridiagonal (a factor joins consecutive inertial cameras) in LDS,  H_zp, g_z -> G
//                 damps z like every Ceres parameter (Jacobi scale from the first linearisation, clamp, / radius),
//                 factors A_zz = L D L^T block by block, solves  Y = A_zz^-1 [H_zp | g_z]  and adds the Schur terms
//                   S -= H_pz Y_p,   rhs -= H_pz y_g
//                 The cost / gradient-maximum of the inertial blocks go into K5's slot lines, so K7 sees totals.
//   K7            unchanged: (U + Lambda_p + S) x_p = gc + rhs, candidate cameras, pose part of the step scalars
//   K7i (here)    x_z = y_g - Y_p x_p: candidate velocities / biases, their part of the model cost change / step norm /
//                 x norm, and the inertial blocks' cost at the candidate  -> BaState::cam_scal
//   K8            unchanged.
// Same LM schedule as the N x N blocked solve (ba_solve_big.hip) it replaces for windows of at most 21 optimised
// cameras: 14 launches per round become 5.
#include "ba_common.h"
#include "imu_dual.h"

#define IMU_MAXCI 21            // inertial cameras (<= optimised cameras of a window the LDS K7 takes)
#define KI_THREADS 512

__device__ __forceinline__ double imu_rl64(double v, int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double imu_rcp(double p)
{
    double r = __builtin_amdgcn_rcp(p);
    r = fma(fma(-p, r, 1.0), r, r);
    r = fma(fma(-p, r, 1.0), r, r);
    return r;
}

// G [9 Ci][n + 1] (row z: H_zp | g_z, becomes Y) and G0 [9 Ci][n] (H_zp kept) live in the N x N buffer of the blocked path
__device__ __forceinline__ double* imu_G(const BaBufs& b) { return b.imu.A; }
__device__ __forceinline__ double* imu_G0(const BaBufs& b, int n) { return b.imu.A + (size_t)9 * b.imu.Ci * (n + 1); }

__global__ __launch_bounds__(KI_THREADS) void ba_imu_eliminate(BaDims d, BaBufs b, BaOpt opt)
{
    __shared__ double Hd[IMU_MAXCI * 81];       // diagonal blocks of A_zz, then their L (unit lower) / D
    __shared__ double Ho[IMU_MAXCI * 81];       // block (q + 1, q), then L_{q+1,q}
    __shared__ double gz[IMU_MAXCI * 9], dinv[IMU_MAXCI * 9];
    __shared__ BaState st;
    __shared__ double s_cost, s_red[KI_THREADS / 64];
    __shared__ int s_fail;
    __shared__ int q_of_slot[IMU_MAXCI + 1];    // inertial slot of the camera that owns pose slot s, or -1
    const int n = d.n, Ci = b.imu.Ci, NZ = 9 * Ci, tid = threadIdx.x, nt = KI_THREADS;
    const int lane = tid & 63, wave = tid >> 6;
    double* G = imu_G(b);
    double* G0 = imu_G0(b, n);
    if (tid == 0) { st = *b.st; s_cost = 0.0; s_fail = 0; }
    for (int i = tid; i < Ci * 81; i += nt) { Hd[i] = 0.0; Ho[i] = 0.0; }
    for (int i = tid; i < NZ; i += nt) gz[i] = 0.0;
    if (tid <= IMU_MAXCI) q_of_slot[tid] = -1;
    for (int i = tid; i < NZ * (n + 1); i += nt) G[i] = 0.0;
    for (int i = tid; i < NZ * n; i += nt) G0[i] = 0.0;
    __syncthreads();
    if (st.done) return;
    for (int c = tid; c < d.C; c += nt) { const int s = b.slot[c]; if (s >= 0 && s <= IMU_MAXCI) q_of_slot[s] = b.imu.inert_slot[c]; }
    const double* Xc = b.Xc + (size_t)st.cur * d.C * 6;
    const double* Xv = b.imu.Xv + (size_t)st.cur * d.C * 9;
    // ---- (1) the factor pairs: local parameter k of the pair (lane k of its 32-lane group) is
    //   0-5 pose_i | 6-8 velocity_i | 9-14 bias_i | 15-20 pose_j | 21-23 velocity_j
    {
        const int grp = tid >> 5, lk = tid & 31;
        for (int fi = grp; fi < b.imu.n_fac; fi += nt >> 5) {
            const ImuFactorDev& F = b.imu.fac[fi];
            const int ci = F.f.cam_i, cj = F.f.cam_j;
            const int si = b.slot[ci], sj = b.slot[cj], qi = b.imu.inert_slot[ci], qj = b.imu.inert_slot[cj];   // qj == qi + 1 (host)
            double r[9], jl[9], rw[6], is[2];
            imu_preintegration_lanes(F, b.imu.gravity, Xc + 6 * ci, Xv + 9 * ci, Xv + 9 * ci + 3, Xc + 6 * cj, Xv + 9 * cj, r, jl);
            imu_bias_walk(F.f, Xv + 9 * ci + 3, Xv + 9 * cj + 3, rw, is);
            // this lane's parameter: a pose column (pc >= 0) or a z row (zr >= 0)
            const bool isP = lk < 6 || (lk >= 15 && lk < 21), live = lk < IMU_NP;
            const bool mine_i = lk < 15;                                  // parameter belongs to camera i
            const int pc = lk < 6 ? 6 * si + lk : 6 * sj + (lk - 15);
            const int zc = lk < 15 ? lk - 6 : lk - 21;                    // component inside the 9-block
            const int zr = lk < 15 ? 9 * qi + zc : 9 * qj + zc;
            double gk = 0.0;
#pragma unroll
            for (int a = 0; a < 9; a++) gk += jl[a] * r[a];
            if (live) {
                if (isP) { if (st.fresh) atomicAdd(&b.gc[pc], gk); }
                else atomicAdd(&gz[zr], gk);
            }
#pragma unroll
            for (int l = 0; l < IMU_NP; l++) {
                double h = 0.0;
#pragma unroll
                for (int a = 0; a < 9; a++) h += jl[a] * __shfl(jl[a], l, 32);
                const bool oP = l < 6 || (l >= 15 && l < 21), o_i = l < 15;           // compile-time per l
                const int opc = l < 6 ? 6 * si + l : 6 * sj + (l - 15);
                const int ozc = l < 15 ? l - 6 : l - 21;
                if (!live) continue;
                if (isP && oP) {
                    if (mine_i == o_i) {                                  // same camera: its 6 x 6 diagonal block lives in U
                        if (st.fresh) atomicAdd(&b.U[(pc / 6) * 36 + (pc % 6) * 6 + (opc % 6)], h);
                    } else if (pc < opc) {
                        atomicAdd(&b.S[(size_t)pc * n + opc], h);         // K7 reads S's upper triangle
                    }
                } else if (!isP && oP) {
                    atomicAdd(&G[(size_t)zr * (n + 1) + opc], h);
                    atomicAdd(&G0[(size_t)zr * n + opc], h);
                } else if (!isP && !oP) {
                    if (mine_i == o_i) atomicAdd(&Hd[(mine_i ? qi : qj) * 81 + zc * 9 + ozc], h);
                    else if (!mine_i) atomicAdd(&Ho[qi * 81 + zc * 9 + ozc], h);      // rows z_j, columns z_i
                }
            }
            if (lk < 6) {                                   // bias walk: -1/sigma on bias_i[a], +1/sigma on bias_j[a]
                const int a = lk;
                const double sg = a < 3 ? is[0] : is[1], s2 = sg * sg;
                double rwa = rw[0];                          // rw[a] without a run-time index (scratch)
#pragma unroll
                for (int q = 1; q < 6; q++) rwa = (a == q) ? rw[q] : rwa;
                atomicAdd(&gz[9 * qi + 3 + a], -sg * rwa);
                atomicAdd(&gz[9 * qj + 3 + a], sg * rwa);
                atomicAdd(&Hd[qi * 81 + (3 + a) * 10], s2);
                atomicAdd(&Hd[qj * 81 + (3 + a) * 10], s2);
                atomicAdd(&Ho[qi * 81 + (3 + a) * 10], -s2);
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
    // ---- (2) Jacobi scale (first linearisation) and damping of the z parameters; their gradient and maximum
    double gm = 0.0;
    for (int t = tid; t < NZ; t += nt) {
        const double h = Hd[(t / 9) * 81 + (t % 9) * 10];
        double sc = b.imu.sc[n + t];
        if (!st.have_scale) { sc = opt.jacobi ? 1.0 / (1.0 + sqrt(h)) : 1.0; b.imu.sc[n + t] = sc; }
        const double s2 = sc * sc, lam = clampd(s2 * h, opt.dmin, opt.dmax) / (st.radius * s2);
        Hd[(t / 9) * 81 + (t % 9) * 10] = h + lam;
        b.imu.lam[n + t] = lam;
        b.imu.gtot[n + t] = gz[t];
        G[(size_t)t * (n + 1) + n] = gz[t];
        gm = fmax(gm, fabs(gz[t]));
    }
    gm = wave_max_nonneg(gm);
    if (lane == 0) s_red[wave] = gm;
    __syncthreads();
    if (tid == 0 && st.fresh) {                             // totals for K7: cost at x and gradient maximum ride in slot 0
        double g = 0.0;
        for (int w = 0; w < nt / 64; w++) g = fmax(g, s_red[w]);
        b.scal[0] += s_cost;
        if (g > 0.0) atomic_max_nonneg(&b.gmax[0], g);
    }
    // ---- (3) A_zz = L D L^T, block tridiagonal: one wave, lane = row of the 18 x 18 window [A_qq . ; A_q+1,q A_q+1,q+1],
    // rows in registers, nine unrolled column steps with v_readlane broadcasts; the window's lower-right block comes out as
    // the next diagonal block with block q's Schur update applied
    if (wave == 0) {
        bool bad = false;
        for (int q = 0; q < Ci; q++) {
            const bool has_next = q + 1 < Ci;
            const int rr = lane < 9 ? lane : (lane < 18 ? lane - 9 : 0);
            double a[18];
#pragma unroll
            for (int k = 0; k < 9; k++) {
                a[k] = lane < 9 ? Hd[q * 81 + rr * 9 + k] : ((lane < 18 && has_next) ? Ho[q * 81 + rr * 9 + k] : 0.0);
                a[9 + k] = (lane >= 9 && lane < 18 && has_next) ? Hd[(q + 1) * 81 + rr * 9 + k] : 0.0;
            }
#pragma unroll
            for (int cc = 0; cc < 9; cc++) {
                int lane_o = lane;
                asm volatile("" : "+v"(lane_o));            // per-step lane masks (hoisted, they spill SGPR pairs)
                const double piv = imu_rl64(a[cc], cc);
                bad = bad || !(piv > 0.0) || !isfinite(piv);
                const double rd = imu_rcp(piv);
                const double lc = a[cc] * rd;
#pragma unroll
                for (int k = cc + 1; k < 18; k++) a[k] -= lc * imu_rl64(a[cc], k);
                a[cc] = lane_o > cc ? lc : a[cc];
                if (lane == cc) dinv[q * 9 + cc] = rd;
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int k = 0; k < 9; k++) {
                if (lane < 9) Hd[q * 81 + rr * 9 + k] = a[k];
                else if (lane < 18 && has_next) { Ho[q * 81 + rr * 9 + k] = a[k]; Hd[(q + 1) * 81 + rr * 9 + k] = a[9 + k]; }
            }
        }
        if (__any(bad) && lane == 0) s_fail = 1;
    }
    __syncthreads();
    if (s_fail) {
        if (tid == 0) b.scal[1] += 1.0;                     // K7's failure count of set 0: the step is invalid
        return;
    }
    // ---- (4) Y = A_zz^-1 G, one thread per column (n pose columns + the gradient): forward L t = g block by block,
    // scale by D^-1, backward L^T y = t; the next block's nine entries are fetched while the current block is solved
    for (int col = tid; col <= n; col += nt) {
        double t[9], nx[9];
#pragma unroll
        for (int k = 0; k < 9; k++) nx[k] = G[(size_t)k * (n + 1) + col];
        double prev[9];
#pragma unroll
        for (int k = 0; k < 9; k++) prev[k] = 0.0;
        for (int q = 0; q < Ci; q++) {
#pragma unroll
            for (int k = 0; k < 9; k++) t[k] = nx[k];
            if (q + 1 < Ci) {
#pragma unroll
                for (int k = 0; k < 9; k++) nx[k] = G[(size_t)(9 * (q + 1) + k) * (n + 1) + col];
            }
            if (q > 0) {
#pragma unroll
                for (int r = 0; r < 9; r++)
#pragma unroll
                    for (int k = 0; k < 9; k++) t[r] -= Ho[(q - 1) * 81 + r * 9 + k] * prev[k];     // L_{q,q-1} D_{q-1} t'_{q-1}: prev holds D t'
            }
#pragma unroll
            for (int r = 1; r < 9; r++)
#pragma unroll
                for (int k = 0; k < r; k++) t[r] -= Hd[q * 81 + r * 9 + k] * t[k];
            // t = L^-1 (...) ; the factor is L D L^T with UNIT L and multipliers l = a / d, so  A x = g  <=>  L w = g, x' = D^-1 w, L^T x = x'
#pragma unroll
            for (int k = 0; k < 9; k++) { prev[k] = t[k]; G[(size_t)(9 * q + k) * (n + 1) + col] = t[k] * dinv[q * 9 + k]; }
        }
        // backward
        double nxt[9];
#pragma unroll
        for (int k = 0; k < 9; k++) nxt[k] = 0.0;
        for (int q = Ci - 1; q >= 0; q--) {
            double y[9];
#pragma unroll
            for (int k = 0; k < 9; k++) y[k] = G[(size_t)(9 * q + k) * (n + 1) + col];
            if (q + 1 < Ci) {
#pragma unroll
                for (int k = 0; k < 9; k++)
#pragma unroll
                    for (int r = 0; r < 9; r++) y[k] -= Ho[q * 81 + r * 9 + k] * nxt[r];              // L_{q+1,q}^T y_{q+1}
            }
#pragma unroll
            for (int k = 7; k >= 0; k--)
#pragma unroll
                for (int r = k + 1; r < 9; r++) y[k] -= Hd[q * 81 + r * 9 + k] * y[r];
#pragma unroll
            for (int k = 0; k < 9; k++) { nxt[k] = y[k]; G[(size_t)(9 * q + k) * (n + 1) + col] = y[k]; }
        }
    }
    __syncthreads();
    // ---- (5) Schur terms onto the pose system: S[k][i] -= sum_z H_zp[z][k] Y[z][i] (k <= i), rhs[k] -= sum_z H_zp[z][k] y_g[z].
    // H_zp's column k is non-zero only in the z blocks of camera(k)'s factors: its own and the two neighbours.
    for (int idx = tid; idx < n * (n + 1); idx += nt) {
        const int k = idx / (n + 1), i = idx % (n + 1);
        if (i < n && i < k) continue;
        const int q0 = q_of_slot[k / 6];                    // inertial slot of the camera that owns pose column k
        if (q0 < 0) continue;
        // 27 rows, all 54 loads in flight together (rows outside [0, Ci) are clamped and weighted 0)
        double u[27], y[27];
#pragma unroll
        for (int e = 0; e < 27; e++) {
            const int z = 9 * (q0 - 1) + e, zc = min(max(z, 0), NZ - 1);
            u[e] = G0[(size_t)zc * n + k];
            y[e] = G[(size_t)zc * (n + 1) + i];
        }
        double acc = 0.0;
#pragma unroll
        for (int e = 0; e < 27; e++) { const int z = 9 * (q0 - 1) + e; acc += (z >= 0 && z < NZ) ? u[e] * y[e] : 0.0; }
        if (i < n) b.S[(size_t)k * n + i] -= acc;
        else b.rhs[k] -= acc;
    }
}

// K7i: the z step from the pose step, candidates, step scalars, candidate cost of the inertial blocks
__global__ __launch_bounds__(KI_THREADS) void ba_imu_expand(BaDims d, BaBufs b, BaOpt opt)
{
    __shared__ BaState st;
    __shared__ double red[KI_THREADS / 64][3];
    __shared__ double s_cand;
    __shared__ int s_bad;
    __shared__ int cam_of_q[IMU_MAXCI + 1];
    const int n = d.n, Ci = b.imu.Ci, NZ = 9 * Ci, tid = threadIdx.x, nt = KI_THREADS;
    if (tid == 0) { st = *b.st; s_cand = 0.0; s_bad = 0; }
    for (int c = tid; c < d.C; c += nt) { const int q = b.imu.inert_slot[c]; if (q >= 0 && q <= IMU_MAXCI) cam_of_q[q] = c; }
    __syncthreads();
    if (st.done || st.solver_failed) return;
    const double* G = imu_G(b);
    const int cand = (st.cur + 1) % (b.ns + 1);
    const double* Xv = b.imu.Xv + (size_t)st.cur * d.C * 9;
    double* Xvn = b.imu.Xv + (size_t)cand * d.C * 9;
    const double* Xn = b.Xc + (size_t)cand * d.C * 6;
    for (int i = tid; i < d.C * 9; i += nt) Xvn[i] = Xv[i];              // cameras without an inertial block keep their state
    __syncthreads();
    double mcc = 0.0, ssq = 0.0, xsq = 0.0;
    bool bad = false;
    for (int t = tid; t < NZ; t += nt) {
        // x_z = y_g - Y_p x_p with x_p = -delta_p (K7 left delta_p in dc);  delta_z = -x_z
        double xz = G[(size_t)t * (n + 1) + n];
#pragma unroll 18
        for (int k = 0; k < n; k++) xz += G[(size_t)t * (n + 1) + k] * b.dc[k];       // n is a multiple of 6
        const double dlt = -xz;
        if (!isfinite(dlt)) bad = true;
        const int cam = cam_of_q[t / 9];
        const double x = Xv[9 * cam + t % 9], xn = x + dlt;
        mcc += 0.5 * (dlt * dlt * b.imu.lam[n + t] - dlt * b.imu.gtot[n + t]);
        ssq += (x - xn) * (x - xn);
        xsq += x * x;
        Xvn[9 * cam + t % 9] = xn;
    }
    if (bad) s_bad = 1;
    mcc = wave_sum(mcc); ssq = wave_sum(ssq); xsq = wave_sum(xsq);
    if ((tid & 63) == 0) { red[tid >> 6][0] = mcc; red[tid >> 6][1] = ssq; red[tid >> 6][2] = xsq; }
    __syncthreads();                                                      // candidates written: the factors read them
    for (int fi = tid; fi < b.imu.n_fac; fi += nt) {
        const ImuFactorDev& F = b.imu.fac[fi];
        const int i = F.f.cam_i, j = F.f.cam_j;
        double r[9], rw[6], is[2], c = 0.0;
        imu_preintegration(F, b.imu.gravity, Xn + 6 * i, Xvn + 9 * i, Xvn + 9 * i + 3, Xn + 6 * j, Xvn + 9 * j, r, nullptr);
        imu_bias_walk(F.f, Xvn + 9 * i + 3, Xvn + 9 * j + 3, rw, is);
        for (int a = 0; a < 9; a++) c += 0.5 * r[a] * r[a];
        for (int a = 0; a < 6; a++) c += 0.5 * rw[a] * rw[a];
        atomicAdd(&s_cand, c);
    }
    __syncthreads();
    if (tid == 0) {
        double a0 = 0, a1 = 0, a2 = 0;
        for (int w = 0; w < nt / 64; w++) { a0 += red[w][0]; a1 += red[w][1]; a2 += red[w][2]; }
        st.cam_scal[0] += a0; st.cam_scal[1] += a1; st.cam_scal[2] += a2; st.cam_scal[3] = s_cand;
        if (s_bad) st.solver_failed = 1;
        *b.st = st;
    }
}

void ba_launch_imu_eliminate(hipStream_t s, const BaDims& d, const BaBufs& b, const BaOpt& opt)
{
    hipLaunchKernelGGL(ba_imu_eliminate, dim3(1), dim3(KI_THREADS), 0, s, d, b, opt);
}
void ba_launch_imu_expand(hipStream_t s, const BaDims& d, const BaBufs& b, const BaOpt& opt)
{
    hipLaunchKernelGGL(ba_imu_expand, dim3(1), dim3(KI_THREADS), 0, s, d, b, opt);
}
int ba_imu_lds_path_max_ci() { return IMU_MAXCI; }
