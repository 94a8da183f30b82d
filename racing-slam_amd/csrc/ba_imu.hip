// ba_imu.hip — inertial bundle adjustment on the LOCAL-WINDOW kernels (K5 MFMA Schur, LDS K7, K8): the velocity / bias
// blocks of the IMU factors (reference src/Optimization.cpp:317-346, src/ImuFactor.cpp:19-118) are eliminated between
// K5 and K7, so that K7 still solves the 6 Cf x 6 Cf pose system it was built for.
//
//   unknowns      poses p (6 per optimised camera), z_c = (velocity 3 | bias 6) per inertial camera, landmarks
//   K5            eliminates the landmarks as always:  S_pp (vision), rhs, U, gc
//   K6a (here)    one workgroup per IMU factor pair, lane = local parameter (one dual partial per lane, imu_dual.h):
//                   H_pp -> U (same camera) / S (pose_i x pose_j),  g_p -> gc,  cost -> K5's slot lines
//                   H_zz (block tridiagonal: a factor joins consecutive inertial cameras), H_zp, g_z -> accumulators
//   K6b (here)    one workgroup: damps z like every Ceres parameter (Jacobi scale from the first linearisation, clamp,
//                 / radius), factors A_zz = L D L^T block by block, forward-substitutes  W = L^-1 [H_zp | g_z]  and adds
//                 the Schur terms  S -= W_p^T D^-1 W_p,  rhs -= W_p^T D^-1 w_g  as an LDS-tiled product
//   K7            unchanged: (U + Lambda_p + S) x_p = gc + rhs, candidate cameras, pose part of the step scalars
//   K7i (here)    x_z = L^-T D^-1 (w_g - W_p x_p): candidate velocities / biases, their part of the model cost change /
//                 step norm / x norm, and the inertial blocks' cost at the candidate  -> BaState::cam_scal
//   K8            unchanged (and clears the inertial accumulators with the others).
// Same LM schedule as the N x N blocked solve (ba_solve_big.hip) it replaces for windows of at most 21 optimised cameras.
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

// Layout behind BaImu::zacc.  SHARED by the speculative sets, zeroed before every round (K8): H [9 Ci][n + 1] (row z:
// H_zp | g_z), the diagonal blocks of H_zz [Ci][81], the sub-diagonal blocks (q + 1, q) [Ci][81], g_z [9 Ci].
// Then PER SET (its radius damps z differently): W = L^-1 H [9 Ci][n + 1], the factor L [Ci][81] | [Ci][81], D^-1 [9 Ci]
// and the damping [9 Ci].
struct ImuView { double *H, *Hd, *Ho, *gz, *W, *Ld, *Lo, *dinv, *lam; };
__host__ __device__ inline size_t imu_zacc_count(int Ci, int n) { return (size_t)9 * Ci * (n + 1) + 162 * (size_t)Ci + 9 * (size_t)Ci; }
__host__ __device__ inline size_t imu_set_count(int Ci, int n) { return (size_t)9 * Ci * (n + 1) + 162 * (size_t)Ci + 18 * (size_t)Ci; }
__device__ __forceinline__ ImuView imu_view(const BaBufs& b, int n, int set)
{
    const int Ci = b.imu.Ci;
    ImuView v;
    v.H = b.imu.zacc; v.Hd = v.H + (size_t)9 * Ci * (n + 1); v.Ho = v.Hd + 81 * Ci; v.gz = v.Ho + 81 * Ci;
    v.W = b.imu.zacc + imu_zacc_count(Ci, n) + (size_t)set * imu_set_count(Ci, n);
    v.Ld = v.W + (size_t)9 * Ci * (n + 1); v.Lo = v.Ld + 81 * Ci; v.dinv = v.Lo + 81 * Ci; v.lam = v.dinv + 9 * Ci;
    return v;
}

// ---------------------------------------------------------------------------------------------- K6a
// local parameter k of the pair (lane k): 0-5 pose_i | 6-8 velocity_i | 9-14 bias_i | 15-20 pose_j | 21-23 velocity_j
__global__ __launch_bounds__(64) void ba_imu_factors(BaDims d, BaBufs b)
{
    const BaState st = *b.st;
    if (st.done || threadIdx.x >= 32) return;
    const int n = d.n, lk = threadIdx.x, fi = blockIdx.x;
    const ImuView V = imu_view(b, n, 0);
    const double* Xc = b.Xc + (size_t)st.cur * d.C * 6;
    const double* Xv = b.imu.Xv + (size_t)st.cur * d.C * 9;
    const ImuFactorDev& F = b.imu.fac[fi];
    const int ci = F.f.cam_i, cj = F.f.cam_j;
    const int si = b.slot[ci], sj = b.slot[cj], qi = b.imu.inert_slot[ci], qj = b.imu.inert_slot[cj];   // qj == qi + 1 (host)
    double r[9], jl[9], rw[6], is[2];
    imu_preintegration_lanes(F, b.imu.gravity, Xc + 6 * ci, Xv + 9 * ci, Xv + 9 * ci + 3, Xc + 6 * cj, Xv + 9 * cj, r, jl);
    imu_bias_walk(F.f, Xv + 9 * ci + 3, Xv + 9 * cj + 3, rw, is);
    const bool isP = lk < 6 || (lk >= 15 && lk < 21), live = lk < IMU_NP;
    const bool mine_i = lk < 15;                                  // parameter belongs to camera i
    const int pc = lk < 6 ? 6 * si + lk : 6 * sj + (lk - 15);
    const int zc = lk < 15 ? lk - 6 : lk - 21;                    // component inside the 9-block
    const int zr = lk < 15 ? 9 * qi + zc : 9 * qj + zc;
    // the other lanes' columns through LDS (a __shfl of a double is two ds_bpermute; these are broadcast reads; the 32
    // lanes are one wave: no barrier needed beyond the LDS write -> read dependency the compiler tracks)
    __shared__ double Jc[9][32];
#pragma unroll
    for (int a = 0; a < 9; a++) Jc[a][lk] = jl[a];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    double gk = 0.0;
#pragma unroll
    for (int a = 0; a < 9; a++) gk += jl[a] * r[a];
    if (live) {
        if (isP) { if (st.fresh) atomicAdd(&b.gc[pc], gk); }
        else atomicAdd(&V.gz[zr], gk);
    }
#pragma unroll
    for (int l = 0; l < IMU_NP; l++) {
        double h = 0.0;
#pragma unroll
        for (int a = 0; a < 9; a++) h += jl[a] * Jc[a][l];
        const bool oP = l < 6 || (l >= 15 && l < 21), o_i = l < 15;           // compile-time per l
        const int opc = l < 6 ? 6 * si + l : 6 * sj + (l - 15);
        const int ozc = l < 15 ? l - 6 : l - 21;
        if (!live) continue;
        if (isP && oP) {
            if (mine_i == o_i) {                                  // same camera: its 6 x 6 diagonal block lives in U
                if (st.fresh) atomicAdd(&b.U[(pc / 6) * 36 + (pc % 6) * 6 + (opc % 6)], h);
            } else if (pc < opc) {
                for (int q = 0; q < st.nact; q++) atomicAdd(&b.S[(size_t)q * n * n + (size_t)pc * n + opc], h);   // every set's S; K7 reads the upper triangle
            }
        } else if (!isP && oP) {
            atomicAdd(&V.H[(size_t)zr * (n + 1) + opc], h);
        } else if (!isP && !oP) {
            if (mine_i == o_i) atomicAdd(&V.Hd[(mine_i ? qi : qj) * 81 + zc * 9 + ozc], h);
            else if (!mine_i) atomicAdd(&V.Ho[qi * 81 + zc * 9 + ozc], h);      // rows z_j, columns z_i
        }
    }
    if (lk < 6) {                                   // bias walk: -1/sigma on bias_i[a], +1/sigma on bias_j[a]
        const int a = lk;
        const double sg = a < 3 ? is[0] : is[1], s2 = sg * sg;
        double rwa = rw[0];                          // rw[a] without a run-time index (scratch)
#pragma unroll
        for (int q = 1; q < 6; q++) rwa = (a == q) ? rw[q] : rwa;
        atomicAdd(&V.gz[9 * qi + 3 + a], -sg * rwa);
        atomicAdd(&V.gz[9 * qj + 3 + a], sg * rwa);
        atomicAdd(&V.Hd[qi * 81 + (3 + a) * 10], s2);
        atomicAdd(&V.Hd[qj * 81 + (3 + a) * 10], s2);
        atomicAdd(&V.Ho[qi * 81 + (3 + a) * 10], -s2);
    }
    if (lk == 0 && st.fresh) {                      // cost at x: into one of K5's slot lines (K7 sums them)
        double c = 0.0;
        for (int a = 0; a < 9; a++) c += 0.5 * r[a] * r[a];
        for (int a = 0; a < 6; a++) c += 0.5 * rw[a] * rw[a];
        atomicAdd(&b.scal[(size_t)(fi & (BA_NSLOT - 1)) * BA_SLOT_STRIDE], c);
    }
}

// ---------------------------------------------------------------------------------------------- K6b
#define KCH 84                  // z rows per staged chunk (a multiple of 4: MFMA k-steps): two passes for 18 cameras
#define KLD 116                 // row stride of a staged chunk (>= 16 * tiles per side; n + 1 <= 112)
#define KTW 4                   // output tiles per wave (28 upper-triangular tiles of 16 x 16 over 8 waves)
__global__ __launch_bounds__(KI_THREADS) void ba_imu_eliminate(BaDims d, BaBufs b, BaOpt opt)
{
    extern __shared__ __attribute__((aligned(16))) double dyn[];      // [KCH][KLD]: chunk of W
    __shared__ double Hd[IMU_MAXCI * 81];       // diagonal blocks of A_zz, then their L (unit lower) / D
    __shared__ double Ho[IMU_MAXCI * 81];       // block (q + 1, q), then L_{q+1,q}
    __shared__ double gz[IMU_MAXCI * 9], dinv[IMU_MAXCI * 9];
    __shared__ BaState st;
    __shared__ double s_red[KI_THREADS / 64];
    __shared__ int s_fail;
    const int n = d.n, Ci = b.imu.Ci, NZ = 9 * Ci, tid = threadIdx.x, nt = KI_THREADS;
    const int lane = tid & 63, wave = tid >> 6;
    const int set = blockIdx.x;                 // speculative radius evaluated by this workgroup (ba_common.h)
    const ImuView V = imu_view(b, n, set);
    if (tid == 0) { st = *b.st; s_fail = 0; }
    for (int i = tid; i < Ci * 81; i += nt) { Hd[i] = V.Hd[i]; Ho[i] = V.Ho[i]; }
    for (int i = tid; i < NZ; i += nt) gz[i] = V.gz[i];
    __syncthreads();
    if (st.done || set >= st.nact) return;
    const double radius = ba_set_radius(st, set);
    // ---- (2) Jacobi scale (first linearisation) and damping of the z parameters; their gradient and maximum
    double gm = 0.0;
    for (int t = tid; t < NZ; t += nt) {
        const double h = Hd[(t / 9) * 81 + (t % 9) * 10];
        double sc = b.imu.sc[n + t];
        if (!st.have_scale) { sc = opt.jacobi ? 1.0 / (1.0 + sqrt(h)) : 1.0; b.imu.sc[n + t] = sc; }      // first round: one set
        const double s2 = sc * sc, lam = clampd(s2 * h, opt.dmin, opt.dmax) / (radius * s2);
        Hd[(t / 9) * 81 + (t % 9) * 10] = h + lam;
        V.lam[t] = lam;
        if (set == 0) b.imu.gtot[n + t] = gz[t];
        gm = fmax(gm, fabs(gz[t]));
    }
    gm = wave_max_nonneg(gm);
    if (lane == 0) s_red[wave] = gm;
    __syncthreads();
    if (tid == 0 && st.fresh && set == 0) {                 // gradient maximum of the z blocks rides in K5's slot 0
        double g = 0.0;
        for (int w = 0; w < nt / 64; w++) g = fmax(g, s_red[w]);
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
        if (tid == 0) b.scal[1 + set] += 1.0;               // K7's failure count of this set: the step is invalid
        return;
    }
    for (int i = tid; i < Ci * 81; i += nt) { V.Ld[i] = Hd[i]; V.Lo[i] = Ho[i]; }          // for K7i's backward substitution
    for (int i = tid; i < NZ; i += nt) V.dinv[i] = dinv[i];
    // ---- (4) W <- L^-1 W, one thread per column (n pose columns + the gradient), block by block; the next block's nine
    // entries are fetched while the current block is solved
    for (int col = tid; col <= n; col += nt) {
        double t[9], nx[9], prev[9];
#pragma unroll
        for (int k = 0; k < 9; k++) { nx[k] = col < n ? V.H[(size_t)k * (n + 1) + col] : gz[k]; prev[k] = 0.0; }
        for (int q = 0; q < Ci; q++) {
#pragma unroll
            for (int k = 0; k < 9; k++) t[k] = nx[k];
            if (q + 1 < Ci) {
#pragma unroll
                for (int k = 0; k < 9; k++) nx[k] = col < n ? V.H[(size_t)(9 * (q + 1) + k) * (n + 1) + col] : gz[9 * (q + 1) + k];
            }
            if (q > 0) {
#pragma unroll
                for (int r = 0; r < 9; r++)
#pragma unroll
                    for (int k = 0; k < 9; k++) t[r] -= Ho[(q - 1) * 81 + r * 9 + k] * prev[k];     // L_{q,q-1} w_{q-1}
            }
#pragma unroll
            for (int r = 1; r < 9; r++)
#pragma unroll
                for (int k = 0; k < r; k++) t[r] -= Hd[q * 81 + r * 9 + k] * t[k];
#pragma unroll
            for (int k = 0; k < 9; k++) { prev[k] = t[k]; V.W[(size_t)(9 * q + k) * (n + 1) + col] = t[k]; }
        }
    }
    __syncthreads();
    // ---- (5) Schur terms onto the pose system: S[k][i] -= sum_z W[z][k] W[z][i] / d_z (k <= i < n), rhs[k] -= ... (i = n),
    // as a product on the f64 matrix cores: 16 x 16 output tiles (the upper-triangular ones, dealt round-robin to the eight
    // waves), K = the z rows, staged KCH at a time as W and D^-1 W (k-major, so an MFMA operand is one ds_read_b64)
    {
        typedef __attribute__((ext_vector_type(4))) double d4;
        double* Wc = dyn;
        const int lr = lane & 15, lq = lane >> 4;
        const int nt16 = (n + 1 + 15) / 16;                 // tiles per side (n = 108: 7)
        // this wave's tiles: t = wave, wave + 8, ... over the (tr <= tc) list
        int tr[KTW], tc[KTW];
        bool tv[KTW];
        d4 acc[KTW];
#pragma unroll
        for (int e = 0; e < KTW; e++) {
            int t = wave + 8 * e, r = 0;
            while (r < nt16 && t >= nt16 - r) { t -= nt16 - r; r++; }       // row r holds nt16 - r tiles (tc = r .. nt16 - 1)
            tv[e] = r < nt16;
            tr[e] = tv[e] ? r : 0;
            tc[e] = tv[e] ? r + t : 0;
            acc[e] = d4{0.0, 0.0, 0.0, 0.0};
        }
        for (int z0 = 0; z0 < NZ; z0 += KCH) {
            const int rows = min(KCH, NZ - z0);
            {
                // thread -> (column c = tid & 127, rows rg, rg + 4, ...): coalesced rows, seven loads of a thread in flight
                const int c = tid & 127, rg = tid >> 7;
                for (int e0 = 0; e0 < KCH / 4; e0 += 7) {
                    double v[7];
#pragma unroll
                    for (int e = 0; e < 7; e++) {
                        const int zz = rg + 4 * (e0 + e);
                        v[e] = (zz < rows && c <= n) ? V.W[(size_t)(z0 + zz) * (n + 1) + c] : 0.0;
                    }
#pragma unroll
                    for (int e = 0; e < 7; e++) {
                        const int zz = rg + 4 * (e0 + e);
                        if (c < KLD && zz < KCH) Wc[zz * KLD + c] = v[e];
                    }
                }
            }
            __syncthreads();
            const int ksteps = (rows + 3) / 4;              // rows beyond `rows` are zero
            for (int kc = 0; kc < ksteps; kc++) {
                const double dz = (4 * kc + lq) < rows ? dinv[z0 + 4 * kc + lq] : 0.0;
#pragma unroll
                for (int e = 0; e < KTW; e++) {
                    if (!tv[e]) continue;                   // wave-uniform
                    const double av = Wc[(4 * kc + lq) * KLD + 16 * tr[e] + lr];
                    const double bv = Wc[(4 * kc + lq) * KLD + 16 * tc[e] + lr] * dz;
                    acc[e] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[e], 0, 0, 0);
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int e = 0; e < KTW; e++) {
            if (!tv[e]) continue;
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int k = 16 * tr[e] + lq + 4 * reg, i = 16 * tc[e] + lr;       // C[row = lq + 4 reg][col = lr]
                if (k < n) {
                    if (i < n) { if (k <= i) b.S[(size_t)set * n * n + (size_t)k * n + i] -= acc[e][reg]; }
                    else if (i == n) b.rhs[(size_t)set * n + k] -= acc[e][reg];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- K7i
// the z step from the pose step, candidates, step scalars, candidate cost of the inertial blocks
__global__ __launch_bounds__(KI_THREADS) void ba_imu_expand(BaDims d, BaBufs b, BaOpt opt)
{
    __shared__ BaState st;
    __shared__ double red[KI_THREADS / 64][3];
    __shared__ double s_cand;
    __shared__ int s_bad;
    __shared__ int cam_of_q[IMU_MAXCI + 1];
    __shared__ double u[IMU_MAXCI * 9];                     // D^-1 (w_g - W_p x_p), then x_z
    const int n = d.n, Ci = b.imu.Ci, NZ = 9 * Ci, tid = threadIdx.x, nt = KI_THREADS;
    const int set = blockIdx.x;
    if (tid == 0) { st = *b.st; s_cand = 0.0; s_bad = 0; }
    for (int c = tid; c < d.C; c += nt) { const int q = b.imu.inert_slot[c]; if (q >= 0 && q <= IMU_MAXCI) cam_of_q[q] = c; }
    __syncthreads();
    if (st.done || set >= st.nact) return;
    if (set == 0 ? st.solver_failed : b.set_out[set].solver_failed) return;
    const ImuView V = imu_view(b, n, set);
    const double* dcs = b.dc + (size_t)set * BA_DC_STRIDE(n);
    const int cand = (st.cur + 1 + set) % (b.ns + 1);
    const double* Xv = b.imu.Xv + (size_t)st.cur * d.C * 9;
    double* Xvn = b.imu.Xv + (size_t)cand * d.C * 9;
    const double* Xn = b.Xc + (size_t)cand * d.C * 6;
    for (int i = tid; i < d.C * 9; i += nt) Xvn[i] = Xv[i];              // cameras without an inertial block keep their state
    // x_p = -delta_p (K7 left delta_p in dc): u = D^-1 (w_g + W_p delta_p)
    for (int t = tid; t < NZ; t += nt) {
        double acc = V.W[(size_t)t * (n + 1) + n];
#pragma unroll 18
        for (int k = 0; k < n; k++) acc += V.W[(size_t)t * (n + 1) + k] * dcs[k];        // n is a multiple of 6
        u[t] = acc * V.dinv[t];
    }
    __syncthreads();
    // L^T x = u from the last block: nine lanes, lane k = component k of the block
    if (tid < 64) {
        const int k = tid < 9 ? tid : 0;
        double ynext = 0.0;                                 // this lane's component of x_{q+1}
        for (int q = Ci - 1; q >= 0; q--) {
            double y = u[9 * q + k];
            if (q + 1 < Ci) {
#pragma unroll
                for (int r = 0; r < 9; r++) y -= V.Lo[q * 81 + r * 9 + k] * imu_rl64(ynext, r);       // L_{q+1,q}^T x_{q+1}
            }
#pragma unroll
            for (int r = 8; r >= 1; r--) {                  // unit upper-triangular solve: x_r is final when step r runs
                const double xr = imu_rl64(y, r);
                y -= (k < r) ? V.Ld[q * 81 + r * 9 + k] * xr : 0.0;
            }
            ynext = y;
            if (tid < 9) u[9 * q + k] = y;
        }
    }
    __syncthreads();
    double mcc = 0.0, ssq = 0.0, xsq = 0.0;
    bool bad = false;
    for (int t = tid; t < NZ; t += nt) {
        const double dlt = -u[t];
        if (!isfinite(dlt)) bad = true;
        const int cam = cam_of_q[t / 9];
        const double x = Xv[9 * cam + t % 9], xn = x + dlt;
        mcc += 0.5 * (dlt * dlt * V.lam[t] - dlt * b.imu.gtot[n + t]);
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
        if (set == 0) {
            st.cam_scal[0] += a0; st.cam_scal[1] += a1; st.cam_scal[2] += a2; st.cam_scal[3] = s_cand;
            if (s_bad) st.solver_failed = 1;
            *b.st = st;
        } else {
            BaSetOut so = b.set_out[set];
            so.cam_scal[0] += a0; so.cam_scal[1] += a1; so.cam_scal[2] += a2; so.cam_scal[3] = s_cand;
            if (s_bad) so.solver_failed = 1;
            b.set_out[set] = so;
        }
    }
}

// ---------------------------------------------------------------------------------------------- host glue
size_t ba_imu_lds_zacc_doubles(int Ci, int n) { return imu_zacc_count(Ci, n); }
size_t ba_imu_lds_total_doubles(int Ci, int n, int ns) { return imu_zacc_count(Ci, n) + (size_t)ns * imu_set_count(Ci, n); }

void ba_launch_imu_eliminate(hipStream_t s, const BaDims& d, const BaBufs& b, const BaOpt& opt)
{
    hipLaunchKernelGGL(ba_imu_factors, dim3(b.imu.n_fac), dim3(64), 0, s, d, b);
    const size_t lds = sizeof(double) * KCH * KLD;
    (void)rs_lds_attr((const void*)ba_imu_eliminate, lds);
    hipLaunchKernelGGL(ba_imu_eliminate, dim3(b.ns), dim3(KI_THREADS), lds, s, d, b, opt);
}
void ba_launch_imu_expand(hipStream_t s, const BaDims& d, const BaBufs& b, const BaOpt& opt)
{
    hipLaunchKernelGGL(ba_imu_expand, dim3(b.ns), dim3(KI_THREADS), 0, s, d, b, opt);
}
int ba_imu_lds_path_max_ci() { return IMU_MAXCI; }
