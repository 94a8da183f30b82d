// ba_common.h — structures and device math shared by the bundle-adjustment
// translation units (ba.hip, ba_solve.hip, ba_schur.hip).
#pragma once
#include <math.h>

#include "common.h"

#define BA_PREP 24            // doubles per camera: R[9] Jl[9] c[3] small pad pad
#define BA_DC_STRIDE(n) ((((n) + 2 + 15) / 16) * 16)     // doubles per speculative set in b.dc: whole 128-byte lines, so that no
                              // line holds the camera steps of two sets (each is published by its own K7 workgroup, ba_solve.hip)
#define BA_PREP_LDS 25        // row stride of the camera blocks when they are staged in LDS: 24 doubles = 48 banks puts every
                              // camera on one of TWO bank offsets (48 c mod 32), and the lanes of a wave read the same entry of
                              // DIFFERENT cameras; 50 banks give 16 cameras 16 disjoint bank pairs
#define BA_THREADS 64
#define BA_NSLOT 64            // partial-sum slots (one 64-B line each) for the scalar reductions:
                              // device-scope atomics on ONE line serialise at ~12 ns each
#define BA_SLOT_STRIDE 8      // doubles per slot line
#define BA_UREP 8             // replicas of the camera-side accumulators (U, gc, rhs): workgroup w adds
                              // into replica w % 8, K7 folds them; cuts same-line atomic traffic 8x
#define BA_DBG_WORDS 96
#define BA_MAX_LDS_N 126      // largest reduced system kept in LDS by K7
#define BA_DEFAULT_SREP 1      // replicas of S on the local-window path (BaBufs::srep)
#define BA_DEFAULT_SETS (BA_MAXSETS < 5 ? BA_MAXSETS : 5)      // default number of speculative radii a round MAY evaluate on the local-window path (ba_round_sets below)
#define BA_CALIBRATED_SETS 3   // ... once the trust-region radius is calibrated
#ifndef BA_MAXSETS
#define BA_MAXSETS 5          // speculative trust-region radii evaluated per round (see "Speculative radii" below)
#endif

struct BaState {
    double radius, decrease_factor, x_cost, initial_cost;
    double cam_scal[4];       // K7 (set 0): mcc_c, step_sq_c, x_sq_c, candidate cost of the inertial blocks
    int iter, successful, invalid_steps, done;
    int termination, cur, have_scale, solver_failed;
    int fresh, usable;
    int consec_accepts;       // successful steps in a row (drives how many radii the next round speculates on)
    int nact;                 // sets evaluated by THIS round (1 .. ns)
    int n_rounds, n_fresh, n_sets;        // accounting: rounds that did work, of those relinearised, sets evaluated
    int hand_lost;                        // set by ba_finalize: a consumer of the fused K7 + K8 launch gave up waiting (the host re-runs the solve)
    int calibrated;                       // 0: no step rejected yet, 1: inside an unresolved rejection streak, 2: an accepted step has
                                          // followed a rejected one (the radius has found the problem's scale) — ba_decide
    int pad_[3];
};

// Speculative radii.  After a REJECTED step Ceres does not relinearise: x stays, the radius becomes
// radius / decrease_factor and the factor doubles (oracle/ba.c, TrustRegionMinimizer).  The radii of the next
// rejections are therefore known in advance, and one round (K5 + K7 + K8) evaluates the LM step for `ns` of them
// at once: set 0 with the state's radius r, set 1 with r / f, set 2 with r / (f * 2f).  The decision at the start
// of the next round walks the sets in order — exactly the iterations the sequential loop would have run — and
// stops at the first accepted step or termination.  Sets >= 1 cost K5 a second / third damped-inverse + SYRK
// phase; K7 and K8 run the sets side by side in extra workgroups.  What a set needs of its own:
//   S, the Schur part of the rhs, damped V^-1, point damping, delta_c, candidate state, step scalars.
// The undamped linearisation (U, gc, V, gp, cost at x) is shared.
struct BaSetOut {             // written by K7's workgroup of set s >= 1 (set 0 writes BaState)
    double cam_scal[4];
    int solver_failed, pad[7];
};

// progress word in pinned host memory: the first kernel of every round publishes where the state machine is, so
// that the host only enqueues the rounds that can still be needed (see rs_bundle_adjust)
struct BaProgress {
    volatile int round, done, iter, pad;
};

// one entry per LM iteration (mirrors rs_ba_iteration in rsgpu.h), written by workgroup 0 when it applies a decision
struct BaTrace {
    double cost, candidate_cost, model_cost_change, radius, step_norm, x_norm;
    int outcome, pad;
    double pad2;
};

struct BaDims {
    int C, Cf, P, M, n;       // n = 6*Cf
    float fx, fy, cx, cy;
    double huber_a;
};

// Inertial residual blocks (reference src/Optimization.cpp:317-346, src/ImuFactor.cpp): the reduced camera system
// grows from 6 unknowns per free camera to N = 6 Cf + 9 Ci (velocity 3 + bias 6 for each of the Ci frames an IMU
// factor touches).  The landmark side (K5, K8) is unchanged — the factors only involve camera-side blocks — and the
// reduced solve runs on the blocked solver with its own prologue / finish (ba_solve_big.hip).  All null / zero on a
// vision-only solve.
struct ImuFactorDev;
struct BaImu {
    int n_fac, Ci, N;
    const ImuFactorDev* fac;       // [n_fac] factors + whiteners
    const int32_t* inert_slot;     // [C] index among the inertial frames or -1
    double* Xv;                    // [2][C][9] velocity (3) | bias (6) per camera, state buffers cur / cur ^ 1
    double* A;                     // [N][N] damped reduced matrix (lower triangle)
    double* yv;                    // [N + 1] right-hand side
    double* lam;                   // [N] damping
    double* sc;                    // [N] Jacobi scale
    double* gtot;                  // [N] camera-side gradient incl. the inertial blocks
    double* Jf;                    // [n_fac][9][24] preintegration Jacobians of this round
    double* zacc;                  // local-window path (ba_imu.hip): accumulators the factor kernel adds into, cleared by K8
    int zacc_n;                    //   [W | H_zz diagonal blocks | H_zz sub-diagonal blocks | g_z], zacc_n doubles (0 = path not in use)
    double gravity[3];
};

struct BaBufs {
    BaImu imu;
    const int32_t* obs_ptr;   // [P+1]
    const int32_t* obs_cam;   // [M]
    const float2* obs_uv;     // [M]
    const int32_t* obs_cs;    // [M] cam | (slot + 1) << 16, built by the landmark grouping (null on the generic path)
    int srep;        // replicas of S that K5's workgroups scatter into (workgroup w adds into replica w % srep, K7 folds them):
                     // the f64 atomics of ~250 workgroups into the ~730 lines of one S run at a fraction of the chip's atomic rate
    size_t s_rep_stride;   // doubles per replica of S = ns * n * n
    int ns;          // speculative sets (1 .. BA_MAXSETS); state buffers rotate over ns + 1 slots:
                     // x lives in slot st.cur, the candidate of set s in slot (st.cur + 1 + s) % (ns + 1)
    double* Xc;      // [ns+1][C][6]
    double* Xp;      // [ns+1][P][3]
    double* prep;    // [ns+1][C][BA_PREP]
    int32_t* slot;   // [C]  reduced-system slot of a free camera or -1
    double* sc;      // [n]
    double* sp;      // [P][3]
    double* Vinv;    // [ns][P][6]  (xx xy xz yy yz zz)
    double* gp;      // [P][3]
    double* lamp;    // [ns][P][3]
    // accumulators, contiguous for one all-reduce: S[srep][ns][n*n] | BA_UREP x { rhs[ns][n] U[Cf*36] gc[n] } | scal | gmax
    double* acc;
    size_t acc_count;
    double* S; double* rhs; double* U; double* gc;   // S: set 0, set s at + s*n*n; rhs/U/gc: replica 0 (rhs of set s at
                                                     // + s*n); replica r at + r * cam_stride
    size_t cam_stride;       // doubles per replica = ns*n + Cf*36 + n
    double* scal;    // [BA_NSLOT][8] per slot: cost_x, fail_count of set 0, 1, 2 (summed by K7)
    double* Vc;      // [P][6] undamped point blocks of the last FRESH linearisation (a rejected step only changes the damping)
    double* Ukeep;   // [Cf*36 + n] folded U | gc of the last fresh linearisation
    double* gmax;    // [BA_NSLOT][8] per slot: bits of a non-negative double (max; K7 folds); THIS rank's block of gmax_all
    double* gmax_all;   // [gmax_blocks][BA_NSLOT][8], inside the all-reduced accumulator block: rank r only writes block r,
    int gmax_blocks;    //   so that the SUM all-reduce of the accumulators also delivers every rank's maximum (no max collective)
    int decided;        // the round's state block *st was written by ba_decide_round in front of K5 (launches with more items than
                        // compute units, batched windows): K5's workgroups load it instead of each re-deriving the decision
    double* pt_scal; // [ns][BA_NSLOT][8] per slot, K8 of THIS round: cand_cost, mcc_p, step_sq_p, x_sq_p
    const double* pt_prev;   // the same block of the PREVIOUS round (read by the next linearisation's decision)
    double* dc;      // [ns][BA_DC_STRIDE(n)]
    BaSetOut* set_out;            // [ns] K7 results of sets >= 1, THIS round
    const BaSetOut* set_prev;     // the previous round's
    BaProgress* prog;             // pinned host memory (null when the caller does not poll)
    unsigned long long* dbg;   // [BA_DBG_WORDS]: in-kernel phase counters in [0, 64) (diagnostic; rs_prof_counters), hand-off words behind them
    unsigned long long hand_timeout;   // fused K7 + K8 launch: ticks of the 100 MHz wall clock a consumer waits for a hand-off word
    BaTrace* trace;          // [max_iter] per-iteration record (rs_ba_get_trace)
    BaState* st;             // state of THIS iteration (st[it & 1])
    const BaState* st_prev;  // state the previous iteration ended with (st[(it + 1) & 1])
};

struct BaOpt {
    int max_iter, max_invalid, jacobi;
    double r0, rmax, rmin, min_rel, dmin, dmax, ftol, gtol, ptol;
};

// ----------------------------------------------------------------- device math
__device__ __forceinline__ void cam_prepare(const double* cam, double* out)
{
    const double ax = cam[0], ay = cam[1], az = cam[2];
    const double th2 = ax * ax + ay * ay + az * az;
    double A, B, Cc, small;
    if (th2 > 2.220446049250313e-16) {
        // one sincos of the half angle instead of two sines (this sits on K7's serial epilogue):
        // sin(th) = 2 sin(th/2) cos(th/2), 1 - cos(th) = 2 sin^2(th/2); reciprocals instead of three divides
        const double th = sqrt(th2);
        double sh, ch;
        sincos(0.5 * th, &sh, &ch);
        const double sth = 2.0 * sh * ch;
        const double ith = 1.0 / th, ith2 = ith * ith;
        A = sth * ith;
        B = 2.0 * sh * sh * ith2;
        Cc = (th - sth) * ith2 * ith;
        small = 0.0;
    } else {   // ceres::AngleAxisRotatePoint's first-order branch: R = I + [w]x, d/dw = -[q]x
        A = 1.0; B = 0.0; Cc = 0.0; small = 1.0;
    }
    const double W[9] = {0, -az, ay, az, 0, -ax, -ay, ax, 0};
    const double W2[9] = {-(ay * ay + az * az), ax * ay, ax * az, ax * ay, -(ax * ax + az * az), ay * az,
                          ax * az, ay * az, -(ax * ax + ay * ay)};
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const double id = (i == 0 || i == 4 || i == 8) ? 1.0 : 0.0;
        out[i] = id + A * W[i] + B * W2[i];          // R
        out[9 + i] = id + B * W[i] + Cc * W2[i];     // left Jacobian of SO(3)
    }
    out[18] = cam[3]; out[19] = cam[4]; out[20] = cam[5];
    out[21] = small; out[22] = 0.0; out[23] = 0.0;
}

// 1 / p and 1 / sqrt(x) to within an ulp or two: v_rcp_f64 / v_rsq_f64 + two Newton steps.  The IEEE sequences the
// compiler emits for `/` and sqrt() are ~40 instructions each with a long dependent tail.
__device__ __forceinline__ double rcp_nr(double p)
{
    double r = __builtin_amdgcn_rcp(p);
    r = fma(fma(-p, r, 1.0), r, r);
    r = fma(fma(-p, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double rsqrt_nr(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    y = y * (1.5 - h * y * y);
    y = y * (1.5 - h * y * y);
    return y;
}

struct ObsLin {
    double r0, r1, w, rho;
    double jc[12];   // 2x6  [d/d aa | d/d centre]
    double jp[6];    // 2x3
};

template <bool JAC>
__device__ __forceinline__ void obs_eval(const double* __restrict__ cp, const double X[3], float2 uv,
                                         const BaDims& d, ObsLin& o)
{
    const double q0 = X[0] - cp[18], q1 = X[1] - cp[19], q2 = X[2] - cp[20];
    const double p0 = cp[0] * q0 + cp[1] * q1 + cp[2] * q2;
    const double p1 = cp[3] * q0 + cp[4] * q1 + cp[5] * q2;
    const double p2 = cp[6] * q0 + cp[7] * q1 + cp[8] * q2;
    const double fx = (double)d.fx, fy = (double)d.fy;
    // one reciprocal of the depth for the residual and the Jacobians, one reciprocal square root for the loss (instead of
    // three IEEE divisions and a square root per observation; the results differ from those by an ulp or two, nine orders
    // below the tolerances this solve is held to)
    const double iz = rcp_nr(p2);
    o.r0 = fx * p0 * iz + (double)d.cx - (double)uv.x;     // src/Optimization.cpp:48-49
    o.r1 = fy * p1 * iz + (double)d.cy - (double)uv.y;
    const double s = o.r0 * o.r0 + o.r1 * o.r1;
    const double b2 = d.huber_a * d.huber_a;
    if (s > b2) {   // ceres::HuberLoss: rho = 2 a sqrt(s) - a^2, weight a / sqrt(s)
        const double irs = rsqrt_nr(s);
        o.rho = 2.0 * d.huber_a * (s * irs) - b2;
        o.w = d.huber_a * irs;
    } else {
        o.rho = s;
        o.w = 1.0;
    }
    if (JAC) {
        const double a = fx * iz, b = fy * iz;
        const double ax = -a * p0 * iz, bx = -b * p1 * iz;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            o.jp[k] = a * cp[k] + ax * cp[6 + k];
            o.jp[3 + k] = b * cp[3 + k] + bx * cp[6 + k];
            o.jc[3 + k] = -o.jp[k];
            o.jc[9 + k] = -o.jp[3 + k];
        }
        const bool small = cp[21] != 0.0;
        const double v0 = small ? q0 : p0, v1 = small ? q1 : p1, v2 = small ? q2 : p2;
#pragma unroll
        for (int k = 0; k < 3; k++) {   // d p / d aa_k = Jl[:,k] x v
            const double m0 = cp[9 + k], m1 = cp[12 + k], m2 = cp[15 + k];
            const double c0 = m1 * v2 - m2 * v1, c1 = m2 * v0 - m0 * v2, c2 = m0 * v1 - m1 * v0;
            o.jc[k] = a * c0 + ax * c2;
            o.jc[6 + k] = b * c1 + bx * c2;
        }
    }
}

// Wave-wide reductions on the VALU's data-parallel-primitive path instead of the LDS crossbar: a __shfl of a double is two
// ds_bpermute_b32 (~100 cycles of latency each, six dependent steps per reduction, and one LDS pipe per CU — a workgroup
// reducing 28 values per thread spent 17 us in them).  DPP moves are ordinary VALU instructions.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v, double fill)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(fill), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane63_f64(double v)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
// total valid in LANE 63 ONLY
__device__ __forceinline__ double wave_sum_lane63(double v)
{
    v += dpp_f64<0xB1, 0xf>(v, 0.0);   // quad_perm [1,0,3,2]: lane ^ 1
    v += dpp_f64<0x4E, 0xf>(v, 0.0);   // quad_perm [2,3,0,1]: lane ^ 2
    v += dpp_f64<0x141, 0xf>(v, 0.0);  // row_half_mirror: the other quad of the 8
    v += dpp_f64<0x140, 0xf>(v, 0.0);  // row_mirror: the other half of the row of 16 -> every lane holds its row's sum
    v += dpp_f64<0x142, 0xa>(v, 0.0);  // row_bcast:15 into rows 1 and 3
    v += dpp_f64<0x143, 0xc>(v, 0.0);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's sum
    return v;
}
// total valid in EVERY lane
__device__ __forceinline__ double wave_sum(double v) { return lane63_f64(wave_sum_lane63(v)); }
// maximum of non-negative values (the fill of the masked rows is 0), valid in every lane
__device__ __forceinline__ double wave_max_nonneg(double v)
{
    v = fmax(v, dpp_f64<0xB1, 0xf>(v, 0.0));
    v = fmax(v, dpp_f64<0x4E, 0xf>(v, 0.0));
    v = fmax(v, dpp_f64<0x141, 0xf>(v, 0.0));
    v = fmax(v, dpp_f64<0x140, 0xf>(v, 0.0));
    v = fmax(v, dpp_f64<0x142, 0xa>(v, 0.0));
    v = fmax(v, dpp_f64<0x143, 0xc>(v, 0.0));
    return lane63_f64(v);
}

// symmetric 3x3 inverse through Cholesky (InvertPSDMatrix<3>); false if not PD
__device__ __forceinline__ bool inv3_psd(const double V[6], double I[6])
{
    const double l00s = V[0];
    if (!(l00s > 0.0)) return false;
    const double l00 = sqrt(l00s);
    const double l10 = V[1] / l00, l20 = V[2] / l00;
    const double l11s = V[3] - l10 * l10;
    if (!(l11s > 0.0)) return false;
    const double l11 = sqrt(l11s);
    const double l21 = (V[4] - l20 * l10) / l11;
    const double l22s = V[5] - l20 * l20 - l21 * l21;
    if (!(l22s > 0.0)) return false;
    const double l22 = sqrt(l22s);
    const double i00 = 1.0 / l00, i11 = 1.0 / l11, i22 = 1.0 / l22;
    const double i10 = -l10 * i00 * i11;
    const double i21 = -l21 * i11 * i22;
    const double i20 = -(l20 * i00 + l21 * i10) * i22;
    I[0] = i00 * i00 + i10 * i10 + i20 * i20;
    I[1] = i10 * i11 + i20 * i21;
    I[2] = i20 * i22;
    I[3] = i11 * i11 + i21 * i21;
    I[4] = i21 * i22;
    I[5] = i22 * i22;
    return isfinite(I[0]) && isfinite(I[3]) && isfinite(I[5]);
}

__device__ __forceinline__ void atomic_max_nonneg(double* addr, double v)
{
    atomicMax((unsigned long long*)addr, (unsigned long long)__double_as_longlong(v));
}

__device__ __forceinline__ double clampd(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }

__device__ __forceinline__ double slot_sum(const double* base, int field);

// ---- the accept / reject decision of one LM step (TrustRegionMinimizer / LevenbergMarquardtStrategy,
// see oracle/ba.c).  There is no separate "decide" launch: the first kernel of iteration `it`
// applies the decision of iteration it-1 itself — every workgroup redundantly, from the previous
// state block and the previous slot sums (both immutable during this launch); workgroup 0 stores
// the result as this iteration's state.
__device__ __forceinline__ void ba_apply_decision(BaState& st, double cand, double mcc_p, double ssq_p, double xsq_p,
                                                  const BaOpt& opt, BaTrace* trace = nullptr)
{
    if (st.done) return;
    BaTrace* tr = trace ? trace + st.iter : nullptr;
    st.iter++;
    cand += st.cam_scal[3];       // cost of the camera-side (inertial) residual blocks at the candidate; 0 without them
    const double mcc = mcc_p + st.cam_scal[0];
    const double step_norm = sqrt(ssq_p + st.cam_scal[1]);
    const double x_norm = sqrt(xsq_p + st.cam_scal[2]);
    st.fresh = 0;
    const int consec_before = st.consec_accepts;
    st.consec_accepts = 0;
    if (tr) {
        tr->cost = st.x_cost; tr->candidate_cost = cand; tr->model_cost_change = mcc; tr->radius = st.radius;
        tr->step_norm = step_norm; tr->x_norm = x_norm; tr->outcome = 0; tr->pad = 0; tr->pad2 = 0.0;
    }
    if (st.solver_failed || !(mcc > 0.0)) {
        // TrustRegionMinimizer::HandleInvalidStep
        if (tr) { tr->outcome = -1; tr->candidate_cost = 0.0; tr->step_norm = 0.0; tr->x_norm = 0.0; if (st.solver_failed) tr->model_cost_change = 0.0; }
        if (++st.invalid_steps >= opt.max_invalid) { st.done = 1; st.termination = RS_BA_FAILURE; }
        else { st.radius /= st.decrease_factor; st.decrease_factor *= 2.0; }
    } else {
        st.invalid_steps = 0;
        if (step_norm <= opt.ptol * (x_norm + opt.ptol)) { st.done = 1; st.termination = RS_BA_CONVERGENCE_PARAMETER; if (tr) tr->outcome = 2; }
        else if (fabs(st.x_cost - cand) <= opt.ftol * st.x_cost) { st.done = 1; st.termination = RS_BA_CONVERGENCE_FUNCTION; if (tr) tr->outcome = 2; }
        else {
            const double rel = (st.x_cost - cand) / mcc;
            if (rel > opt.min_rel && isfinite(cand)) {
                if (tr) tr->outcome = 1;
                st.successful++;                // the caller moves st.cur to the accepted set's buffer
                st.consec_accepts = consec_before + 1;
                const double t = 2.0 * rel - 1.0;
                st.radius = st.radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
                st.radius = fmin(opt.rmax, st.radius);
                st.decrease_factor = 2.0;
                st.fresh = 1;
                st.x_cost = cand;   // replaced by the evaluation at the new point
            } else {
                st.radius /= st.decrease_factor;
                st.decrease_factor *= 2.0;
                if (st.radius < opt.rmin) { st.done = 1; st.termination = RS_BA_CONVERGENCE_RADIUS; }
            }
        }
    }
    st.solver_failed = 0;
    st.have_scale = 1;
    if (!st.done && st.iter >= opt.max_iter) { st.done = 1; st.termination = RS_BA_NO_CONVERGENCE; }
}

// fold the BA_NSLOT partial slots (one wave, lane = slot); result valid in every lane
__device__ __forceinline__ double slot_sum(const double* base, int field)
{
    return wave_sum(base[(size_t)(threadIdx.x & 63) * BA_SLOT_STRIDE + field]);
}
__device__ __forceinline__ double slot_max_bits(const double* base);
// maximum over the slot lines of every rank's block (called by the first wave)
__device__ __forceinline__ double slot_max_all(const BaBufs& b)
{
    double v = 0.0;
    for (int r = 0; r < b.gmax_blocks; r++) v = fmax(v, slot_max_bits(b.gmax_all + (size_t)r * BA_NSLOT * BA_SLOT_STRIDE));
    return v;
}
__device__ __forceinline__ double slot_max_bits(const double* base)
{
    return wave_max_nonneg(__longlong_as_double((long long)((const unsigned long long*)base)[(size_t)(threadIdx.x & 63) * BA_SLOT_STRIDE]));
}


// The decisions of the previous round, applied by the first wave of a workgroup (threadIdx.x < 64): the sets are
// walked in order — set k is LM iteration (iter + k) of the sequential loop — until a step is accepted or the
// loop terminates.  Lane 0 leaves the resulting state in *out.
__device__ __forceinline__ void ba_decide(const BaBufs& b, const BaOpt& opt, int it, BaTrace* trace, BaState* out,
                                          bool count_round = true)
{
    // K8's partial sums of the previous round, one slot line per lane: ALL loads go out first (clamped addresses, no
    // branches), then the wave reductions.  The sums were made by memory-side atomics, so every load is a ~1 us trip; issued
    // set by set behind `if (k < ns)` they were three (now five) round trips in a row and most of the round's decision.
    double ps[BA_MAXSETS][4];
    {
        const int lane = threadIdx.x & 63;
        double raw[BA_MAXSETS][4];
#pragma unroll
        for (int k = 0; k < BA_MAXSETS; k++) {
            const double* base = b.pt_prev + ((size_t)min(k, b.ns - 1) * BA_NSLOT + lane) * BA_SLOT_STRIDE;
#pragma unroll
            for (int f = 0; f < 4; f++) raw[k][f] = it > 0 ? base[f] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < BA_MAXSETS; k++) {
#pragma unroll
            for (int f = 0; f < 4; f++) ps[k][f] = 0.0;
            if (it > 0 && k < b.ns) {
#pragma unroll
                for (int f = 0; f < 4; f++) ps[k][f] = wave_sum(raw[k][f]);
            }
        }
    }
    // the walk over the sets is a ROLLED loop (one copy of the decision's code — divisions, square roots — instead of one per
    // set: the kernels that start with it load that code cold): lane 0 parks the sums in LDS and indexes them by set
    __shared__ double ps_sh[BA_MAXSETS][4];
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < BA_MAXSETS; k++)
#pragma unroll
            for (int f = 0; f < 4; f++) ps_sh[k][f] = ps[k][f];
        BaState s = *b.st_prev;
        if (it > 0) {
            const int nb = b.ns + 1;
            const int nact_prev = s.nact;
            bool accepted = false;
#pragma unroll 1
            for (int k = 0; k < nact_prev && !s.done; k++) {
                if (k > 0) {
                    const BaSetOut so = b.set_prev[k];
#pragma unroll
                    for (int q = 0; q < 4; q++) s.cam_scal[q] = so.cam_scal[q];
                    s.solver_failed = so.solver_failed;
                }
                const volatile double* pk = ps_sh[k];
                ba_apply_decision(s, pk[0], pk[1], pk[2], pk[3], opt, trace);
                if (s.fresh) {
                    if (s.calibrated == 1) s.calibrated = 2;       // an accepted step has resolved a rejection streak
                    s.cur = (s.cur + 1 + k) % nb;
                    accepted = true;
                    break;
                }
                if (s.calibrated == 0) s.calibrated = 1;           // first rejected / invalid step of the solve
            }
            // a round whose every set was rejected: the streak is longer than the round was deep
            if (!accepted && !s.done && s.calibrated == 2) s.calibrated = 1;
        }
        // How many radii this round speculates on.  The first step (initial radius 1e4: practically Gauss-Newton) and a
        // step after two successful ones in a row are most likely accepted — extra sets would be wasted work in K5 — so
        // those rounds evaluate one radius only.  Otherwise: while the radius is UNCALIBRATED (s.calibrated < 2) — it still is
        // Ceres' arbitrary initial 1e4 times the growth of the first accepted steps, no rejection streak has been resolved by
        // an accepted step yet, or the last round was rejected to its last set — a streak can be long (the radius shrinks by
        // 2, 8, 64, 1024 ...), and the round evaluates every set it has (5 by default); once calibrated, streaks are short
        // and three sets cover them (the benchmark window's A RRRR A RR A A: rounds of 1, 5, 3, 3 sets instead of 1, 3, 3, 3, 3).
        s.nact = (it == 0 || s.consec_accepts >= 2) ? 1 : (s.calibrated == 2 ? min(b.ns, BA_CALIBRATED_SETS) : b.ns);
        // never more sets than iterations the solve has left (its last round has often one: the benchmark window's fourth)
        s.nact = max(1, min(s.nact, opt.max_iter - s.iter));
        if (!s.done && count_round) { s.n_rounds++; s.n_fresh += s.fresh; s.n_sets += s.nact; }
        *out = s;
    }
}

// state of round `it` (called by all threads of a workgroup; one barrier).  Every workgroup recomputes the decision
// from the previous round's (immutable) blocks; the OWNER also stores it as this round's state block, writes the trace
// entries and tells the host.
__device__ __forceinline__ BaState ba_round_state(const BaBufs& b, const BaOpt& opt, int it, BaState* sh, bool owner)
{
    if (b.decided) {
        if (threadIdx.x == 0) *sh = *b.st;
        __syncthreads();
        return *sh;
    }
    if (threadIdx.x < 64) {
        ba_decide(b, opt, it, owner ? b.trace : nullptr, sh);
        if (threadIdx.x == 0 && owner) {
            *b.st = *sh;
            if (b.prog) {            // tell the host where the state machine is (pinned memory)
                b.prog->iter = sh->iter;
                b.prog->done = sh->done;
                __threadfence_system();
                b.prog->round = it + 1;
            }
        }
    }
    __syncthreads();
    return *sh;
}
__device__ __forceinline__ BaState ba_state_for_iteration(const BaBufs& b, const BaOpt& opt, int it, BaState* sh)
{
    return ba_round_state(b, opt, it, sh, blockIdx.x == 0);
}

// radius and decrease factor of speculative set k, given the state's (set 0)
__device__ __forceinline__ double ba_set_radius(const BaState& st, int k)
{
    double r = st.radius, f = st.decrease_factor;
    for (int q = 0; q < k; q++) { r /= f; f *= 2.0; }
    return r;
}

#ifndef RS_STAMPS
#define RS_STAMPS 0
#endif
#if RS_STAMPS
#define BA_STAMP_DECL                                                                \
    unsigned long long st_acc__[8] = {0, 0, 0, 0, 0, 0, 0, 0};                       \
    unsigned long long t_prev__ = (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) ? clock64() : 0
#define BA_STAMP(b, idx)                                                             \
    do {                                                                             \
        if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) {                                   \
            const unsigned long long t__ = clock64();                                \
            st_acc__[(idx) & 7] += t__ - t_prev__;                                   \
            t_prev__ = t__;                                                          \
        }                                                                            \
    } while (0)
#define BA_STAMP_FLUSH(b, base)                                                      \
    do {                                                                             \
        if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) {                                   \
            _Pragma("unroll") for (int q__ = 0; q__ < 8; q__++)                      \
                if (st_acc__[q__]) (b).dbg[(base) + q__] += st_acc__[q__];           \
        }                                                                            \
    } while (0)
#else
#define BA_STAMP_DECL do { } while (0)
#define BA_STAMP(b, idx) do { } while (0)
#define BA_STAMP_FLUSH(b, base) do { } while (0)
#endif

// ---- landmark grouping for the MFMA Schur kernel (ba_schur.hip)
struct BaGroup {
    int32_t* sorted;        // [P] landmark order
    int4* lm;               // [P] per sorted position: {landmark, first observation, observation count, 0}
    int32_t* obs_cs;        // [M] cam | (slot + 1) << 16 per observation
    int32_t* bucket;        // [P]
    uint64_t* mask;         // [P][2] free-slot bitmask of the landmark
    int32_t* hist;          // [Cf*Cf + 2] histogram / offsets
    int32_t* cursor;        // [Cf*Cf + 2]
    uint64_t* item_mask;    // [n_items][2]
    int n_items;
    int n_buckets;
    int it_l;               // landmarks per item (40 or 64)
    int32_t* maxspan;       // [1] largest (last - first) free-camera slot of a landmark: the block bandwidth of S (behind the cursors)
};


// ---- one window of a BATCHED solve (rs_bundle_adjust_batch, grid mode): the kernels of B independent windows run as
// ONE launch each, blockIdx.z = window; the per-window arguments live in a device array of these.  `b` holds the
// pointers as carved (round parity 0); the double-buffered blocks are re-pointed per round by ba_win_round.
struct BaWin {
    BaDims d;
    BaBufs b;
    BaGroup g;
    BaState* st_base; double* pts_base; BaSetOut* set_base; size_t pts_block;
    BaProgress* prog;
    const double* cams_in; const double* pts_in; unsigned long long free_mask; uint8_t* cam_free; int32_t* zero_ptr; int zero_n;
    double* cams_out; double* pts_out; BaState* h_st; BaTrace* h_trace; double* h_cams;
};
__device__ __forceinline__ BaBufs ba_win_round(const BaWin& w, int it, bool last)
{
    BaBufs b = w.b;
    b.st = w.st_base + (it & 1); b.st_prev = w.st_base + ((it + 1) & 1);
    b.pt_scal = w.pts_base + (size_t)(it & 1) * w.pts_block; b.pt_prev = w.pts_base + (size_t)((it + 1) & 1) * w.pts_block;
    b.set_out = w.set_base + (size_t)(it & 1) * BA_MAXSETS; b.set_prev = w.set_base + (size_t)((it + 1) & 1) * BA_MAXSETS;
    b.prog = last ? nullptr : w.prog;
    return b;
}
void ba_launch_grouping_batch(hipStream_t s, const BaWin* d_wins, int B, int max_P, int max_items);
void ba_launch_decide(hipStream_t s, const BaBufs& b, const BaOpt& opt, int it);                 // one wave: the round's decision -> *b.st (owner duties incl.)
void ba_launch_decide_batch(hipStream_t s, const BaWin* d_wins, int B, const BaOpt& opt, int it);
void ba_launch_schur_batch(hipStream_t s, const BaWin* d_wins, int B, const BaOpt& opt, int it, int max_items, int it_l, size_t lds);
void ba_launch_reduced_solve_lds_batch(hipStream_t s, const BaWin* d_wins, int B, const BaOpt& opt, int it, int ns, int max_n);
int ba_prepare_reduced_solve_lds_batch(int max_n);
int ba_prepare_schur_batch(size_t lds);
void ba_launch_backsub_batch(hipStream_t s, const BaWin* d_wins, int B, int it, int ns, int max_P, size_t lds);
void ba_group_set_items(BaGroup* g, int P, bool throughput, int batch_item = 0);   // landmarks per item: 32 (or batch_item) in throughput mode, else as ba_group_carve chose

size_t ba_group_bytes(int P, int Cf, int M);
void ba_group_carve(char* base, int P, int Cf, int M, BaGroup* g);
void ba_group_zero_range(const BaGroup& g, int32_t** ptr, int* count);
int ba_launch_grouping(rs_context* ctx, const BaDims& d, const BaBufs& b, const BaGroup& g);
bool ba_setup_fusable(const BaDims& d, const BaGroup& g);
void ba_launch_setup_fused(rs_context* ctx, const BaDims& d, const BaBufs& b, const BaOpt& opt, const BaGroup& g, const double* cams_in,
                           const double* pts_in, unsigned long long free_mask, int from_mask, uint8_t* cam_free);
size_t ba_schur_lds_bytes(int C, int Cf, int it_l = 64, int ns = BA_MAXSETS);   // ns: speculative sets of the solve (sizes the G tables)
int ba_prepare_schur(int C, int Cf);
void ba_launch_schur(hipStream_t s, const BaDims& d, const BaBufs& b, const BaOpt& opt, const BaGroup& g, int it);
// ---- blocked reduced solve for n > BA_MAX_LDS_N (ba_solve_big.hip)
struct rs_context;
size_t ba_big_bytes(int n);
int ba_launch_reduced_solve_big(rs_context* ctx, const BaDims& d, const BaBufs& b, const BaOpt& opt, char* ws, int band);   // band: 0 general blocked, 1 banded (one workgroup), 2 banded, two-sided
int ba_band_max_span();
// ---- inertial reduced solve (ba_solve_big.hip): N = 6 Cf + 9 Ci unknowns
size_t ba_inertial_bytes(int N, int n_fac, int C);
void ba_inertial_carve(char* ws, int N, int n_fac, int C, BaImu* imu, ImuFactorDev** d_fac, int32_t** d_inert);
int ba_launch_reduced_solve_inertial(rs_context* ctx, const BaDims& d, const BaBufs& b, const BaOpt& opt, char* ws);
// ---- inertial blocks eliminated around the LDS reduced solve (ba_imu.hip)
void ba_launch_imu_eliminate(hipStream_t s, const BaDims& d, const BaBufs& b, const BaOpt& opt);
void ba_launch_imu_expand(hipStream_t s, const BaDims& d, const BaBufs& b, const BaOpt& opt);
int ba_imu_lds_path_max_ci();
size_t ba_imu_lds_zacc_doubles(int Ci, int n);
size_t ba_imu_lds_total_doubles(int Ci, int n, int ns);
// ---- LDS-resident reduced solve (ba_solve.hip), n = 6*Cf <= BA_MAX_LDS_N
size_t ba_reduced_solve_lds_bytes(int n);
int ba_prepare_reduced_solve_lds(int n);
void ba_launch_reduced_solve_lds(hipStream_t s, const BaDims& d, const BaBufs& b, const BaOpt& opt);
int ba_solve_backsub_workgroups(const BaDims& d, const BaBufs& b, int n_cu);
void ba_launch_solve_backsub(hipStream_t s, const BaDims& d, const BaBufs& b, const BaOpt& opt, int n_cu);   // K7 + K8 in one launch (ba_solve.hip)
// ---- K5 + K7 + K8 of a round in one launch (ba_round.hip)
bool ba_round_eligible(const BaDims& d);
int ba_round_workgroups(const BaDims& d, const BaBufs& b, const BaGroup& g);
int ba_prepare_round(const BaDims& d);
void ba_launch_round(hipStream_t s, const BaDims& d, const BaBufs& b, const BaOpt& opt, const BaGroup& g, int it);
// ---- back-substitution + candidate cost (ba_update.hip)
size_t ba_backsub_lds_bytes(int C, int n);
void ba_launch_backsub(hipStream_t s, const BaDims& d, const BaBufs& b);
