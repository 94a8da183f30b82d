/*
 * oracle/ba.c — CPU restatement of optimization::bundle_adjust and
 * optimization::refine_pose (reference src/Optimization.cpp:21-72,118-142,
 * 194-267,269-374), vision-only (InertialInput{} default, src/Optimization.h:40-43).
 * TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see rs_oracle.h).
 *
 * Third-party semantics restated (Ceres 2.x, not in /root/reference):
 *  - AutoDiffCostFunction<ReprojectionError,2,6,3>: forward-mode jets through
 *    the functor, Jet arithmetic as in ceres/jet.h (division multiplies by the
 *    reciprocal, sqrt derivative 1/(2 sqrt)).
 *  - HuberLoss(a): rho(s) = s (s <= a^2) else 2 a sqrt(s) - a^2; Corrector with
 *    rho'' <= 0 scales residual and Jacobian by sqrt(rho').
 *  - TrustRegionMinimizer + LevenbergMarquardtStrategy with default options
 *    (SURVEY.md §8 a11): Jacobi scaling 1/(1+|col|) fixed at the first
 *    Jacobian; per step D = sqrt(clamp(|col|^2, 1e-6, 1e32) / radius) on the
 *    SCALED Jacobian; solve (J'J + D'D) y = J'r, step = -y;
 *    model_cost_change = -(J step).(r + J step / 2); parameter- and function-
 *    tolerance tests before the acceptance test; rho = cost change / model
 *    change; accepted: radius /= max(1/3, 1 - (2 rho - 1)^3), capped, decrease
 *    factor reset to 2; rejected: radius /= factor, factor *= 2.
 *  - SPARSE_SCHUR: point blocks eliminated, reduced camera system solved by a
 *    direct Cholesky (here dense LL^T; Ceres uses Eigen's sparse LDL^T — same
 *    solution up to rounding); DENSE_QR for refine_pose is replaced by the
 *    normal equations of the same 6-unknown least squares problem.
 *  - The user-visible state is only updated by successful steps; hitting the
 *    function/parameter tolerance returns without taking the candidate step.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "rs_oracle.h"

#define DBL_EPS 2.220446049250313e-16

/* ------------------------------------------------------------------ jets */
#define NJ 9
typedef struct { double a; double v[NJ]; } jet;

static jet j_const(double x) { jet r; r.a = x; memset(r.v, 0, sizeof r.v); return r; }
static jet j_var(double x, int k) { jet r = j_const(x); r.v[k] = 1.0; return r; }
static jet j_add(jet f, jet g) { jet r; r.a = f.a + g.a; for (int i = 0; i < NJ; i++) r.v[i] = f.v[i] + g.v[i]; return r; }
static jet j_sub(jet f, jet g) { jet r; r.a = f.a - g.a; for (int i = 0; i < NJ; i++) r.v[i] = f.v[i] - g.v[i]; return r; }
static jet j_mul(jet f, jet g) { jet r; r.a = f.a * g.a; for (int i = 0; i < NJ; i++) r.v[i] = f.a * g.v[i] + f.v[i] * g.a; return r; }
static jet j_div(jet f, jet g)
{
    jet r; const double gi = 1.0 / g.a; const double fg = f.a * gi;
    r.a = fg; for (int i = 0; i < NJ; i++) r.v[i] = (f.v[i] - fg * g.v[i]) * gi; return r;
}
static jet j_sqrt(jet f) { jet r; const double t = sqrt(f.a); const double h = 1.0 / (2.0 * t); r.a = t; for (int i = 0; i < NJ; i++) r.v[i] = f.v[i] * h; return r; }
static jet j_cos(jet f) { jet r; const double s = -sin(f.a); r.a = cos(f.a); for (int i = 0; i < NJ; i++) r.v[i] = s * f.v[i]; return r; }
static jet j_sin(jet f) { jet r; const double c = cos(f.a); r.a = sin(f.a); for (int i = 0; i < NJ; i++) r.v[i] = c * f.v[i]; return r; }

/* ceres::AngleAxisRotatePoint<Jet> */
static void rotate_point_jet(const jet aa[3], const jet pt[3], jet out[3])
{
    const jet theta2 = j_add(j_add(j_mul(aa[0], aa[0]), j_mul(aa[1], aa[1])), j_mul(aa[2], aa[2]));
    if (theta2.a > DBL_EPS) {
        const jet theta = j_sqrt(theta2);
        const jet c = j_cos(theta), s = j_sin(theta);
        const jet ti = j_div(j_const(1.0), theta);
        const jet w[3] = {j_mul(aa[0], ti), j_mul(aa[1], ti), j_mul(aa[2], ti)};
        const jet wxp[3] = {j_sub(j_mul(w[1], pt[2]), j_mul(w[2], pt[1])),
                            j_sub(j_mul(w[2], pt[0]), j_mul(w[0], pt[2])),
                            j_sub(j_mul(w[0], pt[1]), j_mul(w[1], pt[0]))};
        const jet dot = j_add(j_add(j_mul(w[0], pt[0]), j_mul(w[1], pt[1])), j_mul(w[2], pt[2]));
        const jet tmp = j_mul(dot, j_sub(j_const(1.0), c));
        for (int i = 0; i < 3; i++)
            out[i] = j_add(j_add(j_mul(pt[i], c), j_mul(wxp[i], s)), j_mul(w[i], tmp));
    } else {
        const jet wxp[3] = {j_sub(j_mul(aa[1], pt[2]), j_mul(aa[2], pt[1])),
                            j_sub(j_mul(aa[2], pt[0]), j_mul(aa[0], pt[2])),
                            j_sub(j_mul(aa[0], pt[1]), j_mul(aa[1], pt[0]))};
        for (int i = 0; i < 3; i++) out[i] = j_add(pt[i], wxp[i]);
    }
}

/* ReprojectionError::operator()<Jet>, src/Optimization.cpp:35-52 */
void orc_reprojection(const double cam[6], const double pt[3], const float uv[2],
                      const float K[4], double r[2], double jc[12], double jp[6])
{
    jet aa[3], c[3], X[3], centered[3], p[3];
    for (int i = 0; i < 3; i++) { aa[i] = j_var(cam[i], i); c[i] = j_var(cam[3 + i], 3 + i); X[i] = j_var(pt[i], 6 + i); }
    for (int i = 0; i < 3; i++) centered[i] = j_sub(X[i], c[i]);
    rotate_point_jet(aa, centered, p);
    jet r0 = j_sub(j_add(j_div(j_mul(j_const((double)K[0]), p[0]), p[2]), j_const((double)K[2])), j_const((double)uv[0]));
    jet r1 = j_sub(j_add(j_div(j_mul(j_const((double)K[1]), p[1]), p[2]), j_const((double)K[3])), j_const((double)uv[1]));
    r[0] = r0.a; r[1] = r1.a;
    for (int k = 0; k < 6; k++) { jc[k] = r0.v[k]; jc[6 + k] = r1.v[k]; }
    for (int k = 0; k < 3; k++) { jp[k] = r0.v[6 + k]; jp[3 + k] = r1.v[6 + k]; }
}

/* ReprojectionError::operator()<double> — the cost-only evaluation */
static void residual_only(const double cam[6], const double pt[3], const float uv[2],
                          const float K[4], double r[2])
{
    double centered[3] = {pt[0] - cam[3], pt[1] - cam[4], pt[2] - cam[5]}, p[3];
    orc_angle_axis_rotate_point(cam, centered, p);
    r[0] = (double)K[0] * p[0] / p[2] + (double)K[2] - (double)uv[0];
    r[1] = (double)K[1] * p[1] / p[2] + (double)K[3] - (double)uv[1];
}

/* ceres::HuberLoss::Evaluate */
static void huber(double a, double s, double rho[3])
{
    const double b = a * a;
    if (s > b) {
        const double r = sqrt(s);
        rho[0] = 2.0 * a * r - b;
        rho[1] = a / r; if (rho[1] < 2.2250738585072014e-308) rho[1] = 2.2250738585072014e-308;
        rho[2] = -rho[1] / (2.0 * s);
    } else {
        rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
    }
}

void orc_ba_default_options(orc_ba_options* o)
{
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

/* ------------------------------------------------------------------ trace */
/* (orc__trace_push is shared with pose_graph.c) */
static __thread orc_ba_iteration* g_trace = NULL;
static __thread int g_trace_cap = 0;
static __thread int* g_trace_count = NULL;
void orc_ba_set_trace(orc_ba_iteration* buf, int capacity, int* count)
{
    g_trace = buf; g_trace_cap = buf ? capacity : 0; g_trace_count = count;
    if (count) *count = 0;
}
void orc__trace_push(double cost, double cand, double mcc, double radius, double step_norm, double x_norm, int outcome)
{
    if (!g_trace || !g_trace_count || *g_trace_count >= g_trace_cap) return;
    orc_ba_iteration* e = g_trace + (*g_trace_count)++;
    e->cost = cost; e->candidate_cost = cand; e->model_cost_change = mcc; e->radius = radius;
    e->step_norm = step_norm; e->x_norm = x_norm; e->outcome = outcome; e->pad = 0;
}

/* ------------------------------------------------------------ small linalg */
/* in-place dense Cholesky A = L L^T (lower), returns 0 on success */
static int chol_factor(double* A, int n)
{
    for (int j = 0; j < n; j++) {
        double d = A[j * n + j];
        for (int k = 0; k < j; k++) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0.0) || !isfinite(d)) return 1;
        d = sqrt(d);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[i * n + j];
            for (int k = 0; k < j; k++) s -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = s / d;
        }
    }
    return 0;
}
static void chol_solve(const double* L, int n, double* b)
{
    for (int i = 0; i < n; i++) {
        double s = b[i];
        for (int k = 0; k < i; k++) s -= L[i * n + k] * b[k];
        b[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = b[i];
        for (int k = i + 1; k < n; k++) s -= L[k * n + i] * b[k];
        b[i] = s / L[i * n + i];
    }
}
/* InvertPSDMatrix<3>: llt().solve(Identity) */
static int inv3_psd(const double* A, double* inv)
{
    double L[9];
    memcpy(L, A, sizeof L);
    if (chol_factor(L, 3)) return 1;
    for (int c = 0; c < 3; c++) {
        double e[3] = {0, 0, 0};
        e[c] = 1.0;
        chol_solve(L, 3, e);
        for (int r = 0; r < 3; r++) inv[r * 3 + c] = e[r];
    }
    return 0;
}

/* ------------------------------------------------------------- the problem */
typedef struct {
    int C, P, M;
    int points_constant;        /* refine_pose: points are constant parameter blocks */
    const uint8_t* cam_free;
    const int32_t* obs_ptr;
    const int32_t* obs_cam;
    const float* obs_uv;
    const float* K;
    double huber_a;
    /* inertial residual blocks (src/Optimization.cpp:237-258,317-346); all NULL / 0 for a vision-only solve */
    int n_fac;                  /* IMU factor pairs: preintegration (9) + bias random walk (6) each */
    const orc_imu_factor* fac;
    const double* gravity;
    int delta_only;             /* refine_pose with an InertialDelta: the preintegration residual alone, frame i constant */
    const double* prev_pose;    /* [6] [3] [6]: the constant blocks of that residual */
    const double* prev_vel;
    const double* prev_bias;
    const double* prior_R;      /* refine_pose with a RotationPrior: predicted rotation (row-major), sigma */
    double prior_sigma;
    /* derived */
    int* cam_slot;              /* [C] index among active free cameras or -1 */
    int Cf;
    int* obs_pt;                /* [M] */
    int* inert_slot;            /* [C] index among the frames that carry velocity / bias blocks, or -1 */
    int Ci;
    int nc;                     /* camera-side unknowns: 6 Cf poses, then 9 per inertial frame (velocity 3, bias 6) */
    int n_extra;                /* extra residual blocks */
} problem;

/* ---- extra (camera-side) residual blocks: each has <= 9 residuals over <= 24 local parameters; col[k] is the
 * camera-side column of local parameter k or -1 when that parameter is constant / absent */
#define EX_RES 9
#define EX_PAR 24
typedef struct { int n_res; int n_par; int col[EX_PAR]; } extra_shape;

static int n_extras(const problem* pr)
{
    if (pr->prior_R) return 1;
    if (pr->delta_only) return 1;
    return 2 * pr->n_fac;
}

static int pose_col(const problem* pr, int c, int k) { return pr->cam_slot[c] >= 0 ? 6 * pr->cam_slot[c] + k : -1; }
static int vel_col(const problem* pr, int c, int k) { return pr->inert_slot[c] >= 0 ? 6 * pr->Cf + 9 * pr->inert_slot[c] + k : -1; }
static int bias_col(const problem* pr, int c, int k) { return pr->inert_slot[c] >= 0 ? 6 * pr->Cf + 9 * pr->inert_slot[c] + 3 + k : -1; }

/* residuals r[EX_RES] (+ Jacobian J[EX_RES][EX_PAR] when J != NULL) and the shape of extra block e at the given state */
static void eval_extra(const problem* pr, int e, const double* cams, const double* vel, const double* bias,
                       double* r, double* J, extra_shape* sh)
{
    memset(r, 0, sizeof(double) * EX_RES);
    if (J) memset(J, 0, sizeof(double) * EX_RES * EX_PAR);
    for (int k = 0; k < EX_PAR; k++) sh->col[k] = -1;
    if (pr->prior_R) {                                      /* PredictedRotationError on the one camera */
        double jj[18];
        orc_rotation_prior(pr->prior_R, pr->prior_sigma, cams, r, J ? jj : NULL);
        sh->n_res = 3; sh->n_par = 6;
        for (int k = 0; k < 6; k++) sh->col[k] = pose_col(pr, 0, k);
        if (J) for (int a = 0; a < 3; a++) for (int k = 0; k < 6; k++) J[a * EX_PAR + k] = jj[a * 6 + k];
        return;
    }
    if (pr->delta_only) {                                   /* refine_pose: previous frame constant, this frame = camera 0 */
        orc_imu_preintegration(pr->fac, pr->gravity, pr->prev_pose, pr->prev_vel, pr->prev_bias, cams, vel, r, J);
        sh->n_res = 9; sh->n_par = 24;
        for (int k = 0; k < 6; k++) sh->col[15 + k] = pose_col(pr, 0, k);
        for (int k = 0; k < 3; k++) sh->col[21 + k] = vel_col(pr, 0, k);
        return;
    }
    const orc_imu_factor* f = pr->fac + e / 2;
    const int i = f->cam_i, j = f->cam_j;
    if (e % 2 == 0) {                                       /* preintegration, src/Optimization.cpp:333-339 */
        orc_imu_preintegration(f, pr->gravity, cams + 6 * i, vel + 3 * i, bias + 6 * i, cams + 6 * j, vel + 3 * j, r, J);
        sh->n_res = 9; sh->n_par = 24;
        for (int k = 0; k < 6; k++) { sh->col[k] = pose_col(pr, i, k); sh->col[9 + k] = bias_col(pr, i, k); sh->col[15 + k] = pose_col(pr, j, k); }
        for (int k = 0; k < 3; k++) { sh->col[6 + k] = vel_col(pr, i, k); sh->col[21 + k] = vel_col(pr, j, k); }
    } else {                                                /* bias random walk, :340-344 */
        double jj[72];
        orc_imu_bias_walk(f, bias + 6 * i, bias + 6 * j, r, J ? jj : NULL);
        sh->n_res = 6; sh->n_par = 12;
        for (int k = 0; k < 6; k++) { sh->col[k] = bias_col(pr, i, k); sh->col[6 + k] = bias_col(pr, j, k); }
        if (J) for (int a = 0; a < 6; a++) for (int k = 0; k < 12; k++) J[a * EX_PAR + k] = jj[a * 12 + k];
    }
}

/* 1/2 sum |r|^2 of the extra blocks (no loss function: src/Optimization.cpp:246,254,334,341 pass nullptr) */
static double extras_cost(const problem* pr, const double* cams, const double* vel, const double* bias)
{
    double cost = 0.0, r[EX_RES];
    extra_shape sh;
    for (int e = 0; e < pr->n_extra; e++) {
        eval_extra(pr, e, cams, vel, bias, r, NULL, &sh);
        for (int a = 0; a < sh.n_res; a++) cost += 0.5 * r[a] * r[a];
    }
    return cost;
}

static double eval_cost(const problem* pr, const double* cams, const double* pts)
{
    double cost = 0.0;
#ifdef ORC_OMP      /* all-cores baseline build only (liboracle_omp.so); the summation order then differs */
#pragma omp parallel for schedule(static) reduction(+ : cost)
#endif
    for (int p = 0; p < pr->P; p++) {
        for (int o = pr->obs_ptr[p]; o < pr->obs_ptr[p + 1]; o++) {
            double r[2], rho[3];
            residual_only(cams + 6 * pr->obs_cam[o], pts + 3 * p, pr->obs_uv + 2 * o, pr->K, r);
            huber(pr->huber_a, r[0] * r[0] + r[1] * r[1], rho);
            cost += 0.5 * rho[0];
        }
    }
    return cost;
}

/* residuals + robustified Jacobians at (cams, pts); also the (unscaled) gradient max norm */
static double eval_jacobian(const problem* pr, const double* cams, const double* pts, const double* vel,
                            const double* bias, double* R, double* JC, double* JP, double* RF, double* JF,
                            extra_shape* SH, double* grad_max)
{
    double cost = 0.0;
    double* gc = (double*)calloc((size_t)pr->C * 6 + 1, sizeof(double));
    double gmax = 0.0;
#ifdef ORC_OMP      /* the autodiff of every observation in parallel first; the (cheap) gradient sums stay serial below */
#pragma omp parallel for schedule(static) reduction(+ : cost)
    for (int p = 0; p < pr->P; p++)
        for (int o = pr->obs_ptr[p]; o < pr->obs_ptr[p + 1]; o++) {
            double* r = R + 2 * o; double* jc = JC + 12 * o; double* jp = JP + 6 * o;
            orc_reprojection(cams + 6 * pr->obs_cam[o], pts + 3 * p, pr->obs_uv + 2 * o, pr->K, r, jc, jp);
            double rho[3];
            huber(pr->huber_a, r[0] * r[0] + r[1] * r[1], rho);
            cost += 0.5 * rho[0];
            const double sr = sqrt(rho[1]);
            for (int k = 0; k < 12; k++) jc[k] *= sr;
            for (int k = 0; k < 6; k++) jp[k] *= sr;
            r[0] *= sr; r[1] *= sr;
        }
#endif
    for (int p = 0; p < pr->P; p++) {
        double gp[3] = {0, 0, 0};
        for (int o = pr->obs_ptr[p]; o < pr->obs_ptr[p + 1]; o++) {
            double* r = R + 2 * o; double* jc = JC + 12 * o; double* jp = JP + 6 * o;
            const int c = pr->obs_cam[o];
#ifndef ORC_OMP
            orc_reprojection(cams + 6 * c, pts + 3 * p, pr->obs_uv + 2 * o, pr->K, r, jc, jp);
            double rho[3];
            huber(pr->huber_a, r[0] * r[0] + r[1] * r[1], rho);
            cost += 0.5 * rho[0];
            const double sr = sqrt(rho[1]);          /* Corrector, rho'' <= 0 */
            for (int k = 0; k < 12; k++) jc[k] *= sr;
            for (int k = 0; k < 6; k++) jp[k] *= sr;
            r[0] *= sr; r[1] *= sr;
#endif
            for (int k = 0; k < 6; k++) gc[6 * c + k] += jc[k] * r[0] + jc[6 + k] * r[1];
            for (int k = 0; k < 3; k++) gp[k] += jp[k] * r[0] + jp[3 + k] * r[1];
        }
        if (!pr->points_constant)
            for (int k = 0; k < 3; k++) if (fabs(gp[k]) > gmax) gmax = fabs(gp[k]);
    }
    /* camera-side gradient by column: reprojection part + the extra residual blocks */
    double* gcol = (double*)calloc((size_t)pr->nc + 1, sizeof(double));
    for (int c = 0; c < pr->C; c++)
        if (pr->cam_slot[c] >= 0)
            for (int k = 0; k < 6; k++) gcol[6 * pr->cam_slot[c] + k] = gc[6 * c + k];
    for (int e = 0; e < pr->n_extra; e++) {
        double* r = RF + EX_RES * e; double* J = JF + EX_RES * EX_PAR * e;
        eval_extra(pr, e, cams, vel, bias, r, J, SH + e);
        for (int a = 0; a < SH[e].n_res; a++) {
            cost += 0.5 * r[a] * r[a];
            for (int k = 0; k < SH[e].n_par; k++)
                if (SH[e].col[k] >= 0) gcol[SH[e].col[k]] += J[a * EX_PAR + k] * r[a];
        }
    }
    for (int i = 0; i < pr->nc; i++) if (fabs(gcol[i]) > gmax) gmax = fabs(gcol[i]);
    free(gcol);
    free(gc);
    *grad_max = gmax;
    return cost;
}

static int lm_solve(problem* pr, double* cams, double* pts, double* vel, double* bias, const orc_ba_options* opt,
                    orc_ba_summary* sum)
{
    const int C = pr->C, P = pr->P, M = pr->M;
    /* active free cameras = free and carrying at least one residual block (reprojection or inertial);
     * inertial frames = frames whose velocity / bias blocks appear in a residual block */
    pr->cam_slot = (int*)malloc(sizeof(int) * (size_t)(C > 0 ? C : 1));
    pr->inert_slot = (int*)malloc(sizeof(int) * (size_t)(C > 0 ? C : 1));
    pr->obs_pt = (int*)malloc(sizeof(int) * (size_t)(M > 0 ? M : 1));
    int* cam_nobs = (int*)calloc((size_t)C + 1, sizeof(int));
    int* cam_inert = (int*)calloc((size_t)C + 1, sizeof(int));
    for (int p = 0; p < P; p++)
        for (int o = pr->obs_ptr[p]; o < pr->obs_ptr[p + 1]; o++) { pr->obs_pt[o] = p; cam_nobs[pr->obs_cam[o]]++; }
    if (pr->prior_R) cam_nobs[0]++;
    if (pr->delta_only) { cam_nobs[0]++; cam_inert[0] = 1; }
    else for (int f = 0; f < pr->n_fac; f++) {
        cam_nobs[pr->fac[f].cam_i]++; cam_nobs[pr->fac[f].cam_j]++;
        cam_inert[pr->fac[f].cam_i] = 1; cam_inert[pr->fac[f].cam_j] = 1;
    }
    pr->Cf = 0; pr->Ci = 0;
    for (int c = 0; c < C; c++) pr->cam_slot[c] = (pr->cam_free[c] && cam_nobs[c] > 0) ? pr->Cf++ : -1;
    for (int c = 0; c < C; c++) pr->inert_slot[c] = cam_inert[c] ? pr->Ci++ : -1;
    free(cam_nobs); free(cam_inert);
    pr->n_extra = n_extras(pr);
    const int Cf = pr->Cf, nc = 6 * Cf + 9 * pr->Ci, NE = pr->n_extra;
    pr->nc = nc;
    const int NP = pr->points_constant ? 0 : P;
    double zero9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (!vel) vel = zero9;      /* vision-only: never read (no inertial frames) */
    if (!bias) bias = zero9;
    const size_t nvb = pr->Ci > 0 ? (size_t)C : 1;

    double* R = (double*)malloc(sizeof(double) * 2 * (size_t)(M + 1));
    double* JC = (double*)malloc(sizeof(double) * 12 * (size_t)(M + 1));
    double* JP = (double*)malloc(sizeof(double) * 6 * (size_t)(M + 1));
    double* RF = (double*)calloc((size_t)EX_RES * (NE + 1), sizeof(double));
    double* JF = (double*)calloc((size_t)EX_RES * EX_PAR * (NE + 1), sizeof(double));
    extra_shape* SH = (extra_shape*)calloc((size_t)NE + 1, sizeof(extra_shape));
    double* scale_c = (double*)malloc(sizeof(double) * (size_t)(nc + 1));
    double* scale_p = (double*)malloc(sizeof(double) * 3 * (size_t)(NP + 1));
    double* diag_c = (double*)malloc(sizeof(double) * (size_t)(nc + 1));
    double* diag_p = (double*)malloc(sizeof(double) * 3 * (size_t)(NP + 1));
    double* S = (double*)malloc(sizeof(double) * ((size_t)nc * nc + 1));
    double* rhs = (double*)malloc(sizeof(double) * (size_t)(nc + 1));
    double* Vinv = (double*)malloc(sizeof(double) * 9 * (size_t)(NP + 1));
    double* etb = (double*)malloc(sizeof(double) * 3 * (size_t)(NP + 1));
    double* step_c = (double*)malloc(sizeof(double) * (size_t)(nc + 1));
    double* step_p = (double*)malloc(sizeof(double) * 3 * (size_t)(NP + 1));
    double* cand_c = (double*)malloc(sizeof(double) * 6 * (size_t)(C + 1));
    double* cand_p = (double*)malloc(sizeof(double) * 3 * (size_t)(P + 1));
    double* cand_v = (double*)malloc(sizeof(double) * 3 * (nvb + 1));
    double* cand_b = (double*)malloc(sizeof(double) * 6 * (nvb + 1));
    double* x_c = (double*)malloc(sizeof(double) * 6 * (size_t)(C + 1));
    double* x_p = (double*)malloc(sizeof(double) * 3 * (size_t)(P + 1));
    double* x_v = (double*)malloc(sizeof(double) * 3 * (nvb + 1));
    double* x_b = (double*)malloc(sizeof(double) * 6 * (nvb + 1));
    memcpy(x_c, cams, sizeof(double) * 6 * (size_t)C);
    memcpy(x_p, pts, sizeof(double) * 3 * (size_t)P);
    if (pr->Ci > 0) { memcpy(x_v, vel, sizeof(double) * 3 * (size_t)C); memcpy(x_b, bias, sizeof(double) * 6 * (size_t)C); }

    double radius = opt->initial_trust_region_radius, decrease_factor = 2.0;
    int invalid_steps = 0;
    double grad_max;
    double x_cost = eval_jacobian(pr, x_c, x_p, x_v, x_b, R, JC, JP, RF, JF, SH, &grad_max);
    sum->initial_cost = x_cost; sum->final_cost = x_cost;
    sum->iterations = 0; sum->successful_steps = 0; sum->termination = 0;
    double minimum_cost = x_cost;

    /* Jacobi scaling from the first Jacobian (TrustRegionMinimizer::IterationZero) */
    for (int i = 0; i < nc; i++) scale_c[i] = 0.0;
    for (int i = 0; i < 3 * NP; i++) scale_p[i] = 0.0;
    for (int o = 0; o < M; o++) {
        const int s = pr->cam_slot[pr->obs_cam[o]];
        if (s >= 0) for (int k = 0; k < 6; k++) scale_c[6 * s + k] += JC[12 * o + k] * JC[12 * o + k] + JC[12 * o + 6 + k] * JC[12 * o + 6 + k];
        if (NP) for (int k = 0; k < 3; k++) scale_p[3 * pr->obs_pt[o] + k] += JP[6 * o + k] * JP[6 * o + k] + JP[6 * o + 3 + k] * JP[6 * o + 3 + k];
    }
    for (int e = 0; e < NE; e++)
        for (int a = 0; a < SH[e].n_res; a++)
            for (int k = 0; k < SH[e].n_par; k++)
                if (SH[e].col[k] >= 0) scale_c[SH[e].col[k]] += JF[(e * EX_RES + a) * EX_PAR + k] * JF[(e * EX_RES + a) * EX_PAR + k];
    for (int i = 0; i < nc; i++) scale_c[i] = opt->jacobi_scaling ? 1.0 / (1.0 + sqrt(scale_c[i])) : 1.0;
    for (int i = 0; i < 3 * NP; i++) scale_p[i] = opt->jacobi_scaling ? 1.0 / (1.0 + sqrt(scale_p[i])) : 1.0;

    int jac_needs_scaling = 1;
    int done = 0;
    if (!isfinite(x_cost)) { sum->termination = 5; done = 1; }
    else if (grad_max <= opt->gradient_tolerance) { sum->termination = 3; done = 1; }

    while (!done) {
        if (sum->iterations >= opt->max_num_iterations) { sum->termination = 0; break; }
        sum->iterations++;
        if (jac_needs_scaling) {
            for (int o = 0; o < M; o++) {
                const int s = pr->cam_slot[pr->obs_cam[o]];
                if (s >= 0) for (int k = 0; k < 6; k++) { JC[12 * o + k] *= scale_c[6 * s + k]; JC[12 * o + 6 + k] *= scale_c[6 * s + k]; }
                if (NP) for (int k = 0; k < 3; k++) { JP[6 * o + k] *= scale_p[3 * pr->obs_pt[o] + k]; JP[6 * o + 3 + k] *= scale_p[3 * pr->obs_pt[o] + k]; }
            }
            for (int e = 0; e < NE; e++)
                for (int a = 0; a < SH[e].n_res; a++)
                    for (int k = 0; k < SH[e].n_par; k++)
                        if (SH[e].col[k] >= 0) JF[(e * EX_RES + a) * EX_PAR + k] *= scale_c[SH[e].col[k]];
            jac_needs_scaling = 0;
        }
        /* LevenbergMarquardtStrategy::ComputeStep */
        for (int i = 0; i < nc; i++) diag_c[i] = 0.0;
        for (int i = 0; i < 3 * NP; i++) diag_p[i] = 0.0;
        for (int o = 0; o < M; o++) {
            const int s = pr->cam_slot[pr->obs_cam[o]];
            if (s >= 0) for (int k = 0; k < 6; k++) diag_c[6 * s + k] += JC[12 * o + k] * JC[12 * o + k] + JC[12 * o + 6 + k] * JC[12 * o + 6 + k];
            if (NP) for (int k = 0; k < 3; k++) diag_p[3 * pr->obs_pt[o] + k] += JP[6 * o + k] * JP[6 * o + k] + JP[6 * o + 3 + k] * JP[6 * o + 3 + k];
        }
        for (int e = 0; e < NE; e++)
            for (int a = 0; a < SH[e].n_res; a++)
                for (int k = 0; k < SH[e].n_par; k++)
                    if (SH[e].col[k] >= 0) diag_c[SH[e].col[k]] += JF[(e * EX_RES + a) * EX_PAR + k] * JF[(e * EX_RES + a) * EX_PAR + k];
        for (int i = 0; i < nc; i++) diag_c[i] = fmin(fmax(diag_c[i], opt->min_lm_diagonal), opt->max_lm_diagonal) / radius;
        for (int i = 0; i < 3 * NP; i++) diag_p[i] = fmin(fmax(diag_p[i], opt->min_lm_diagonal), opt->max_lm_diagonal) / radius;
        /* (diag_* now hold D^2) */

        /* Schur elimination of the point blocks */
        memset(S, 0, sizeof(double) * ((size_t)nc * nc + 1));
        memset(rhs, 0, sizeof(double) * (size_t)(nc + 1));
        for (int i = 0; i < nc; i++) S[(size_t)i * nc + i] = diag_c[i];
        for (int o = 0; o < M; o++) {
            const int s = pr->cam_slot[pr->obs_cam[o]];
            if (s < 0) continue;
            const double* jc = JC + 12 * o; const double* r = R + 2 * o;
            for (int a = 0; a < 6; a++) {
                for (int b = 0; b < 6; b++) S[(size_t)(6 * s + a) * nc + 6 * s + b] += jc[a] * jc[b] + jc[6 + a] * jc[6 + b];
                rhs[6 * s + a] += jc[a] * r[0] + jc[6 + a] * r[1];
            }
        }
        /* the extra residual blocks only touch camera-side columns: J^T J and J^T r go straight into the reduced system */
        for (int e = 0; e < NE; e++)
            for (int a = 0; a < SH[e].n_res; a++) {
                const double* jr = JF + (e * EX_RES + a) * EX_PAR;
                for (int k = 0; k < SH[e].n_par; k++) {
                    const int ck = SH[e].col[k];
                    if (ck < 0) continue;
                    rhs[ck] += jr[k] * RF[e * EX_RES + a];
                    for (int l = 0; l < SH[e].n_par; l++)
                        if (SH[e].col[l] >= 0) S[(size_t)ck * nc + SH[e].col[l]] += jr[k] * jr[l];
                }
            }
        int solver_failed = 0;
#ifdef ORC_OMP      /* landmark blocks in parallel, per-thread copies of S and rhs summed at the end */
#pragma omp parallel for schedule(static) reduction(+ : S[: nc * nc], rhs[: nc]) reduction(| : solver_failed)
        for (int p = 0; p < NP; p++) {
#else
        for (int p = 0; p < NP && !solver_failed; p++) {
#endif
            double ete[9] = {0}, eb[3] = {0};
            const int o0 = pr->obs_ptr[p], o1 = pr->obs_ptr[p + 1];
            for (int o = o0; o < o1; o++) {
                const double* jp = JP + 6 * o; const double* r = R + 2 * o;
                for (int a = 0; a < 3; a++) {
                    for (int b = 0; b < 3; b++) ete[a * 3 + b] += jp[a] * jp[b] + jp[3 + a] * jp[3 + b];
                    eb[a] += jp[a] * r[0] + jp[3 + a] * r[1];
                }
            }
            for (int a = 0; a < 3; a++) ete[a * 3 + a] += diag_p[3 * p + a];
#ifdef ORC_OMP
            if (inv3_psd(ete, Vinv + 9 * p)) { solver_failed = 1; continue; }
#else
            if (inv3_psd(ete, Vinv + 9 * p)) { solver_failed = 1; break; }
#endif
            memcpy(etb + 3 * p, eb, sizeof eb);
            const double* vi = Vinv + 9 * p;
            for (int oi = o0; oi < o1; oi++) {
                const int si = pr->cam_slot[pr->obs_cam[oi]];
                if (si < 0) continue;
                /* W_i = Jc_i^T Jp_i (6x3), Y = W_i Vinv */
                double Wi[18], Y[18];
                for (int a = 0; a < 6; a++)
                    for (int d = 0; d < 3; d++) Wi[a * 3 + d] = JC[12 * oi + a] * JP[6 * oi + d] + JC[12 * oi + 6 + a] * JP[6 * oi + 3 + d];
                for (int a = 0; a < 6; a++)
                    for (int d = 0; d < 3; d++) Y[a * 3 + d] = Wi[a * 3] * vi[d] + Wi[a * 3 + 1] * vi[3 + d] + Wi[a * 3 + 2] * vi[6 + d];
                for (int a = 0; a < 6; a++) rhs[6 * si + a] -= Y[a * 3] * eb[0] + Y[a * 3 + 1] * eb[1] + Y[a * 3 + 2] * eb[2];
                for (int oj = o0; oj < o1; oj++) {
                    const int sj = pr->cam_slot[pr->obs_cam[oj]];
                    if (sj < 0) continue;
                    for (int a = 0; a < 6; a++)
                        for (int b = 0; b < 6; b++) {
                            double wjb[3];
                            for (int d = 0; d < 3; d++) wjb[d] = JC[12 * oj + b] * JP[6 * oj + d] + JC[12 * oj + 6 + b] * JP[6 * oj + 3 + d];
                            S[(size_t)(6 * si + a) * nc + 6 * sj + b] -= Y[a * 3] * wjb[0] + Y[a * 3 + 1] * wjb[1] + Y[a * 3 + 2] * wjb[2];
                        }
                }
            }
        }
        if (!solver_failed && nc > 0) {
            if (chol_factor(S, nc)) solver_failed = 1;
            else { memcpy(step_c, rhs, sizeof(double) * (size_t)nc); chol_solve(S, nc, step_c); }
        }
        if (!solver_failed) {
            /* back substitution */
#ifdef ORC_OMP
#pragma omp parallel for schedule(static)
#endif
            for (int p = 0; p < NP; p++) {
                double t[3] = {etb[3 * p], etb[3 * p + 1], etb[3 * p + 2]};
                for (int o = pr->obs_ptr[p]; o < pr->obs_ptr[p + 1]; o++) {
                    const int s = pr->cam_slot[pr->obs_cam[o]];
                    if (s < 0) continue;
                    const double* jc = JC + 12 * o; const double* jp = JP + 6 * o;
                    double m0 = 0, m1 = 0;
                    for (int a = 0; a < 6; a++) { m0 += jc[a] * step_c[6 * s + a]; m1 += jc[6 + a] * step_c[6 * s + a]; }
                    for (int d = 0; d < 3; d++) t[d] -= jp[d] * m0 + jp[3 + d] * m1;
                }
                const double* vi = Vinv + 9 * p;
                for (int d = 0; d < 3; d++) step_p[3 * p + d] = vi[d * 3] * t[0] + vi[d * 3 + 1] * t[1] + vi[d * 3 + 2] * t[2];
            }
            for (int i = 0; i < nc; i++) { if (!isfinite(step_c[i])) solver_failed = 1; step_c[i] = -step_c[i]; }
            for (int i = 0; i < 3 * NP; i++) { if (!isfinite(step_p[i])) solver_failed = 1; step_p[i] = -step_p[i]; }
        }
        double model_cost_change = 0.0;
        if (!solver_failed) {
#ifdef ORC_OMP
#pragma omp parallel for schedule(static) reduction(+ : model_cost_change)
#endif
            for (int o = 0; o < M; o++) {
                const int s = pr->cam_slot[pr->obs_cam[o]];
                double m0 = 0, m1 = 0;
                if (s >= 0) for (int a = 0; a < 6; a++) { m0 += JC[12 * o + a] * step_c[6 * s + a]; m1 += JC[12 * o + 6 + a] * step_c[6 * s + a]; }
                if (NP) for (int d = 0; d < 3; d++) { m0 += JP[6 * o + d] * step_p[3 * pr->obs_pt[o] + d]; m1 += JP[6 * o + 3 + d] * step_p[3 * pr->obs_pt[o] + d]; }
                model_cost_change -= m0 * (R[2 * o] + m0 / 2.0) + m1 * (R[2 * o + 1] + m1 / 2.0);
            }
            for (int e = 0; e < NE; e++)
                for (int a = 0; a < SH[e].n_res; a++) {
                    double m = 0.0;
                    for (int k = 0; k < SH[e].n_par; k++)
                        if (SH[e].col[k] >= 0) m += JF[(e * EX_RES + a) * EX_PAR + k] * step_c[SH[e].col[k]];
                    model_cost_change -= m * (RF[e * EX_RES + a] + m / 2.0);
                }
        }
        if (solver_failed || !(model_cost_change > 0.0)) {
            /* TrustRegionMinimizer::HandleInvalidStep */
            orc__trace_push(x_cost, 0.0, solver_failed ? 0.0 : model_cost_change, radius, 0.0, 0.0, -1);
            if (++invalid_steps >= opt->max_num_consecutive_invalid_steps) { sum->termination = 5; break; }
            radius = radius / decrease_factor; decrease_factor *= 2.0;   /* StepIsInvalid -> StepRejected(0) */
            continue;
        }
        invalid_steps = 0;
        /* candidate = x + scaling .* step */
        memcpy(cand_c, x_c, sizeof(double) * 6 * (size_t)C);
        memcpy(cand_p, x_p, sizeof(double) * 3 * (size_t)P);
        if (pr->Ci > 0) { memcpy(cand_v, x_v, sizeof(double) * 3 * (size_t)C); memcpy(cand_b, x_b, sizeof(double) * 6 * (size_t)C); }
        double step_sq = 0.0, x_sq = 0.0;
        for (int c = 0; c < C; c++) {
            const int s = pr->cam_slot[c];
            if (s < 0) continue;
            for (int k = 0; k < 6; k++) {
                const double d = step_c[6 * s + k] * scale_c[6 * s + k];
                cand_c[6 * c + k] = x_c[6 * c + k] + d;
                const double diff = x_c[6 * c + k] - cand_c[6 * c + k];
                step_sq += diff * diff; x_sq += x_c[6 * c + k] * x_c[6 * c + k];
            }
        }
        for (int c = 0; c < C; c++) {
            const int q = pr->inert_slot[c];
            if (q < 0) continue;
            for (int k = 0; k < 9; k++) {
                const int col = 6 * Cf + 9 * q + k;
                double* xs = k < 3 ? &x_v[3 * c + k] : &x_b[6 * c + k - 3];
                double* cs = k < 3 ? &cand_v[3 * c + k] : &cand_b[6 * c + k - 3];
                *cs = *xs + step_c[col] * scale_c[col];
                const double diff = *xs - *cs;
                step_sq += diff * diff; x_sq += *xs * *xs;
            }
        }
        for (int i = 0; i < 3 * NP; i++) {
            const double d = step_p[i] * scale_p[i];
            cand_p[i] = x_p[i] + d;
            const double diff = x_p[i] - cand_p[i];
            step_sq += diff * diff; x_sq += x_p[i] * x_p[i];
        }
        const double cand_cost = eval_cost(pr, cand_c, cand_p) + extras_cost(pr, cand_c, cand_v, cand_b);
        const double step_norm = sqrt(step_sq), x_norm = sqrt(x_sq);
        /* ParameterToleranceReached */
        if (step_norm <= opt->parameter_tolerance * (x_norm + opt->parameter_tolerance)) {
            orc__trace_push(x_cost, cand_cost, model_cost_change, radius, step_norm, x_norm, 2);
            sum->termination = 2; break;
        }
        /* FunctionToleranceReached */
        if (fabs(x_cost - cand_cost) <= opt->function_tolerance * x_cost) {
            orc__trace_push(x_cost, cand_cost, model_cost_change, radius, step_norm, x_norm, 2);
            sum->termination = 1; break;
        }
        const double rel = (x_cost - cand_cost) / model_cost_change;
        orc__trace_push(x_cost, cand_cost, model_cost_change, radius, step_norm, x_norm,
                   (rel > opt->min_relative_decrease && isfinite(cand_cost)) ? 1 : 0);
        if (rel > opt->min_relative_decrease && isfinite(cand_cost)) {
            /* HandleSuccessfulStep */
            memcpy(x_c, cand_c, sizeof(double) * 6 * (size_t)C);
            memcpy(x_p, cand_p, sizeof(double) * 3 * (size_t)P);
            if (pr->Ci > 0) { memcpy(x_v, cand_v, sizeof(double) * 3 * (size_t)C); memcpy(x_b, cand_b, sizeof(double) * 6 * (size_t)C); }
            x_cost = eval_jacobian(pr, x_c, x_p, x_v, x_b, R, JC, JP, RF, JF, SH, &grad_max);
            jac_needs_scaling = 1;
            sum->successful_steps++;
            radius = radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rel - 1.0, 3));
            radius = fmin(opt->max_trust_region_radius, radius);
            decrease_factor = 2.0;
            if (x_cost < minimum_cost) {
                minimum_cost = x_cost;
                memcpy(cams, x_c, sizeof(double) * 6 * (size_t)C);   /* user state follows the best iterate */
                memcpy(pts, x_p, sizeof(double) * 3 * (size_t)P);
                if (pr->Ci > 0) { memcpy(vel, x_v, sizeof(double) * 3 * (size_t)C); memcpy(bias, x_b, sizeof(double) * 6 * (size_t)C); }
            }
            if (grad_max <= opt->gradient_tolerance) { sum->termination = 3; break; }
        } else {
            radius = radius / decrease_factor; decrease_factor *= 2.0;
            if (radius < opt->min_trust_region_radius) { sum->termination = 4; break; }
        }
    }
    sum->final_cost = minimum_cost;
    sum->final_radius = radius;
    /* solve()'s accept rule, src/Optimization.cpp:136-141 */
    sum->usable = (sum->termination != 5) && isfinite(sum->final_cost) && sum->final_cost <= sum->initial_cost;

    free(pr->cam_slot); free(pr->inert_slot); free(pr->obs_pt);
    free(R); free(JC); free(JP); free(RF); free(JF); free(SH); free(scale_c); free(scale_p); free(diag_c); free(diag_p);
    free(S); free(rhs); free(Vinv); free(etb); free(step_c); free(step_p);
    free(cand_c); free(cand_p); free(cand_v); free(cand_b); free(x_c); free(x_p); free(x_v); free(x_b);
    return 0;
}

int orc_bundle_adjust(int n_cameras, int n_points, int n_obs, double* cameras,
                      const uint8_t* cam_free, double* points, const int32_t* obs_ptr,
                      const int32_t* obs_cam, const float* obs_uv, const float K[4],
                      const orc_ba_options* options, orc_ba_summary* summary)
{
    orc_ba_options def;
    if (!options) { orc_ba_default_options(&def); options = &def; }
    if (n_cameras < 0 || n_points < 0 || n_obs < 0) return 1;
    problem pr;
    memset(&pr, 0, sizeof pr);
    pr.C = n_cameras; pr.P = n_points; pr.M = n_obs; pr.points_constant = 0;
    pr.cam_free = cam_free; pr.obs_ptr = obs_ptr; pr.obs_cam = obs_cam; pr.obs_uv = obs_uv;
    pr.K = K; pr.huber_a = options->huber_delta;
    /* work on copies; write back only if usable (src/Optimization.cpp:360-372) */
    double* c = (double*)malloc(sizeof(double) * 6 * (size_t)(n_cameras + 1));
    double* p = (double*)malloc(sizeof(double) * 3 * (size_t)(n_points + 1));
    memcpy(c, cameras, sizeof(double) * 6 * (size_t)n_cameras);
    memcpy(p, points, sizeof(double) * 3 * (size_t)n_points);
    lm_solve(&pr, c, p, NULL, NULL, options, summary);
    if (summary->usable) {
        for (int i = 0; i < n_cameras; i++)
            if (cam_free[i]) memcpy(cameras + 6 * i, c + 6 * i, 6 * sizeof(double));
        memcpy(points, p, sizeof(double) * 3 * (size_t)n_points);
    }
    free(c); free(p);
    return 0;
}

int orc_refine_pose(double camera[6], const double* points, const float* uv, int n,
                    const float K[4], const orc_ba_options* options, orc_ba_summary* summary)
{
    orc_ba_options def;
    if (!options) { orc_ba_default_options(&def); options = &def; }
    memset(summary, 0, sizeof *summary);
    if (n <= 0) return 0;                       /* src/Optimization.cpp:227-229 */
    problem pr;
    memset(&pr, 0, sizeof pr);
    pr.C = 1; pr.P = n; pr.M = n; pr.points_constant = 1;
    uint8_t fr = 1;
    int32_t* ptr = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1));
    int32_t* cam = (int32_t*)calloc((size_t)n, sizeof(int32_t));
    for (int i = 0; i <= n; i++) ptr[i] = i;
    pr.cam_free = &fr; pr.obs_ptr = ptr; pr.obs_cam = cam; pr.obs_uv = uv; pr.K = K;
    pr.huber_a = options->huber_delta;
    double c[6];
    memcpy(c, camera, sizeof c);
    double* p = (double*)malloc(sizeof(double) * 3 * (size_t)n);
    memcpy(p, points, sizeof(double) * 3 * (size_t)n);
    lm_solve(&pr, c, p, NULL, NULL, options, summary);
    if (summary->usable) memcpy(camera, c, sizeof c);
    free(ptr); free(cam); free(p);
    return 0;
}

/* bundle_adjust with InertialInput::usable() (src/Optimization.cpp:317-346): one preintegration + one bias-walk
 * residual block per factor; velocity [C][3] / bias [C][6] are written back for the free frames on a usable solve
 * (unpack_inertial, :363-368). */
int orc_bundle_adjust_inertial(int n_cameras, int n_points, int n_obs, double* cameras, const uint8_t* cam_free,
                               double* points, const int32_t* obs_ptr, const int32_t* obs_cam, const float* obs_uv,
                               const float K[4], double* velocity, double* bias, const orc_imu_factor* factors,
                               int n_factors, const double gravity[3], const orc_ba_options* options,
                               orc_ba_summary* summary)
{
    orc_ba_options def;
    if (!options) { orc_ba_default_options(&def); options = &def; }
    if (n_cameras < 0 || n_points < 0 || n_obs < 0 || n_factors < 0) return 1;
    problem pr;
    memset(&pr, 0, sizeof pr);
    pr.C = n_cameras; pr.P = n_points; pr.M = n_obs; pr.points_constant = 0;
    pr.cam_free = cam_free; pr.obs_ptr = obs_ptr; pr.obs_cam = obs_cam; pr.obs_uv = obs_uv;
    pr.K = K; pr.huber_a = options->huber_delta;
    pr.n_fac = n_factors; pr.fac = factors; pr.gravity = gravity;
    double* c = (double*)malloc(sizeof(double) * 6 * (size_t)(n_cameras + 1));
    double* p = (double*)malloc(sizeof(double) * 3 * (size_t)(n_points + 1));
    double* v = (double*)malloc(sizeof(double) * 3 * (size_t)(n_cameras + 1));
    double* b = (double*)malloc(sizeof(double) * 6 * (size_t)(n_cameras + 1));
    memcpy(c, cameras, sizeof(double) * 6 * (size_t)n_cameras);
    memcpy(p, points, sizeof(double) * 3 * (size_t)n_points);
    memcpy(v, velocity, sizeof(double) * 3 * (size_t)n_cameras);
    memcpy(b, bias, sizeof(double) * 6 * (size_t)n_cameras);
    lm_solve(&pr, c, p, v, b, options, summary);
    if (summary->usable) {
        for (int i = 0; i < n_cameras; i++)
            if (cam_free[i]) {
                memcpy(cameras + 6 * i, c + 6 * i, 6 * sizeof(double));
                memcpy(velocity + 3 * i, v + 3 * i, 3 * sizeof(double));
                memcpy(bias + 6 * i, b + 6 * i, 6 * sizeof(double));
            }
        memcpy(points, p, sizeof(double) * 3 * (size_t)n_points);
    }
    free(c); free(p); free(v); free(b);
    return 0;
}

/* refine_pose with an InertialConstraint (src/Optimization.cpp:231-267): kind 1 = RotationPrior (predicted rotation
 * row-major, sigma), kind 2 = InertialDelta (the previous frame's pose / velocity / bias constant, this frame's
 * velocity a free block written back on success).  kind 0 = orc_refine_pose. */
int orc_refine_pose_inertial(double camera[6], const double* points, const float* uv, int n, const float K[4], int kind,
                             const double predicted[9], double sigma, const double prev_pose[6],
                             const double prev_velocity[3], const double prev_bias[6], const orc_imu_factor* delta,
                             const double gravity[3], double velocity[3], const orc_ba_options* options,
                             orc_ba_summary* summary)
{
    orc_ba_options def;
    if (!options) { orc_ba_default_options(&def); options = &def; }
    memset(summary, 0, sizeof *summary);
    if (n <= 0) return 0;                       /* src/Optimization.cpp:227-229 */
    problem pr;
    memset(&pr, 0, sizeof pr);
    pr.C = 1; pr.P = n; pr.M = n; pr.points_constant = 1;
    uint8_t fr = 1;
    int32_t* ptr = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1));
    int32_t* cam = (int32_t*)calloc((size_t)n, sizeof(int32_t));
    for (int i = 0; i <= n; i++) ptr[i] = i;
    pr.cam_free = &fr; pr.obs_ptr = ptr; pr.obs_cam = cam; pr.obs_uv = uv; pr.K = K;
    pr.huber_a = options->huber_delta;
    if (kind == 1 && sigma > 0.0) { pr.prior_R = predicted; pr.prior_sigma = sigma; }              /* RotationPrior::enabled */
    if (kind == 2 && delta && delta->duration > 0.0) {                                             /* InertialDelta::enabled */
        pr.delta_only = 1; pr.fac = delta; pr.gravity = gravity;
        pr.prev_pose = prev_pose; pr.prev_vel = prev_velocity; pr.prev_bias = prev_bias;
    }
    double c[6], v[3] = {0, 0, 0}, b[6] = {0, 0, 0, 0, 0, 0};
    memcpy(c, camera, sizeof c);
    if (pr.delta_only) memcpy(v, velocity, sizeof v);
    double* p = (double*)malloc(sizeof(double) * 3 * (size_t)n);
    memcpy(p, points, sizeof(double) * 3 * (size_t)n);
    lm_solve(&pr, c, p, pr.delta_only ? v : NULL, pr.delta_only ? b : NULL, options, summary);
    if (summary->usable) {
        memcpy(camera, c, sizeof c);
        if (pr.delta_only) memcpy(velocity, v, sizeof v);
    }
    free(ptr); free(cam); free(p);
    return 0;
}

int orc_ba_linearize(int n_cameras, int n_points, const double* cameras, const double* points,
                     const int32_t* obs_ptr, const int32_t* obs_cam, const float* obs_uv,
                     const float K[4], double huber_delta, double* U, double* gc,
                     double* V, double* gp, double* cost)
{
    memset(U, 0, sizeof(double) * 36 * (size_t)n_cameras);
    memset(gc, 0, sizeof(double) * 6 * (size_t)n_cameras);
    memset(V, 0, sizeof(double) * 9 * (size_t)n_points);
    memset(gp, 0, sizeof(double) * 3 * (size_t)n_points);
    double total = 0.0;
    for (int p = 0; p < n_points; p++) {
        for (int o = obs_ptr[p]; o < obs_ptr[p + 1]; o++) {
            double r[2], jc[12], jp[6], rho[3];
            const int c = obs_cam[o];
            orc_reprojection(cameras + 6 * c, points + 3 * p, obs_uv + 2 * o, K, r, jc, jp);
            huber(huber_delta, r[0] * r[0] + r[1] * r[1], rho);
            total += 0.5 * rho[0];
            const double w = rho[1];
            for (int a = 0; a < 6; a++) {
                for (int b = 0; b < 6; b++) U[36 * c + a * 6 + b] += w * (jc[a] * jc[b] + jc[6 + a] * jc[6 + b]);
                gc[6 * c + a] += w * (jc[a] * r[0] + jc[6 + a] * r[1]);
            }
            for (int a = 0; a < 3; a++) {
                for (int b = 0; b < 3; b++) V[9 * p + a * 3 + b] += w * (jp[a] * jp[b] + jp[3 + a] * jp[3 + b]);
                gp[3 * p + a] += w * (jp[a] * r[0] + jp[3 + a] * r[1]);
            }
        }
    }
    *cost = total;
    return 0;
}
