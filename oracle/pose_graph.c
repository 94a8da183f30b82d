/*
 * oracle/pose_graph.c — CPU restatement of optimization::pose_graph (reference src/Optimization.cpp:376-639): pose-graph
 * optimisation over the key frames after a loop closure, SE(3) (6 unknowns per key frame: angle-axis + centre) or
 * 4-DoF (yaw about "up" + centre, when gravity is known).  TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (rs_oracle.h).
 *
 *   edges        one between consecutive key frames (measured relative pose = pose_i * pose_{i+1}^-1, sigmas 0.02 / 0.2,
 *                no loss, :586-588) and one per loop constraint (sigmas 0.05 / 0.5, HuberLoss(1.0), :589-594)
 *   residual     RelativePoseError :383-426 / RelativePose4DoFError :429-492: log(R_meas^T R_from R_to^T) / sigma_rot,
 *                (R_from (c_to - c_from) - t_meas) / sigma_trans
 *   solve        first key frame constant (:596-600), 20 iterations, SPARSE_NORMAL_CHOLESKY, otherwise Ceres defaults
 *                (the trust-region loop of oracle/ba.c, here on the dense normal equations), usable rule :610-616
 *   write-back   apply_corrected_pose :499-510 (f32), per key frame
 * Third-party semantics restated (Ceres 2.x): forward-mode jets (12 wide), rotation.h conversions (jetn.h), HuberLoss +
 * Corrector (rho'' <= 0: residual and Jacobian scaled by sqrt(rho')), TrustRegionMinimizer / LevenbergMarquardtStrategy.
 * Specified here: Eigen's 4x4 float inverse in pose_relative (:494-497) is restated as adjugate / determinant (inv4f).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "rs_oracle.h"

#define NJ 12
#include "jetn.h"

void orc__trace_push(double cost, double cand, double mcc, double radius, double step_norm, double x_norm, int outcome);   /* ba.c */

#define SEQ_SIGMA_ROT 0.02
#define SEQ_SIGMA_TRANS 0.2
#define LOOP_SIGMA_ROT 0.05
#define LOOP_SIGMA_TRANS 0.5

/* General inverse of a row-major 4x4 float matrix.  Eigen's fixed-size Matrix4f::inverse() is cofactor based (its
 * operation order is an implementation detail upstream); SPECIFIED HERE as adjugate / determinant in f32 with
 *   minor(r, k)  = the 3x3 determinant of the rows != r and columns != k in ascending order, expanded along its first
 *                  row:  (a (e i - f h) - b (d i - f g)) + c (d h - e g)
 *   adj[k][r]    = (-1)^(r+k) minor(r, k),   det = ((m00 adj00 + m01 adj10) + m02 adj20) + m03 adj30,   inv = adj / det. */
static float minor3(const float* m, int r, int k)
{
    int rows[3], cols[3], nr = 0, nc = 0;
    for (int i = 0; i < 4; i++) {
        if (i != r) rows[nr++] = i;
        if (i != k) cols[nc++] = i;
    }
#define E(i, j) m[4 * rows[i] + cols[j]]
    const float t0 = E(0, 0) * (E(1, 1) * E(2, 2) - E(1, 2) * E(2, 1));
    const float t1 = E(0, 1) * (E(1, 0) * E(2, 2) - E(1, 2) * E(2, 0));
    const float t2 = E(0, 2) * (E(1, 0) * E(2, 1) - E(1, 1) * E(2, 0));
#undef E
    return (t0 - t1) + t2;
}
static void inv4f(const float* m, float* out)
{
    float adj[16];
    for (int r = 0; r < 4; r++)
        for (int k = 0; k < 4; k++) {
            const float mn = minor3(m, r, k);
            adj[4 * k + r] = ((r + k) % 2) ? -mn : mn;
        }
    const float det = ((m[0] * adj[0] + m[1] * adj[4]) + m[2] * adj[8]) + m[3] * adj[12];
    for (int i = 0; i < 16; i++) out[i] = adj[i] / det;
}

/* pose_relative, :494-497: from.pose (as double) * inverse(to.pose) (inverse in f32, then widened) */
void orc_pose_relative(const float from[16], const float to[16], double rel[16])
{
    float ti[16];
    inv4f(to, ti);
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            double s = 0.0;
            for (int k = 0; k < 4; k++) s += (double)from[4 * r + k] * (double)ti[4 * k + c];
            rel[4 * r + c] = s;
        }
}

typedef struct {
    int from, to, loop;
    double Rm[9], tm[3];        /* measured relative rotation (row-major) / translation */
} pg_edge;

typedef struct {
    int n, ne, four_dof, bs;    /* key frames, edges, mode, block size (6 or 4) */
    const pg_edge* e;
    const double* R0;           /* [n][9] row-major initial rotations (4-DoF) */
    double up[3];
} pg_problem;

/* residual r[6] (+ J[6][12] over (block_from | block_to), unused slots zero) of edge k at x */
static void edge_eval(const pg_problem* pr, int k, const double* x, double r[6], double* J)
{
    const pg_edge* e = pr->e + k;
    const int bs = pr->bs;
    const double* xf = x + (size_t)bs * e->from;
    const double* xt = x + (size_t)bs * e->to;
    jet Rf[9], Rt[9], cf[3], ct[3];
    if (!pr->four_dof) {
        jet af[3], at[3];
        for (int q = 0; q < 3; q++) { af[q] = jv(xf[q], q); at[q] = jv(xt[q], 6 + q); cf[q] = jv(xf[3 + q], 3 + q); ct[q] = jv(xt[3 + q], 9 + q); }
        aa_to_matrix(af, Rf);
        aa_to_matrix(at, Rt);
    } else {
        /* rotation_cw(yaw, R0) = R0 * AngleAxisToRotationMatrix(-up * yaw), :446-452 */
        for (int side = 0; side < 2; side++) {
            const double* xs = side ? xt : xf;
            const jet yaw = jv(xs[0], side ? 4 : 0);
            jet aa[3], Rd[9], R0j[9];
            for (int q = 0; q < 3; q++) aa[q] = jscale(yaw, -pr->up[q]);
            aa_to_matrix(aa, Rd);
            const double* R0 = pr->R0 + 9 * (size_t)(side ? e->to : e->from);
            for (int rr = 0; rr < 3; rr++)
                for (int c = 0; c < 3; c++) RM(R0j, rr, c) = jc(R0[3 * rr + c]);
            mm(R0j, Rd, side ? Rt : Rf, 0, 0);
            for (int q = 0; q < 3; q++) { if (side) ct[q] = jv(xs[1 + q], 5 + q); else cf[q] = jv(xs[1 + q], 1 + q); }
        }
    }
    jet Re[9], Rm[9], Rerr[9], d[3], te[3], rv[3];
    mm(Rf, Rt, Re, 0, 1);                                       /* R_est = R_from R_to^T */
    for (int q = 0; q < 3; q++) d[q] = jsub(ct[q], cf[q]);
    mv(Rf, d, te);                                              /* t_est = R_from (c_to - c_from) */
    for (int rr = 0; rr < 3; rr++)
        for (int c = 0; c < 3; c++) RM(Rm, rr, c) = jc(e->Rm[3 * rr + c]);
    mm(Rm, Re, Rerr, 1, 0);                                     /* R_meas^T R_est */
    matrix_to_aa(Rerr, rv);
    const double sr = e->loop ? LOOP_SIGMA_ROT : SEQ_SIGMA_ROT, st = e->loop ? LOOP_SIGMA_TRANS : SEQ_SIGMA_TRANS;
    for (int q = 0; q < 3; q++) {
        const jet a = jscale(rv[q], 1.0 / sr), b = jscale(jsub(te[q], jc(e->tm[q])), 1.0 / st);
        r[q] = a.a; r[3 + q] = b.a;
        if (J) { memcpy(J + q * NJ, a.v, sizeof a.v); memcpy(J + (3 + q) * NJ, b.v, sizeof b.v); }
    }
}

/* robustified residuals / Jacobians of all edges; returns the cost 1/2 sum rho */
static double pg_eval(const pg_problem* pr, const double* x, double* R, double* J)
{
    double cost = 0.0;
    for (int k = 0; k < pr->ne; k++) {
        double* r = R + 6 * k;
        double* j = J ? J + (size_t)6 * NJ * k : NULL;
        edge_eval(pr, k, x, r, j);
        double s = 0.0;
        for (int a = 0; a < 6; a++) s += r[a] * r[a];
        double rho = s, rho1 = 1.0;
        if (pr->e[k].loop && s > 1.0) { const double q = sqrt(s); rho = 2.0 * q - 1.0; rho1 = 1.0 / q; }   /* HuberLoss(1.0) */
        cost += 0.5 * rho;
        const double sc = sqrt(rho1);
        for (int a = 0; a < 6; a++) r[a] *= sc;
        if (j) for (int a = 0; a < 6 * NJ; a++) j[a] *= sc;
    }
    return cost;
}

static int chol_solve_dense(double* A, int n, double* b)
{
    for (int j = 0; j < n; j++) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; k++) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
        if (!(d > 0.0) || !isfinite(d)) return 1;
        d = sqrt(d);
        A[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[(size_t)i * n + j];
            for (int k = 0; k < j; k++) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
            A[(size_t)i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= A[(size_t)i * n + k] * b[k]; b[i] = s / A[(size_t)i * n + i]; }
    for (int i = n - 1; i >= 0; i--) { double s = b[i]; for (int k = i + 1; k < n; k++) s -= A[(size_t)k * n + i] * b[k]; b[i] = s / A[(size_t)i * n + i]; }
    return 0;
}

void orc_pose_graph_edge(int four_dof, const double* x_from, const double* x_to, const double R0_from[9], const double R0_to[9],
                         const double up[3], const double relative[16], int loop, double r[6], double J[72])
{
    pg_problem pr;
    pg_edge e;
    double x[12], R0[18];
    memset(&pr, 0, sizeof pr);
    memset(R0, 0, sizeof R0);
    const int bs = four_dof ? 4 : 6;
    pr.n = 2; pr.ne = 1; pr.four_dof = four_dof; pr.bs = bs; pr.e = &e; pr.R0 = R0;
    if (four_dof) { memcpy(R0, R0_from, sizeof(double) * 9); memcpy(R0 + 9, R0_to, sizeof(double) * 9); memcpy(pr.up, up, sizeof pr.up); }
    memcpy(x, x_from, sizeof(double) * bs);
    memcpy(x + bs, x_to, sizeof(double) * bs);
    e.from = 0; e.to = 1; e.loop = loop;
    for (int rr = 0; rr < 3; rr++) { for (int c = 0; c < 3; c++) e.Rm[3 * rr + c] = relative[4 * rr + c]; e.tm[rr] = relative[4 * rr + 3]; }
    edge_eval(&pr, 0, x, r, J);
}

int orc_pose_graph(int n, const float* poses, const orc_pg_edge* loops, int n_loops, int four_dof, const double gravity[3],
                   const orc_ba_options* options, float* out_poses, orc_ba_summary* sum)
{
    orc_ba_options def;
    if (!options) { orc_ba_default_options(&def); def.max_num_iterations = 20; options = &def; }      /* PGO_ITERATIONS, :120 */
    memset(sum, 0, sizeof *sum);
    if (out_poses) memcpy(out_poses, poses, sizeof(float) * 16 * (size_t)(n > 0 ? n : 0));
    if (n < 3 || n_loops <= 0) return 0;                                                              /* :546-548 */
    pg_problem pr;
    memset(&pr, 0, sizeof pr);
    pr.n = n;
    pr.up[2] = 1.0;
    if (four_dof) {                                                                                   /* :550-557 */
        const double g2 = gravity[0] * gravity[0] + gravity[1] * gravity[1] + gravity[2] * gravity[2];
        if (g2 < 1e-6) four_dof = 0;
        else for (int q = 0; q < 3; q++) pr.up[q] = -gravity[q] / sqrt(g2);
    }
    pr.four_dof = four_dof;
    pr.bs = four_dof ? 4 : 6;
    const int bs = pr.bs;
    double* x = (double*)calloc((size_t)bs * n, sizeof(double));
    double* R0 = (double*)malloc(sizeof(double) * 9 * (size_t)n);
    for (int i = 0; i < n; i++) {                                                                     /* :566-573 */
        double cam[6];
        orc_pack_pose(poses + 16 * (size_t)i, cam);
        for (int rr = 0; rr < 3; rr++)
            for (int c = 0; c < 3; c++) R0[9 * (size_t)i + 3 * rr + c] = (double)poses[16 * (size_t)i + 4 * rr + c];
        if (four_dof) { x[4 * i] = 0.0; for (int q = 0; q < 3; q++) x[4 * i + 1 + q] = cam[3 + q]; }
        else for (int q = 0; q < 6; q++) x[6 * i + q] = cam[q];
    }
    pr.R0 = R0;
    pg_edge* E = (pg_edge*)malloc(sizeof(pg_edge) * (size_t)(n - 1 + n_loops));
    int ne = 0;
    for (int i = 0; i + 1 < n; i++) {                                                                 /* :586-588 */
        double rel[16];
        orc_pose_relative(poses + 16 * (size_t)i, poses + 16 * (size_t)(i + 1), rel);
        E[ne].from = i; E[ne].to = i + 1; E[ne].loop = 0;
        for (int rr = 0; rr < 3; rr++) { for (int c = 0; c < 3; c++) E[ne].Rm[3 * rr + c] = rel[4 * rr + c]; E[ne].tm[rr] = rel[4 * rr + 3]; }
        ne++;
    }
    for (int l = 0; l < n_loops; l++) {                                                               /* :589-594 */
        if (loops[l].from < 0 || loops[l].to < 0 || loops[l].from >= n || loops[l].to >= n || loops[l].from == loops[l].to) continue;
        E[ne].from = loops[l].from; E[ne].to = loops[l].to; E[ne].loop = 1;
        for (int rr = 0; rr < 3; rr++) { for (int c = 0; c < 3; c++) E[ne].Rm[3 * rr + c] = loops[l].relative[4 * rr + c]; E[ne].tm[rr] = loops[l].relative[4 * rr + 3]; }
        ne++;
    }
    pr.e = E; pr.ne = ne;
    /* free columns: every key frame but the first (:596-600) */
    const int nu = bs * (n - 1);
#define COL(kf, q) ((kf) == 0 ? -1 : bs * ((kf) - 1) + (q))
    double* Rv = (double*)malloc(sizeof(double) * 6 * (size_t)ne);
    double* Jv = (double*)malloc(sizeof(double) * 6 * NJ * (size_t)ne);
    double* Rc = (double*)malloc(sizeof(double) * 6 * (size_t)ne);
    double* H = (double*)malloc(sizeof(double) * (size_t)nu * nu);
    double* g = (double*)malloc(sizeof(double) * (size_t)nu);
    double* scale = (double*)malloc(sizeof(double) * (size_t)nu);
    double* diag = (double*)malloc(sizeof(double) * (size_t)nu);
    double* step = (double*)malloc(sizeof(double) * (size_t)nu);
    double* cand = (double*)malloc(sizeof(double) * (size_t)bs * n);
    double* best = (double*)malloc(sizeof(double) * (size_t)bs * n);
    memcpy(best, x, sizeof(double) * (size_t)bs * n);

    /* column of local jet slot s of edge k */
#define ECOL(k, s) ((s) < (four_dof ? 4 : 6) ? COL(E[k].from, (s)) : COL(E[k].to, (s) - (four_dof ? 4 : 6)))
    const int nslot = 2 * bs;
    double x_cost = pg_eval(&pr, x, Rv, Jv);
    sum->initial_cost = x_cost;
    double minimum_cost = x_cost, radius = options->initial_trust_region_radius, factor = 2.0;
    int invalid = 0, done = 0;
    /* gradient + Jacobi scale from the first Jacobian */
    for (int i = 0; i < nu; i++) { g[i] = 0.0; scale[i] = 0.0; }
    for (int k = 0; k < ne; k++)
        for (int a = 0; a < 6; a++)
            for (int s = 0; s < nslot; s++) {
                const int c = ECOL(k, s);
                if (c < 0) continue;
                const double j = Jv[((size_t)6 * k + a) * NJ + s];
                g[c] += j * Rv[6 * k + a]; scale[c] += j * j;
            }
    double gmax = 0.0;
    for (int i = 0; i < nu; i++) { if (fabs(g[i]) > gmax) gmax = fabs(g[i]); scale[i] = options->jacobi_scaling ? 1.0 / (1.0 + sqrt(scale[i])) : 1.0; }
    if (!isfinite(x_cost)) { sum->termination = 5; done = 1; }
    else if (gmax <= options->gradient_tolerance) { sum->termination = 3; done = 1; }
    while (!done) {
        if (sum->iterations >= options->max_num_iterations) { sum->termination = 0; break; }
        sum->iterations++;
        /* scaled normal equations  (Js^T Js + D^2) y = Js^T r */
        memset(H, 0, sizeof(double) * (size_t)nu * nu);
        for (int i = 0; i < nu; i++) g[i] = 0.0;
        for (int k = 0; k < ne; k++)
            for (int a = 0; a < 6; a++)
                for (int s = 0; s < nslot; s++) {
                    const int c = ECOL(k, s);
                    if (c < 0) continue;
                    const double js = Jv[((size_t)6 * k + a) * NJ + s] * scale[c];
                    g[c] += js * Rv[6 * k + a];
                    for (int t = 0; t < nslot; t++) {
                        const int c2 = ECOL(k, t);
                        if (c2 >= 0) H[(size_t)c * nu + c2] += js * Jv[((size_t)6 * k + a) * NJ + t] * scale[c2];
                    }
                }
        for (int i = 0; i < nu; i++) {
            diag[i] = fmin(fmax(H[(size_t)i * nu + i], options->min_lm_diagonal), options->max_lm_diagonal) / radius;
            H[(size_t)i * nu + i] += diag[i];
            step[i] = g[i];
        }
        int failed = chol_solve_dense(H, nu, step);
        double mcc = 0.0;
        if (!failed) {
            for (int i = 0; i < nu; i++) { if (!isfinite(step[i])) failed = 1; step[i] = -step[i]; }
            for (int k = 0; k < ne && !failed; k++)
                for (int a = 0; a < 6; a++) {
                    double m = 0.0;
                    for (int s = 0; s < nslot; s++) { const int c = ECOL(k, s); if (c >= 0) m += Jv[((size_t)6 * k + a) * NJ + s] * scale[c] * step[c]; }
                    mcc -= m * (Rv[6 * k + a] + m / 2.0);
                }
        }
        if (failed || !(mcc > 0.0)) {
            orc__trace_push(x_cost, 0.0, failed ? 0.0 : mcc, radius, 0.0, 0.0, -1);
            if (++invalid >= options->max_num_consecutive_invalid_steps) { sum->termination = 5; break; }
            radius /= factor; factor *= 2.0;
            continue;
        }
        invalid = 0;
        memcpy(cand, x, sizeof(double) * (size_t)bs * n);
        double ssq = 0.0, xsq = 0.0;
        for (int kf = 1; kf < n; kf++)
            for (int q = 0; q < bs; q++) {
                const int c = COL(kf, q);
                cand[bs * kf + q] = x[bs * kf + q] + step[c] * scale[c];
                const double df = x[bs * kf + q] - cand[bs * kf + q];
                ssq += df * df; xsq += x[bs * kf + q] * x[bs * kf + q];
            }
        const double cand_cost = pg_eval(&pr, cand, Rc, NULL);
        if (sqrt(ssq) <= options->parameter_tolerance * (sqrt(xsq) + options->parameter_tolerance)) {
            orc__trace_push(x_cost, cand_cost, mcc, radius, sqrt(ssq), sqrt(xsq), 2);
            sum->termination = 2; break;
        }
        if (fabs(x_cost - cand_cost) <= options->function_tolerance * x_cost) {
            orc__trace_push(x_cost, cand_cost, mcc, radius, sqrt(ssq), sqrt(xsq), 2);
            sum->termination = 1; break;
        }
        const double rel = (x_cost - cand_cost) / mcc;
        orc__trace_push(x_cost, cand_cost, mcc, radius, sqrt(ssq), sqrt(xsq), (rel > options->min_relative_decrease && isfinite(cand_cost)) ? 1 : 0);
        if (rel > options->min_relative_decrease && isfinite(cand_cost)) {
            memcpy(x, cand, sizeof(double) * (size_t)bs * n);
            x_cost = pg_eval(&pr, x, Rv, Jv);
            sum->successful_steps++;
            radius = fmin(options->max_trust_region_radius, radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rel - 1.0, 3)));
            factor = 2.0;
            if (x_cost < minimum_cost) { minimum_cost = x_cost; memcpy(best, x, sizeof(double) * (size_t)bs * n); }
            gmax = 0.0;
            for (int i = 0; i < nu; i++) g[i] = 0.0;
            for (int k = 0; k < ne; k++)
                for (int a = 0; a < 6; a++)
                    for (int s = 0; s < nslot; s++) { const int c = ECOL(k, s); if (c >= 0) g[c] += Jv[((size_t)6 * k + a) * NJ + s] * Rv[6 * k + a]; }
            for (int i = 0; i < nu; i++) if (fabs(g[i]) > gmax) gmax = fabs(g[i]);
            if (gmax <= options->gradient_tolerance) { sum->termination = 3; break; }
        } else {
            radius /= factor; factor *= 2.0;
            if (radius < options->min_trust_region_radius) { sum->termination = 4; break; }
        }
    }
    sum->final_cost = minimum_cost;
    sum->final_radius = radius;
    sum->usable = (sum->termination != 5) && isfinite(minimum_cost) && minimum_cost <= sum->initial_cost;    /* :610-616 */
    if (sum->usable && out_poses)
        for (int i = 0; i < n; i++) {                                                                        /* :618-632 */
            float Rf32[9], cf32[3];
            if (four_dof) {
                /* R0 * AngleAxisd(-yaw, up).toRotationMatrix() in double, then apply_corrected_pose casts to float */
                const double yaw = -best[4 * i], c = cos(yaw), s = sin(yaw), t = 1.0 - c;
                const double u0 = pr.up[0], u1 = pr.up[1], u2 = pr.up[2];
                const double A[9] = {c + t * u0 * u0, t * u0 * u1 - s * u2, t * u0 * u2 + s * u1,
                                     t * u0 * u1 + s * u2, c + t * u1 * u1, t * u1 * u2 - s * u0,
                                     t * u0 * u2 - s * u1, t * u1 * u2 + s * u0, c + t * u2 * u2};
                for (int rr = 0; rr < 3; rr++)
                    for (int cc = 0; cc < 3; cc++) {
                        double v = 0.0;
                        for (int k = 0; k < 3; k++) v += R0[9 * (size_t)i + 3 * rr + k] * A[3 * k + cc];
                        Rf32[3 * rr + cc] = (float)v;
                    }
                for (int q = 0; q < 3; q++) cf32[q] = (float)best[4 * i + 1 + q];
            } else {
                /* rodrigues_to_matrix(Vector3f(...)).cast<double>() then cast back to float: the f32 conversion of unpack_pose */
                double cam[6];
                float P[16];
                for (int q = 0; q < 6; q++) cam[q] = best[6 * i + q];
                orc_unpack_pose(cam, P);
                for (int rr = 0; rr < 3; rr++) for (int cc = 0; cc < 3; cc++) Rf32[3 * rr + cc] = P[4 * rr + cc];
                for (int q = 0; q < 3; q++) cf32[q] = (float)best[6 * i + 3 + q];
            }
            float* P = out_poses + 16 * (size_t)i;                                                            /* :499-504 */
            for (int rr = 0; rr < 3; rr++) {
                for (int cc = 0; cc < 3; cc++) P[4 * rr + cc] = Rf32[3 * rr + cc];
                P[4 * rr + 3] = (-Rf32[3 * rr] * cf32[0] + -Rf32[3 * rr + 1] * cf32[1]) + -Rf32[3 * rr + 2] * cf32[2];
            }
            P[12] = 0.f; P[13] = 0.f; P[14] = 0.f; P[15] = 1.f;
        }
    free(x); free(R0); free(E); free(Rv); free(Jv); free(Rc); free(H); free(g); free(scale); free(diag); free(step); free(cand); free(best);
    return 0;
}
