// pose_graph.cpp — optimization::pose_graph of the reference (src/Optimization.cpp:376-639) as a HOST entry point of
// librsgpu.so.
//
// Why the host: the problem is one 6-residual edge per consecutive key-frame pair plus a handful of loop edges — a
// few hundred residual blocks, solved once per loop closure (src/Slam.cpp:258-268) — and its normal equations are a
// sparse, nearly banded matrix whose factorisation is a short dependent chain.  The reference solves it with Ceres'
// SPARSE_NORMAL_CHOLESKY on the CPU in the same place.  What IS data-parallel about a loop closure — moving every map
// point with its owning key frame (transform_points, :512-536) — runs on the GPU (K14, tracks.hip), and
// rs_map_pose_graph (map.hip) chains the two on the resident map.
//
// Formulation (written for this library, not taken from the oracle):
//   * residual blocks RelativePoseError (:383-426) / RelativePose4DoFError (:429-492) evaluated on forward-mode dual
//     numbers (12 resp. 8 partials) — what ceres::AutoDiffCostFunction does; HuberLoss(1.0) on the loop edges through
//     Ceres' corrector (rho'' <= 0: residual and Jacobian scaled by sqrt(rho'));
//   * normal equations in a block ENVELOPE (skyline) store after a reverse Cuthill-McKee ordering of the key-frame
//     graph: with lap-closing loops the natural order has bandwidth = one lap, RCM interleaves the laps and brings
//     it down to a few blocks, so a factorisation costs O(n * band^2) instead of O(n^3);
//   * the trust-region loop of Ceres 2.x with the reference's settings (PGO_ITERATIONS = 20, :120; otherwise
//     defaults): Jacobi scaling from the first Jacobian, D^2 = clamp(diag, 1e-6, 1e32) / radius, model cost change
//     -(J s).(r + J s / 2), parameter / function tolerance before the acceptance test, radius update rules, the
//     "usable" rule of :610-616; per-iteration record as for rs_bundle_adjust.
//   * write-back exactly as apply_corrected_pose (:499-510) does it, in f32.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <queue>
#include <vector>

#include "../../include/rsgpu.h"

namespace {

constexpr double kEps = 2.220446049250313e-16;
constexpr double kSeqSigmaRot = 0.02, kSeqSigmaTrans = 0.2, kLoopSigmaRot = 0.05, kLoopSigmaTrans = 0.5;   // :378-381

// ---------------------------------------------------------------------------------------------- dual numbers
template <int N>
struct Dual {
    double a;
    double v[N];
    Dual() {}
    explicit Dual(double x) : a(x) { for (int i = 0; i < N; i++) v[i] = 0.0; }
    Dual(double x, int k) : a(x) { for (int i = 0; i < N; i++) v[i] = 0.0; v[k] = 1.0; }
};
template <int N> inline Dual<N> operator+(const Dual<N>& f, const Dual<N>& g) { Dual<N> r; r.a = f.a + g.a; for (int i = 0; i < N; i++) r.v[i] = f.v[i] + g.v[i]; return r; }
template <int N> inline Dual<N> operator-(const Dual<N>& f, const Dual<N>& g) { Dual<N> r; r.a = f.a - g.a; for (int i = 0; i < N; i++) r.v[i] = f.v[i] - g.v[i]; return r; }
template <int N> inline Dual<N> operator-(const Dual<N>& f) { Dual<N> r; r.a = -f.a; for (int i = 0; i < N; i++) r.v[i] = -f.v[i]; return r; }
template <int N> inline Dual<N> operator*(const Dual<N>& f, const Dual<N>& g) { Dual<N> r; r.a = f.a * g.a; for (int i = 0; i < N; i++) r.v[i] = f.a * g.v[i] + f.v[i] * g.a; return r; }
template <int N> inline Dual<N> operator*(const Dual<N>& f, double s) { Dual<N> r; r.a = f.a * s; for (int i = 0; i < N; i++) r.v[i] = f.v[i] * s; return r; }
template <int N> inline Dual<N> operator/(const Dual<N>& f, const Dual<N>& g)
{
    Dual<N> r; const double gi = 1.0 / g.a, fg = f.a * gi;
    r.a = fg; for (int i = 0; i < N; i++) r.v[i] = (f.v[i] - fg * g.v[i]) * gi; return r;
}
template <int N> inline Dual<N> dsqrt(const Dual<N>& f) { Dual<N> r; const double t = sqrt(f.a), h = 1.0 / (2.0 * t); r.a = t; for (int i = 0; i < N; i++) r.v[i] = f.v[i] * h; return r; }
template <int N> inline Dual<N> dcos(const Dual<N>& f) { Dual<N> r; const double s = -sin(f.a); r.a = cos(f.a); for (int i = 0; i < N; i++) r.v[i] = s * f.v[i]; return r; }
template <int N> inline Dual<N> dsin(const Dual<N>& f) { Dual<N> r; const double c = cos(f.a); r.a = sin(f.a); for (int i = 0; i < N; i++) r.v[i] = c * f.v[i]; return r; }
template <int N> inline Dual<N> datan2(const Dual<N>& g, const Dual<N>& f)
{
    Dual<N> r; const double t = 1.0 / (f.a * f.a + g.a * g.a);
    r.a = atan2(g.a, f.a); for (int i = 0; i < N; i++) r.v[i] = t * (f.a * g.v[i] - g.a * f.v[i]); return r;
}

// 3x3 matrices of duals, ROW-major here (M[3 r + c])
template <int N> using Mat = Dual<N>[9];

// ceres::AngleAxisToRotationMatrix
template <int N>
void rotation_of(const Dual<N> aa[3], Dual<N> R[9])
{
    const Dual<N> th2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
    const Dual<N> one(1.0);
    if (th2.a > kEps) {
        const Dual<N> th = dsqrt(th2);
        const Dual<N> wx = aa[0] / th, wy = aa[1] / th, wz = aa[2] / th;
        const Dual<N> ct = dcos(th), st = dsin(th), omc = one - ct;
        R[0] = ct + wx * wx * omc;          R[1] = wx * wy * omc - wz * st;     R[2] = wy * st + wx * wz * omc;
        R[3] = wz * st + wx * wy * omc;     R[4] = ct + wy * wy * omc;          R[5] = -(wx * st) + wy * wz * omc;
        R[6] = -(wy * st) + wx * wz * omc;  R[7] = wx * st + wy * wz * omc;     R[8] = ct + wz * wz * omc;
    } else {                                 // first-order branch: I + [aa]x
        R[0] = one;    R[1] = -aa[2]; R[2] = aa[1];
        R[3] = aa[2];  R[4] = one;    R[5] = -aa[0];
        R[6] = -aa[1]; R[7] = aa[0];  R[8] = one;
    }
}

// ceres::RotationMatrixToAngleAxis = RotationMatrixToQuaternion + QuaternionToAngleAxis
template <int N>
void log_of(const Dual<N> R[9], Dual<N> aa[3])
{
    auto M = [&](int r, int c) -> const Dual<N>& { return R[3 * r + c]; };
    Dual<N> q[4];
    const Dual<N> trace = M(0, 0) + M(1, 1) + M(2, 2);
    if (trace.a >= 0.0) {
        Dual<N> t = dsqrt(trace + Dual<N>(1.0));
        q[0] = t * 0.5;
        t = Dual<N>(0.5) / t;
        q[1] = (M(2, 1) - M(1, 2)) * t;
        q[2] = (M(0, 2) - M(2, 0)) * t;
        q[3] = (M(1, 0) - M(0, 1)) * t;
    } else {
        int i = 0;
        if (M(1, 1).a > M(0, 0).a) i = 1;
        if (M(2, 2).a > M(i, i).a) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        Dual<N> t = dsqrt(M(i, i) - M(j, j) - M(k, k) + Dual<N>(1.0));
        q[i + 1] = t * 0.5;
        t = Dual<N>(0.5) / t;
        q[0] = (M(k, j) - M(j, k)) * t;
        q[j + 1] = (M(j, i) + M(i, j)) * t;
        q[k + 1] = (M(k, i) + M(i, k)) * t;
    }
    const Dual<N> s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    if (s2.a > 0.0) {
        const Dual<N> s = dsqrt(s2);
        const Dual<N> two_theta = ((q[0].a < 0.0) ? datan2(-s, -q[0]) : datan2(s, q[0])) * 2.0;
        const Dual<N> k = two_theta / s;
        for (int a = 0; a < 3; a++) aa[a] = q[a + 1] * k;
    } else {
        for (int a = 0; a < 3; a++) aa[a] = q[a + 1] * 2.0;
    }
}

// C = op(A) op(B)
template <int N>
void mul(const Dual<N>* A, bool ta, const Dual<N>* B, bool tb, Dual<N>* C)
{
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            Dual<N> s = (ta ? A[r] : A[3 * r]) * (tb ? B[3 * c] : B[c]);
            for (int k = 1; k < 3; k++) s = s + (ta ? A[3 * k + r] : A[3 * r + k]) * (tb ? B[3 * c + k] : B[3 * k + c]);
            C[3 * r + c] = s;
        }
}

struct Edge {
    int from, to;
    bool loop;
    double Rm[9], tm[3];       // measured relative rotation (row-major) / translation
};

struct Graph {
    int n = 0, bs = 6;
    bool four_dof = false;
    double up[3] = {0.0, 0.0, 1.0};
    std::vector<Edge> edges;
    std::vector<double> R0;    // [n][9] initial rotations (4-DoF parametrisation, :446-452)
};

// residual [6] and d residual / d (block_from | block_to) [6][2 bs] of one edge
template <int BS>
void edge_residual(const Graph& g, const Edge& e, const double* xf, const double* xt, double r[6], double* J)
{
    constexpr int N = 2 * BS;
    using D = Dual<N>;
    D Rf[9], Rt[9], cf[3], ct[3];
    if (BS == 6) {
        D af[3], at[3];
        for (int q = 0; q < 3; q++) { af[q] = D(xf[q], q); at[q] = D(xt[q], 6 + q); cf[q] = D(xf[3 + q], 3 + q); ct[q] = D(xt[3 + q], 9 + q); }
        rotation_of<N>(af, Rf);
        rotation_of<N>(at, Rt);
    } else {
        for (int side = 0; side < 2; side++) {                     // rotation_cw(yaw, R0) = R0 * R(-up * yaw), :446-452
            const double* xs = side ? xt : xf;
            const D yaw(xs[0], side ? 4 : 0);
            D aa[3], Rd[9], R0d[9];
            for (int q = 0; q < 3; q++) aa[q] = yaw * (-g.up[q]);
            rotation_of<N>(aa, Rd);
            const double* R0 = &g.R0[9 * (size_t)(side ? e.to : e.from)];
            for (int q = 0; q < 9; q++) R0d[q] = D(R0[q]);
            mul<N>(R0d, false, Rd, false, side ? Rt : Rf);
            for (int q = 0; q < 3; q++) (side ? ct : cf)[q] = D(xs[1 + q], (side ? 5 : 1) + q);
        }
    }
    D Rest[9], Rmeas[9], Rerr[9], d[3], rv[3];
    mul<N>(Rf, false, Rt, true, Rest);                              // R_est = R_from R_to^T, :408 / :465
    for (int q = 0; q < 3; q++) d[q] = ct[q] - cf[q];
    for (int q = 0; q < 9; q++) Rmeas[q] = D(e.Rm[q]);
    mul<N>(Rmeas, true, Rest, false, Rerr);                         // R_meas^T R_est, :410
    log_of<N>(Rerr, rv);
    const double sr = e.loop ? kLoopSigmaRot : kSeqSigmaRot, st = e.loop ? kLoopSigmaTrans : kSeqSigmaTrans;
    for (int q = 0; q < 3; q++) {
        const D te = Rf[3 * q] * d[0] + Rf[3 * q + 1] * d[1] + Rf[3 * q + 2] * d[2];   // t_est = R_from (c_to - c_from), :409
        const D a = rv[q] * (1.0 / sr), b = (te - D(e.tm[q])) * (1.0 / st);
        r[q] = a.a;
        r[3 + q] = b.a;
        if (J) { memcpy(J + (size_t)q * N, a.v, sizeof a.v); memcpy(J + (size_t)(3 + q) * N, b.v, sizeof b.v); }
    }
}

// all edges at x: corrected residuals R [ne][6], corrected Jacobians J [ne][6][2 bs] (or nullptr); returns the cost
double evaluate(const Graph& g, const std::vector<double>& x, std::vector<double>& R, std::vector<double>* J)
{
    const int bs = g.bs, w = 2 * bs;
    double cost = 0.0;
    for (size_t k = 0; k < g.edges.size(); k++) {
        const Edge& e = g.edges[k];
        double* r = &R[6 * k];
        double* j = J ? &(*J)[(size_t)6 * w * k] : nullptr;
        if (bs == 6) edge_residual<6>(g, e, &x[6 * (size_t)e.from], &x[6 * (size_t)e.to], r, j);
        else edge_residual<4>(g, e, &x[4 * (size_t)e.from], &x[4 * (size_t)e.to], r, j);
        double s = 0.0;
        for (int a = 0; a < 6; a++) s += r[a] * r[a];
        double rho = s, rho1 = 1.0;
        if (e.loop && s > 1.0) { const double q = sqrt(s); rho = 2.0 * q - 1.0; rho1 = 1.0 / q; }   // ceres::HuberLoss(1.0), :583
        cost += 0.5 * rho;
        const double sc = sqrt(rho1);
        if (sc != 1.0) {
            for (int a = 0; a < 6; a++) r[a] *= sc;
            if (j) for (int a = 0; a < 6 * w; a++) j[a] *= sc;
        }
    }
    return cost;
}

// ------------------------------------------------------------------------------------ ordering + envelope store
// reverse Cuthill-McKee on the graph of the free key frames (blocks 0 .. nb-1)
std::vector<int> reverse_cuthill_mckee(int nb, const std::vector<std::vector<int>>& adj)
{
    std::vector<int> order, level(nb), seen(nb, 0);
    order.reserve((size_t)nb);
    auto bfs = [&](int root, std::vector<int>* out) {          // returns the last vertex of the deepest level with least degree
        std::fill(level.begin(), level.end(), -1);
        std::queue<int> q;
        q.push(root);
        level[root] = 0;
        int far = root;
        while (!q.empty()) {
            const int v = q.front();
            q.pop();
            if (out) out->push_back(v);
            if (level[v] > level[far] || (level[v] == level[far] && adj[v].size() < adj[far].size())) far = v;
            std::vector<int> nb_sorted;
            for (int u : adj[v]) if (level[u] < 0 && !seen[u]) { level[u] = level[v] + 1; nb_sorted.push_back(u); }
            std::sort(nb_sorted.begin(), nb_sorted.end(), [&](int a, int b) { return adj[a].size() != adj[b].size() ? adj[a].size() < adj[b].size() : a < b; });
            for (int u : nb_sorted) q.push(u);
        }
        return far;
    };
    for (int start = 0; start < nb; start++) {
        if (seen[start]) continue;
        int root = start;
        for (int pass = 0; pass < 4; pass++) {                  // pseudo-peripheral vertex: walk to the far end a few times
            const int far = bfs(root, nullptr);
            if (far == root) break;
            root = far;
        }
        std::vector<int> comp;
        bfs(root, &comp);
        for (int v : comp) seen[v] = 1;
        order.insert(order.end(), comp.begin(), comp.end());
    }
    std::reverse(order.begin(), order.end());
    return order;                                               // order[new] = old
}

struct Envelope {
    int N = 0;
    std::vector<int> first;        // first stored column of scalar row i
    std::vector<size_t> ptr;       // row i occupies val[ptr[i] .. ptr[i] + (i - first[i])], diagonal last
    std::vector<double> val;
    double& at(int i, int j) { return val[ptr[i] + (size_t)(j - first[i])]; }
    void clear() { std::fill(val.begin(), val.end(), 0.0); }
    // in-place A = L L^T restricted to the envelope (no fill outside it); false when not positive definite
    bool factor()
    {
        for (int i = 0; i < N; i++) {
            double* Li = &val[ptr[i]] - first[i];
            for (int j = first[i]; j < i; j++) {
                const double* Lj = &val[ptr[j]] - first[j];
                double s = Li[j];
                for (int k = std::max(first[i], first[j]); k < j; k++) s -= Li[k] * Lj[k];
                Li[j] = s / Lj[j];
            }
            double d = Li[i];
            for (int k = first[i]; k < i; k++) d -= Li[k] * Li[k];
            if (!(d > 0.0) || !std::isfinite(d)) return false;
            Li[i] = sqrt(d);
        }
        return true;
    }
    void solve(double* b) const
    {
        for (int i = 0; i < N; i++) {
            const double* Li = &val[ptr[i]] - first[i];
            double s = b[i];
            for (int k = first[i]; k < i; k++) s -= Li[k] * b[k];
            b[i] = s / Li[i];
        }
        for (int i = N - 1; i >= 0; i--) {
            const double* Li = &val[ptr[i]] - first[i];
            const double xi = b[i] / Li[i];
            b[i] = xi;
            for (int k = first[i]; k < i; k++) b[k] -= Li[k] * xi;
        }
    }
};

void push_trace(rs_ba_iteration* tr, int cap, int* cnt, double cost, double cand, double mcc, double radius, double sn, double xn, int outcome)
{
    if (!cnt) return;
    if (tr && *cnt < cap) {
        rs_ba_iteration& e = tr[*cnt];
        memset(&e, 0, sizeof e);
        e.cost = cost; e.candidate_cost = cand; e.model_cost_change = mcc; e.radius = radius; e.step_norm = sn; e.x_norm = xn; e.outcome = outcome;
    }
    (*cnt)++;
}

// General 4x4 f32 inverse.  The reference calls Eigen's Matrix4f::inverse() (cofactor based; operation order unspecified
// upstream); specified for this library as adjugate / determinant with every 3x3 minor (rows / columns in ascending
// order) expanded along its first row, (a (e i - f h) - b (d i - f g)) + c (d h - e g), and
// det = ((m00 A00 + m01 A10) + m02 A20) + m03 A30.
void inverse4(const float* m, float* out)
{
    float adj[16];
    for (int r = 0; r < 4; r++)
        for (int k = 0; k < 4; k++) {
            int rr[3], cc[3], a = 0, b = 0;
            for (int i = 0; i < 4; i++) { if (i != r) rr[a++] = i; if (i != k) cc[b++] = i; }
            auto e = [&](int i, int j) { return m[4 * rr[i] + cc[j]]; };
            const float minor = (e(0, 0) * (e(1, 1) * e(2, 2) - e(1, 2) * e(2, 1)) - e(0, 1) * (e(1, 0) * e(2, 2) - e(1, 2) * e(2, 0))) +
                                e(0, 2) * (e(1, 0) * e(2, 1) - e(1, 1) * e(2, 0));
            adj[4 * k + r] = ((r + k) & 1) ? -minor : minor;        // adjugate = transposed cofactors
        }
    const float det = ((m[0] * adj[0] + m[1] * adj[4]) + m[2] * adj[8]) + m[3] * adj[12];
    for (int i = 0; i < 16; i++) out[i] = adj[i] / det;
}

}  // namespace

extern "C" void rs_pack_pose(const float T[16], double cam[6]);
extern "C" void rs_unpack_pose(const double cam[6], float T[16]);

extern "C" void rs_pose_relative(const float h_from[16], const float h_to[16], double h_relative[16])
{
    float inv[16];
    inverse4(h_to, inv);
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            double s = 0.0;
            for (int k = 0; k < 4; k++) s += (double)h_from[4 * r + k] * (double)inv[4 * k + c];
            h_relative[4 * r + c] = s;
        }
}

extern "C" int rs_pose_graph(int n_kf, const float* h_poses, const rs_pose_graph_edge* h_loops, int n_loops, int four_dof,
                             const double h_gravity[3], const rs_ba_options* options, float* h_out_poses,
                             float* h_velocity_rotation, rs_ba_summary* h_summary, rs_ba_iteration* h_trace,
                             int trace_capacity, int* h_trace_count)
{
    if (n_kf < 0 || n_loops < 0 || (n_kf > 0 && (!h_poses || !h_out_poses)) || (n_loops > 0 && !h_loops) || !h_summary ||
        (four_dof && !h_gravity) || trace_capacity < 0)
        return RS_ERR_INVALID;
    rs_ba_options def;
    if (!options) { rs_ba_default_options(&def); def.max_num_iterations = 20; options = &def; }     // PGO_ITERATIONS, :120
    memset(h_summary, 0, sizeof *h_summary);
    if (h_trace_count) *h_trace_count = 0;
    const size_t n = (size_t)n_kf;
    if (n && h_out_poses != h_poses) memcpy(h_out_poses, h_poses, sizeof(float) * 16 * n);
    if (h_velocity_rotation)
        for (size_t i = 0; i < n; i++)
            for (int q = 0; q < 9; q++) h_velocity_rotation[9 * i + q] = (q % 4 == 0) ? 1.0f : 0.0f;
    if (n_kf < 3 || n_loops == 0) return RS_OK;                                                       // :546-548

    Graph g;
    g.n = n_kf;
    if (four_dof) {                                                                                   // :550-557
        const double g2 = h_gravity[0] * h_gravity[0] + h_gravity[1] * h_gravity[1] + h_gravity[2] * h_gravity[2];
        if (g2 < 1e-6) four_dof = 0;
        else { const double gn = sqrt(g2); for (int q = 0; q < 3; q++) g.up[q] = -h_gravity[q] / gn; }
    }
    g.four_dof = four_dof != 0;
    g.bs = four_dof ? 4 : 6;
    const int bs = g.bs, w = 2 * bs;
    std::vector<double> x((size_t)bs * n);
    g.R0.resize(9 * n);
    for (size_t i = 0; i < n; i++) {                                                                  // :566-573
        double cam[6];
        rs_pack_pose(h_poses + 16 * i, cam);
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) g.R0[9 * i + 3 * r + c] = (double)h_poses[16 * i + 4 * r + c];
        if (four_dof) { x[4 * i] = 0.0; for (int q = 0; q < 3; q++) x[4 * i + 1 + q] = cam[3 + q]; }
        else for (int q = 0; q < 6; q++) x[6 * i + q] = cam[q];
    }
    auto add_edge = [&](int from, int to, const double* rel, bool loop) {                             // :575-584
        Edge e;
        e.from = from; e.to = to; e.loop = loop;
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) e.Rm[3 * r + c] = rel[4 * r + c]; e.tm[r] = rel[4 * r + 3]; }
        g.edges.push_back(e);
    };
    for (int i = 0; i + 1 < n_kf; i++) {                                                              // :586-588
        double rel[16];
        rs_pose_relative(h_poses + 16 * (size_t)i, h_poses + 16 * (size_t)(i + 1), rel);
        add_edge(i, i + 1, rel, false);
    }
    for (int l = 0; l < n_loops; l++) {                                                               // :589-594
        const rs_pose_graph_edge& L = h_loops[l];
        if (L.from < 0 || L.to < 0 || L.from >= n_kf || L.to >= n_kf || L.from == L.to) continue;
        add_edge(L.from, L.to, L.relative, true);
    }
    const size_t ne = g.edges.size();

    // unknowns: every key frame but the first (:596-600); ordering and envelope of the block graph
    const int nb = n_kf - 1;
    std::vector<std::vector<int>> adj((size_t)nb);
    for (const Edge& e : g.edges)
        if (e.from > 0 && e.to > 0) { adj[(size_t)e.from - 1].push_back(e.to - 1); adj[(size_t)e.to - 1].push_back(e.from - 1); }
    for (auto& a : adj) { std::sort(a.begin(), a.end()); a.erase(std::unique(a.begin(), a.end()), a.end()); }
    const std::vector<int> order = reverse_cuthill_mckee(nb, adj);
    std::vector<int> slot((size_t)nb);                      // slot[old block] = new block
    for (int i = 0; i < nb; i++) slot[(size_t)order[(size_t)i]] = i;
    std::vector<int> first_block((size_t)nb);
    for (int i = 0; i < nb; i++) first_block[(size_t)i] = i;
    for (int v = 0; v < nb; v++)
        for (int u : adj[(size_t)v]) {
            const int a = slot[(size_t)v], b = slot[(size_t)u];
            if (b < a) first_block[(size_t)a] = std::min(first_block[(size_t)a], b);
        }
    Envelope H;
    H.N = nb * bs;
    H.first.resize((size_t)H.N);
    H.ptr.resize((size_t)H.N + 1);
    size_t total = 0;
    for (int i = 0; i < H.N; i++) {
        H.first[(size_t)i] = first_block[(size_t)(i / bs)] * bs;
        H.ptr[(size_t)i] = total;
        total += (size_t)(i - H.first[(size_t)i] + 1);
    }
    H.ptr[(size_t)H.N] = total;
    H.val.assign(total, 0.0);
    auto col_of = [&](int kf, int q) { return kf == 0 ? -1 : slot[(size_t)kf - 1] * bs + q; };
    // column of local slot s of edge k
    std::vector<int> ecol(ne * (size_t)w);
    for (size_t k = 0; k < ne; k++)
        for (int s = 0; s < w; s++) ecol[k * w + s] = s < bs ? col_of(g.edges[k].from, s) : col_of(g.edges[k].to, s - bs);

    const int N = H.N;
    std::vector<double> R(6 * ne), J((size_t)6 * w * ne), Rc(6 * ne), grad((size_t)N), scale((size_t)N), step((size_t)N), cand(x), best(x);
    double x_cost = evaluate(g, x, R, &J);
    h_summary->initial_cost = x_cost;
    h_summary->final_cost = x_cost;
    double minimum_cost = x_cost, radius = options->initial_trust_region_radius, decrease = 2.0;
    int invalid = 0;
    auto gradient_max = [&]() {
        std::fill(grad.begin(), grad.end(), 0.0);
        for (size_t k = 0; k < ne; k++)
            for (int a = 0; a < 6; a++)
                for (int s = 0; s < w; s++) { const int c = ecol[k * w + s]; if (c >= 0) grad[(size_t)c] += J[(6 * k + a) * w + s] * R[6 * k + a]; }
        double m = 0.0;
        for (double v : grad) m = std::max(m, fabs(v));
        return m;
    };
    double gmax = gradient_max();
    std::fill(scale.begin(), scale.end(), 0.0);
    for (size_t k = 0; k < ne; k++)
        for (int a = 0; a < 6; a++)
            for (int s = 0; s < w; s++) { const int c = ecol[k * w + s]; if (c >= 0) scale[(size_t)c] += J[(6 * k + a) * w + s] * J[(6 * k + a) * w + s]; }
    for (double& v : scale) v = options->jacobi_scaling ? 1.0 / (1.0 + sqrt(v)) : 1.0;

    bool done = false;
    if (!std::isfinite(x_cost)) { h_summary->termination = RS_BA_FAILURE; done = true; }
    else if (gmax <= options->gradient_tolerance) { h_summary->termination = RS_BA_CONVERGENCE_GRADIENT; done = true; }
    while (!done) {
        if (h_summary->iterations >= options->max_num_iterations) { h_summary->termination = RS_BA_NO_CONVERGENCE; break; }
        h_summary->iterations++;
        // (Js^T Js + D^2) y = Js^T r on the scaled Jacobian
        H.clear();
        std::fill(grad.begin(), grad.end(), 0.0);
        for (size_t k = 0; k < ne; k++)
            for (int a = 0; a < 6; a++) {
                const double* jr = &J[(6 * k + a) * w];
                for (int s = 0; s < w; s++) {
                    const int c = ecol[k * w + s];
                    if (c < 0) continue;
                    const double js = jr[s] * scale[(size_t)c];
                    grad[(size_t)c] += js * R[6 * k + a];
                    for (int t = 0; t < w; t++) {
                        const int c2 = ecol[k * w + t];
                        if (c2 >= 0 && c2 <= c) H.at(c, c2) += js * jr[t] * scale[(size_t)c2];
                    }
                }
            }
        for (int i = 0; i < N; i++) {
            double& d = H.at(i, i);
            d += std::min(std::max(d, options->min_lm_diagonal), options->max_lm_diagonal) / radius;
            step[(size_t)i] = grad[(size_t)i];
        }
        bool failed = !H.factor();
        double mcc = 0.0;
        if (!failed) {
            H.solve(step.data());
            for (double& v : step) { if (!std::isfinite(v)) failed = true; v = -v; }
        }
        if (!failed)
            for (size_t k = 0; k < ne; k++)
                for (int a = 0; a < 6; a++) {
                    double m = 0.0;
                    for (int s = 0; s < w; s++) { const int c = ecol[k * w + s]; if (c >= 0) m += J[(6 * k + a) * w + s] * scale[(size_t)c] * step[(size_t)c]; }
                    mcc -= m * (R[6 * k + a] + m / 2.0);
                }
        if (failed || !(mcc > 0.0)) {                                           // TrustRegionMinimizer::HandleInvalidStep
            push_trace(h_trace, trace_capacity, h_trace_count, x_cost, 0.0, failed ? 0.0 : mcc, radius, 0.0, 0.0, -1);
            if (++invalid >= options->max_num_consecutive_invalid_steps) { h_summary->termination = RS_BA_FAILURE; break; }
            radius /= decrease;
            decrease *= 2.0;
            continue;
        }
        invalid = 0;
        cand = x;
        double ssq = 0.0, xsq = 0.0;
        for (int kf = 1; kf < n_kf; kf++)
            for (int q = 0; q < bs; q++) {
                const int c = col_of(kf, q);
                const size_t i = (size_t)bs * kf + q;
                cand[i] = x[i] + step[(size_t)c] * scale[(size_t)c];
                const double df = x[i] - cand[i];
                ssq += df * df;
                xsq += x[i] * x[i];
            }
        const double cand_cost = evaluate(g, cand, Rc, nullptr);
        const double sn = sqrt(ssq), xn = sqrt(xsq);
        if (sn <= options->parameter_tolerance * (xn + options->parameter_tolerance)) {
            push_trace(h_trace, trace_capacity, h_trace_count, x_cost, cand_cost, mcc, radius, sn, xn, 2);
            h_summary->termination = RS_BA_CONVERGENCE_PARAMETER;
            break;
        }
        if (fabs(x_cost - cand_cost) <= options->function_tolerance * x_cost) {
            push_trace(h_trace, trace_capacity, h_trace_count, x_cost, cand_cost, mcc, radius, sn, xn, 2);
            h_summary->termination = RS_BA_CONVERGENCE_FUNCTION;
            break;
        }
        const double rel = (x_cost - cand_cost) / mcc;
        const bool accept = rel > options->min_relative_decrease && std::isfinite(cand_cost);
        push_trace(h_trace, trace_capacity, h_trace_count, x_cost, cand_cost, mcc, radius, sn, xn, accept ? 1 : 0);
        if (accept) {
            x = cand;
            x_cost = evaluate(g, x, R, &J);
            h_summary->successful_steps++;
            radius = std::min(options->max_trust_region_radius, radius / std::max(1.0 / 3.0, 1.0 - pow(2.0 * rel - 1.0, 3)));
            decrease = 2.0;
            if (x_cost < minimum_cost) { minimum_cost = x_cost; best = x; }
            if (gradient_max() <= options->gradient_tolerance) { h_summary->termination = RS_BA_CONVERGENCE_GRADIENT; break; }
        } else {
            radius /= decrease;
            decrease *= 2.0;
            if (radius < options->min_trust_region_radius) { h_summary->termination = RS_BA_CONVERGENCE_RADIUS; break; }
        }
    }
    h_summary->final_cost = minimum_cost;
    h_summary->final_radius = radius;
    h_summary->usable = (h_summary->termination != RS_BA_FAILURE && std::isfinite(minimum_cost) && minimum_cost <= h_summary->initial_cost) ? 1 : 0;   // :610-616
    if (!h_summary->usable) return RS_OK;

    for (size_t i = 0; i < n; i++) {                                                                  // :618-632
        float Rn[9], cn[3];
        if (four_dof) {
            // R0 * AngleAxisd(-yaw, up).toRotationMatrix() in f64, narrowed by apply_corrected_pose
            const double ang = -best[4 * i], c = cos(ang), s = sin(ang), t = 1.0 - c;
            const double ux = g.up[0], uy = g.up[1], uz = g.up[2];
            const double A[9] = {c + t * ux * ux, t * ux * uy - s * uz, t * ux * uz + s * uy,
                                 t * ux * uy + s * uz, c + t * uy * uy, t * uy * uz - s * ux,
                                 t * ux * uz - s * uy, t * uy * uz + s * ux, c + t * uz * uz};
            for (int r = 0; r < 3; r++)
                for (int cc = 0; cc < 3; cc++) {
                    double v = 0.0;
                    for (int k = 0; k < 3; k++) v += g.R0[9 * i + 3 * r + k] * A[3 * k + cc];
                    Rn[3 * r + cc] = (float)v;
                }
            for (int q = 0; q < 3; q++) cn[q] = (float)best[4 * i + 1 + q];
        } else {
            float T[16];
            rs_unpack_pose(&best[6 * i], T);                // rodrigues_to_matrix(Vector3f(...)) as unpack_pose does it
            for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) Rn[3 * r + cc] = T[4 * r + cc];
            for (int q = 0; q < 3; q++) cn[q] = (float)best[6 * i + 3 + q];
        }
        float* P = h_out_poses + 16 * i;                                                              // :499-504
        const float* Pold = h_poses + 16 * i;
        if (h_velocity_rotation)                                                                      // R_delta = R_new^T R_old, :505
            for (int r = 0; r < 3; r++)
                for (int cc = 0; cc < 3; cc++)
                    h_velocity_rotation[9 * i + 3 * r + cc] = (Rn[r] * Pold[cc] + Rn[3 + r] * Pold[4 + cc]) + Rn[6 + r] * Pold[8 + cc];
        for (int r = 0; r < 3; r++) {
            for (int cc = 0; cc < 3; cc++) P[4 * r + cc] = Rn[3 * r + cc];
            P[4 * r + 3] = (-Rn[3 * r] * cn[0] + -Rn[3 * r + 1] * cn[1]) + -Rn[3 * r + 2] * cn[2];
        }
        P[12] = 0.f; P[13] = 0.f; P[14] = 0.f; P[15] = 1.f;
    }
    return RS_OK;
}
