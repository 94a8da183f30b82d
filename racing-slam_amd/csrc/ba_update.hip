// ba_update.hip — K8: back-substitution of the point blocks, candidate state and
// the robust cost at the candidate (the second half of one LM step of
// ceres::Solve, reference src/Optimization.cpp:360: SchurEliminator::BackSubstitute,
// candidate evaluation of TrustRegionMinimizer).
//
// delta_p = -V^-1 (g_p + sum_i W_i^T delta_c_i);  x_cand = x + delta;  cost(x_cand).
// Four lanes share a landmark (lane = landmark + 16 * sub, the sub-lanes split
// the landmark's observations and combine with two xor-shuffles), a workgroup of
// 4 waves covers 64 landmarks.  The camera blocks (rotation, left Jacobian,
// centre) of BOTH the current and the candidate state are staged in LDS; the
// candidate's blocks come from K7.  Jacobians are recomputed, never read from HBM.
#include "ba_common.h"

#define K8_THREADS 256

static __device__ __forceinline__ void ba_backsub_cost4_body(const BaDims& d, const BaBufs& b)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    // ---- loads that depend on nothing but the landmark index go out first, together with the state block
    const int lane = threadIdx.x & 63, l = lane & 15, sub = lane >> 4;
    const int p = blockIdx.x * 64 + (threadIdx.x >> 6) * 16 + l;
    const bool valid = p < d.P;
    const int set = blockIdx.y;             // speculative radius evaluated by this workgroup (ba_common.h)
    int o0 = 0, nobs = 0;
    double g[3] = {0, 0, 0}, I[6] = {0, 0, 0, 0, 0, 0}, lamp[3] = {0, 0, 0}, Xq[BA_MAXSETS + 1][3];
#pragma unroll
    for (int q = 0; q <= BA_MAXSETS; q++) Xq[q][0] = Xq[q][1] = Xq[q][2] = 0.0;
    if (valid) {
        o0 = b.obs_ptr[p];
        nobs = b.obs_ptr[p + 1] - o0;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            g[k] = b.gp[3 * (size_t)p + k];
            lamp[k] = b.lamp[((size_t)set * d.P + p) * 3 + k];
        }
        // x may live in any of the ns + 1 state buffers: all of them are fetched before the state block is known
#pragma unroll
        for (int q = 0; q <= BA_MAXSETS; q++)
            if (q <= b.ns) {
#pragma unroll
                for (int k = 0; k < 3; k++) Xq[q][k] = b.Xp[((size_t)q * d.P + p) * 3 + k];
            }
#pragma unroll
        for (int k = 0; k < 6; k++) I[k] = b.Vinv[((size_t)set * d.P + p) * 6 + k];
    }
    const BaState st = *b.st;
    if (st.done) return;
    const int set_failed = set == 0 ? st.solver_failed : b.set_out[set].solver_failed;
    // K7 has consumed the accumulators: clear them for the next linearisation (no separate launch)
    const size_t gtid = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
    const size_t gnth = (size_t)gridDim.y * gridDim.x * blockDim.x;
    for (size_t i = gtid; i < b.acc_count; i += gnth) b.acc[i] = 0.0;
    for (size_t i = gtid; i < BA_NSLOT * BA_SLOT_STRIDE; i += gnth) b.gmax[i] = 0.0;
    for (size_t i = gtid; i < (size_t)b.imu.zacc_n; i += gnth) b.imu.zacc[i] = 0.0;      // inertial accumulators (ba_imu.hip)
    if (set_failed || set >= st.nact) return;
    double* cprep = lds;                                    // [C][BA_PREP] current
    double* cprepn = lds + (size_t)d.C * BA_PREP;           // [C][BA_PREP] candidate
    double* dcl = cprepn + (size_t)d.C * BA_PREP;           // [n] delta_c
    const double* gprep = b.prep + (size_t)st.cur * d.C * BA_PREP;
    const int cand = (st.cur + 1 + set) % (b.ns + 1);        // this set's candidate buffer (written by K7)
    double* gprepn = b.prep + (size_t)cand * d.C * BA_PREP;
    // second round trip: K7's candidate camera blocks (prep[cur^1]) and the current ones -> LDS, and the
    // first two observations of every lane (later rounds load on demand)
    int cs_pre[2] = {0, 0};
    float2 uv_pre[2] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f)};
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int j = sub + 4 * r;
        if (j < nobs) {
            const int oi = o0 + j;
            if (b.obs_cs) cs_pre[r] = b.obs_cs[oi];
            else { const int c = b.obs_cam[oi]; cs_pre[r] = c | ((b.slot[c] + 1) << 16); }
            uv_pre[r] = b.obs_uv[oi];
        }
    }
    for (int i = threadIdx.x; i < d.C * BA_PREP; i += blockDim.x) { cprep[i] = gprep[i]; cprepn[i] = gprepn[i]; }
    for (int i = threadIdx.x; i < d.n; i += blockDim.x) dcl[i] = b.dc[(size_t)set * (d.n + 2) + i];
    __syncthreads();
    const double* prep = cprep;

    double* Xn = b.Xp + (size_t)cand * d.P * 3;
    double cost = 0.0, mcc = 0.0, ssq = 0.0, xsq = 0.0;
    double X[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        double v = Xq[0][k];
#pragma unroll
        for (int q = 1; q <= BA_MAXSETS; q++) v = (st.cur == q) ? Xq[q][k] : v;
        X[k] = v;
    }
    double t[3] = {0, 0, 0};
    ObsLin o;
    for (int j = sub, r = 0; j < nobs; j += 4, r++) {
        int cs;
        float2 uvv;
        if (r < 2) { cs = cs_pre[r & 1]; uvv = uv_pre[r & 1]; }
        else {
            const int oi = o0 + j;
            if (b.obs_cs) cs = b.obs_cs[oi];
            else { const int c = b.obs_cam[oi]; cs = c | ((b.slot[c] + 1) << 16); }
            uvv = b.obs_uv[oi];
        }
        const int c = cs & 0xFFFF, s = (cs >> 16) - 1;
        if (s < 0) continue;
        obs_eval<true>(prep + (size_t)c * BA_PREP, X, uvv, d, o);
        double m0 = 0.0, m1 = 0.0;
#pragma unroll
        for (int a = 0; a < 6; a++) { const double dc = dcl[6 * s + a]; m0 += o.jc[a] * dc; m1 += o.jc[6 + a] * dc; }
#pragma unroll
        for (int k = 0; k < 3; k++) t[k] += o.w * (o.jp[k] * m0 + o.jp[3 + k] * m1);   // W_i^T delta_c
    }
#pragma unroll
    for (int k = 0; k < 3; k++) { t[k] += __shfl_xor(t[k], 16, 64); t[k] += __shfl_xor(t[k], 32, 64); }
    double Xc[3] = {0, 0, 0};
    if (valid) {
        const double I0 = I[0], I1 = I[1], I2 = I[2], I3 = I[3], I4 = I[4], I5 = I[5];
        const double tt[3] = {t[0] + g[0], t[1] + g[1], t[2] + g[2]};
        const double dp[3] = {-(I0 * tt[0] + I1 * tt[1] + I2 * tt[2]), -(I1 * tt[0] + I3 * tt[1] + I4 * tt[2]),
                              -(I2 * tt[0] + I4 * tt[1] + I5 * tt[2])};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            Xc[k] = X[k] + dp[k];
            if (sub == 0) {
                Xn[3 * (size_t)p + k] = Xc[k];
                mcc += 0.5 * (dp[k] * dp[k] * lamp[k] - dp[k] * g[k]);
                ssq += (X[k] - Xc[k]) * (X[k] - Xc[k]);
                xsq += X[k] * X[k];
            }
        }
    }
    for (int j = sub, r = 0; j < nobs; j += 4, r++) {
        int c;
        float2 uvv;
        if (r < 2) { c = cs_pre[r & 1] & 0xFFFF; uvv = uv_pre[r & 1]; }
        else { c = b.obs_cam[o0 + j]; uvv = b.obs_uv[o0 + j]; }
        obs_eval<false>(cprepn + (size_t)c * BA_PREP, Xc, uvv, d, o);
        cost += 0.5 * o.rho;
    }
    cost = wave_sum(cost); mcc = wave_sum(mcc); ssq = wave_sum(ssq); xsq = wave_sum(xsq);
    __shared__ double redw[K8_THREADS / 64][4];
    if (lane == 0) { redw[threadIdx.x >> 6][0] = cost; redw[threadIdx.x >> 6][1] = mcc; redw[threadIdx.x >> 6][2] = ssq; redw[threadIdx.x >> 6][3] = xsq; }
    __syncthreads();
    if (threadIdx.x < 4) {       // one atomic per workgroup and scalar, spread over BA_NSLOT lines
        double v = 0.0;
        for (int w = 0; w < K8_THREADS / 64; w++) v += redw[w][threadIdx.x];
        atomicAdd(&b.pt_scal[((size_t)set * BA_NSLOT + (blockIdx.x & (BA_NSLOT - 1))) * BA_SLOT_STRIDE + threadIdx.x], v);
    }
}

__global__ __launch_bounds__(K8_THREADS) void ba_backsub_cost4(BaDims d, BaBufs b) { ba_backsub_cost4_body(d, b); }
// batched: blockIdx.x = landmark block, blockIdx.y = speculative set, blockIdx.z = window
__global__ __launch_bounds__(K8_THREADS) void ba_backsub_cost4_batch(const BaWin* w, int it)
{
    const BaWin& x = w[blockIdx.z];
    const BaBufs b = ba_win_round(x, it, false);
    ba_backsub_cost4_body(x.d, b);
}

size_t ba_backsub_lds_bytes(int C, int n)
{
    return sizeof(double) * (2 * (size_t)C * BA_PREP + (size_t)n + 8);
}

void ba_launch_backsub(hipStream_t s, const BaDims& d, const BaBufs& b)
{
    hipLaunchKernelGGL(ba_backsub_cost4, dim3((d.P + 63) / 64, b.ns), dim3(K8_THREADS), ba_backsub_lds_bytes(d.C, d.n), s, d, b);
}

void ba_launch_backsub_batch(hipStream_t s, const BaWin* d_wins, int B, int it, int ns, int max_P, size_t lds)
{
    if (lds > 48 * 1024) (void)rs_lds_attr((const void*)ba_backsub_cost4_batch, lds);
    hipLaunchKernelGGL(ba_backsub_cost4_batch, dim3((max_P + 63) / 64, ns, B), dim3(K8_THREADS), lds, s, d_wins, it);
}
