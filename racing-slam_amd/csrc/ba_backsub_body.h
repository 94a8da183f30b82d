// ba_backsub_body.h — the body of K8 (ba_update.hip), shared with the fused solve + back-substitution kernel
// (ba_solve.hip, ba_solve_backsub): see ba_update.hip for what it computes.
#pragma once
#include "ba_common.h"

#define K8_THREADS 256
#define K8_PRE 3                 // observation rounds (4 lanes each) of a landmark held in registers
#define K8_MAX_THREADS 512        // the fused kernel runs this body with K7's 512 threads (128 landmarks per workgroup)

// Hand-off words of the fused kernel (ba_solve.hip, ba_solve_backsub), one pair per speculative set, each
// round << 2 | code:
//   b.dbg[BA_HAND_TAKEN + set]  K7 holds the accumulators in registers / LDS (code 0), or its gradient test has ended
//                               the solve (code 2: nothing left to do)
//   b.dbg[BA_HAND + set]        delta_c is complete in memory (code 0), or the solver has failed (code 1)
// b.dbg[BA_HAND_ERR] counts consumers that gave up waiting (ba_finalize turns that into RS_BA_FAILURE).
#define BA_HAND 64              // [BA_MAXSETS] (8 reserved)
#define BA_HAND_TAKEN 72        // [BA_MAXSETS]
#define BA_HAND_ERR 80
// One launch per LM round (ba_round, ba_solve.hip): the item workgroups count themselves here when their part of the
// linearisation — S, rhs, U, gc, cost and failure slots, all accumulated by memory-side atomics — is complete; the K7
// workgroups wait for n_rounds * n_items.  A line of its own (the stamps use 0 - 15 and 32 - 43).
#define BA_SDONE 24
#define BA_HAND_TIMEOUT_TICKS 400000ull      // default: 4 ms of the 100 MHz wall clock (BaBufs::hand_timeout; rs_context_set_int
                                             // "ba_handoff_timeout_us"): a lost producer must not hang the GPU

// L1-bypassing (sc1) accesses of handed-off data.  The pointers are cast to the GLOBAL address space: inside a function that
// is not inlined into its kernel (ba_round's K7) the compiler only knows generic pointers and would emit flat_load / flat_store,
// which the hand-off forms of MI355X_MICROARCH.md exclude ("global_ / buffer_ sc1 loads to registers, never flat_").
typedef __attribute__((address_space(1))) double ba_gdouble;
typedef __attribute__((address_space(1))) unsigned long long ba_gu64;
__device__ __forceinline__ double ba_load_sc1(const double* p) { return __hip_atomic_load((const ba_gdouble*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ba_store_sc1(double* p, double v) { __hip_atomic_store((ba_gdouble*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ba_load_word_sc1(const unsigned long long* p) { return __hip_atomic_load((const ba_gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ba_store_word_sc1(unsigned long long* p, unsigned long long v) { __hip_atomic_store((ba_gu64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one lane: waits until set k's producer of round `want` has published the word; returns its code, or 4 after the time-out
__device__ __forceinline__ unsigned ba_hand_wait(const unsigned long long* hand, int k, unsigned want, unsigned long long timeout_ticks)
{
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        const unsigned long long v = ba_load_word_sc1(hand + k);
        if ((unsigned)(v >> 2) == want) return (unsigned)(v & 3ull);
        if (wall_clock64() - t0 > timeout_ticks) return 4u;
        __builtin_amdgcn_s_sleep(2);
    }
}

#if RS_STAMPS
#define K78_STAMP(b, i) do { if (threadIdx.x == 0 && (FUSED) && vb == 0 && set0 == 0 && pass == 0) (b).dbg[32 + (i)] = wall_clock64(); } while (0)
#else
#define K78_STAMP(b, i) do { } while (0)
#endif

// FUSED = false: the K8 launch (blockIdx.x = landmark block of 64, blockIdx.y = set; K7 has finished).
// FUSED = true: a consumer workgroup of ba_solve_backsub: landmark block `vb` of blockDim.x / 4 landmarks, sets `set0`,
// `set0 + set_stride`, ... of the round's active ones (set_stride = 0: `set0` only).  The fused grid holds the workgroups of
// as many sets as are resident together (<= BA_CALIBRATED_SETS; fewer for windows of more than 10.7 k landmarks); in a deeper
// round a workgroup evaluates its next radius with the landmark records, the observations and the current cameras' blocks it
// already holds, instead of a second shift of workgroups that would start from their own loads behind the hand-off.
// All loads that do not depend on K7 are issued first; the accumulators are cleared once every active set's K7 has
// taken them (BA_HAND_TAKEN); delta_c is read behind the set's BA_HAND word with L1-bypassing loads, and the candidate
// cameras' blocks are formed here (K7 forms the same for the next round after it has published).
template <bool FUSED>
// st_in: the round's state where the caller holds it already (ba_round: the state block is being written by another
// workgroup of the same launch); nullptr = read the state block.
static __device__ __forceinline__ void ba_backsub_cost4_body(const BaDims& d, const BaBufs& b, const int vb, const int set0, const int set_stride,
                                                            const size_t wg_index, const size_t wg_count, const BaState* st_in = nullptr)
{

    extern __shared__ __attribute__((aligned(16))) double lds[];
    // ---- loads that depend on nothing but the landmark index go out first, together with the state block
    const int lane = threadIdx.x & 63, l = lane & 15, sub = lane >> 4;
    const int p = vb * (int)(blockDim.x >> 2) + (threadIdx.x >> 6) * 16 + l;
    const bool valid = p < d.P;
    int pass = 0;
    const int set1 = set_stride > 0 && set0 + set_stride < b.ns ? set0 + set_stride : -1;     // second radius (its V^-1 and damping travel with the first's)
    int o0 = 0, nobs = 0;
    double g[3] = {0, 0, 0}, I[6] = {0, 0, 0, 0, 0, 0}, lamp[3] = {0, 0, 0}, I1[6] = {0, 0, 0, 0, 0, 0}, lamp1[3] = {0, 0, 0}, Xq[BA_MAXSETS + 1][3];
#pragma unroll
    for (int q = 0; q <= BA_MAXSETS; q++) Xq[q][0] = Xq[q][1] = Xq[q][2] = 0.0;
    if (valid) {
        o0 = b.obs_ptr[p];
        nobs = b.obs_ptr[p + 1] - o0;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            g[k] = b.gp[3 * (size_t)p + k];
            lamp[k] = b.lamp[((size_t)set0 * d.P + p) * 3 + k];
            if (set1 >= 0) lamp1[k] = b.lamp[((size_t)set1 * d.P + p) * 3 + k];
        }
        // x may live in any of the ns + 1 state buffers: all of them are fetched before the state block is known
#pragma unroll
        for (int q = 0; q <= BA_MAXSETS; q++)
            if (q <= b.ns) {
#pragma unroll
                for (int k = 0; k < 3; k++) Xq[q][k] = b.Xp[((size_t)q * d.P + p) * 3 + k];
            }
#pragma unroll
        for (int k = 0; k < 6; k++) {
            I[k] = b.Vinv[((size_t)set0 * d.P + p) * 6 + k];
            if (set1 >= 0) I1[k] = b.Vinv[((size_t)set1 * d.P + p) * 6 + k];
        }
    }
    K78_STAMP(b, 0);
    const BaState st = st_in ? *st_in : *b.st;
    if (st.done) return;
    // K7 has consumed the accumulators: clear them for the next linearisation (no separate launch)
    const size_t gtid = wg_index * blockDim.x + threadIdx.x;
    const size_t gnth = wg_count * blockDim.x;
    auto clear_accumulators = [&]() {
        for (size_t i = gtid; i < b.acc_count; i += gnth) b.acc[i] = 0.0;
        for (size_t i = gtid; i < BA_NSLOT * BA_SLOT_STRIDE; i += gnth) b.gmax[i] = 0.0;
        for (size_t i = gtid; i < (size_t)b.imu.zacc_n; i += gnth) b.imu.zacc[i] = 0.0;      // inertial accumulators (ba_imu.hip)
    };
    __shared__ unsigned hand_code[BA_MAXSETS];
    if (!FUSED) {
        const int set_failed = set0 == 0 ? st.solver_failed : b.set_out[set0].solver_failed;
        clear_accumulators();
        if (set_failed || set0 >= st.nact) return;
    } else {
        // every active set's K7 has taken the accumulators: clear them now, under the factorisation
        if ((int)threadIdx.x < st.nact) hand_code[threadIdx.x] = ba_hand_wait(b.dbg + BA_HAND_TAKEN, (int)threadIdx.x, (unsigned)st.n_rounds, b.hand_timeout);
        __syncthreads();
        bool lost = false, conv = false;
        for (int k = 0; k < st.nact; k++) { lost = lost || hand_code[k] == 4u; conv = conv || (hand_code[k] & 2u); }
        if (lost) { if (threadIdx.x == 0) atomicAdd(b.dbg + BA_HAND_ERR, 1ull); return; }
        if (conv) return;                           // converged in K7's gradient test: the solve is over (st.done there)
        clear_accumulators();
        if (set0 >= st.nact) return;
        __syncthreads();                            // hand_code is reused below
    }
    double* cprep = lds;                                    // [C][BA_PREP_LDS] current
    double* cprepn = lds + (size_t)d.C * BA_PREP_LDS;       // [C][BA_PREP_LDS] candidate
    double* dcl = cprepn + (size_t)d.C * BA_PREP_LDS;       // [n] delta_c
    double* xcl = dcl + d.n;                                // [C][6] current cameras    (fused only)
    int* sll = (int*)(xcl + (size_t)d.C * 6);               // [C] slot map              (fused only)
    const double* gprep = b.prep + (size_t)st.cur * d.C * BA_PREP;
    // second round trip: the current cameras' blocks -> LDS, and the first K8_PRE observations of every lane (12 per
    // landmark; later rounds load on demand, inside the loops)
    int cs_pre[K8_PRE];
    float2 uv_pre[K8_PRE];
#pragma unroll
    for (int r = 0; r < K8_PRE; r++) {
        cs_pre[r] = 0; uv_pre[r] = make_float2(0.f, 0.f);
        const int j = sub + 4 * r;
        if (j < nobs) {
            const int oi = o0 + j;
            if (b.obs_cs) cs_pre[r] = b.obs_cs[oi];
            else { const int c = b.obs_cam[oi]; cs_pre[r] = c | ((b.slot[c] + 1) << 16); }
            uv_pre[r] = b.obs_uv[oi];
        }
    }
    for (int i = threadIdx.x; i < d.C * BA_PREP; i += blockDim.x) cprep[(i / BA_PREP) * BA_PREP_LDS + i % BA_PREP] = gprep[i];
    if (FUSED) {
        // everything above is in flight while this set's K7 (workgroup `set` of the same launch) is still solving
        for (int i = threadIdx.x; i < d.C * 6; i += blockDim.x) xcl[i] = b.Xc[(size_t)st.cur * d.C * 6 + i];
        for (int i = threadIdx.x; i < d.C; i += blockDim.x) sll[i] = b.slot[i];
    }
    K78_STAMP(b, 1);
    double X[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        double v = Xq[0][k];
#pragma unroll
        for (int q = 1; q <= BA_MAXSETS; q++) v = (st.cur == q) ? Xq[q][k] : v;
        X[k] = v;
    }
    const double* prep = cprep;
    __shared__ double redw[K8_MAX_THREADS / 64][4];

#pragma unroll 1
    for (;; pass++) {
        const int set = set0 + pass * set_stride;               // the radius evaluated in this pass (ba_common.h)
        if (pass && (set_stride <= 0 || set >= b.ns)) break;
        if (set >= st.nact) break;
        if (pass == 1) {
#pragma unroll
            for (int k = 0; k < 6; k++) I[k] = I1[k];
#pragma unroll
            for (int k = 0; k < 3; k++) lamp[k] = lamp1[k];
        } else if (pass > 1 && valid) {                         // (fewer than three sets resident: a third pass loads its own)
#pragma unroll
            for (int k = 0; k < 6; k++) I[k] = b.Vinv[((size_t)set * d.P + p) * 6 + k];
#pragma unroll
            for (int k = 0; k < 3; k++) lamp[k] = b.lamp[((size_t)set * d.P + p) * 3 + k];
        }
        if (pass) __syncthreads();                              // dcl, cprepn, hand_code and redw are reused
        const int cand = (st.cur + 1 + set) % (b.ns + 1);        // this set's candidate buffer (written by K7)
        const double* gprepn = b.prep + (size_t)cand * d.C * BA_PREP;
        if (!FUSED) {
            // K7's candidate camera blocks and its delta_c
            for (int i = threadIdx.x; i < d.C * BA_PREP; i += blockDim.x) cprepn[(i / BA_PREP) * BA_PREP_LDS + i % BA_PREP] = gprepn[i];
            for (int i = threadIdx.x; i < d.n; i += blockDim.x) dcl[i] = b.dc[(size_t)set * BA_DC_STRIDE(d.n) + i];
        } else {
            if (threadIdx.x == 0) hand_code[set] = ba_hand_wait(b.dbg + BA_HAND, set, (unsigned)st.n_rounds, b.hand_timeout);
            K78_STAMP(b, 2);
            __syncthreads();
            const unsigned code = hand_code[set];
            if (code == 4u) { if (threadIdx.x == 0) atomicAdd(b.dbg + BA_HAND_ERR, 1ull); return; }
            if (code & 1u) continue;                    // this set's solver failed
            for (int i = threadIdx.x; i < d.n; i += blockDim.x) dcl[i] = ba_load_sc1(b.dc + (size_t)set * BA_DC_STRIDE(d.n) + i);
            __syncthreads();
            // the candidate's cameras and their blocks, exactly as K7's epilogue forms them for the next linearisation
            for (int c = threadIdx.x; c < d.C; c += blockDim.x) {
                const int s = sll[c];
                double xn[6];
#pragma unroll
                for (int k = 0; k < 6; k++) xn[k] = s >= 0 ? xcl[6 * c + k] + dcl[6 * s + k] : xcl[6 * c + k];
                cam_prepare(xn, cprepn + (size_t)c * BA_PREP_LDS);
            }
        }
        __syncthreads();
        K78_STAMP(b, 3);

        double* Xn = b.Xp + (size_t)cand * d.P * 3;
        double cost = 0.0, mcc = 0.0, ssq = 0.0, xsq = 0.0;
        double t[3] = {0, 0, 0};
        ObsLin o;
        for (int j = sub, r = 0; j < nobs; j += 4, r++) {
            int cs;
            float2 uvv;
            if (r < K8_PRE) { cs = r == 0 ? cs_pre[0] : r == 1 ? cs_pre[1] : cs_pre[2]; uvv = r == 0 ? uv_pre[0] : r == 1 ? uv_pre[1] : uv_pre[2]; }
            else {
                const int oi = o0 + j;
                if (b.obs_cs) cs = b.obs_cs[oi];
                else { const int c = b.obs_cam[oi]; cs = c | ((b.slot[c] + 1) << 16); }
                uvv = b.obs_uv[oi];
            }
            const int c = cs & 0xFFFF, s = (cs >> 16) - 1;
            if (s < 0) continue;
            obs_eval<true>(prep + (size_t)c * BA_PREP_LDS, X, uvv, d, o);
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
            const double I0 = I[0], I1_ = I[1], I2 = I[2], I3 = I[3], I4 = I[4], I5 = I[5];
            const double tt[3] = {t[0] + g[0], t[1] + g[1], t[2] + g[2]};
            const double dp[3] = {-(I0 * tt[0] + I1_ * tt[1] + I2 * tt[2]), -(I1_ * tt[0] + I3 * tt[1] + I4 * tt[2]),
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
            if (r < K8_PRE) { c = (r == 0 ? cs_pre[0] : r == 1 ? cs_pre[1] : cs_pre[2]) & 0xFFFF; uvv = r == 0 ? uv_pre[0] : r == 1 ? uv_pre[1] : uv_pre[2]; }
            else { c = b.obs_cam[o0 + j]; uvv = b.obs_uv[o0 + j]; }
            obs_eval<false>(cprepn + (size_t)c * BA_PREP_LDS, Xc, uvv, d, o);
            cost += 0.5 * o.rho;
        }
        cost = wave_sum(cost); mcc = wave_sum(mcc); ssq = wave_sum(ssq); xsq = wave_sum(xsq);
        if (lane == 0) { redw[threadIdx.x >> 6][0] = cost; redw[threadIdx.x >> 6][1] = mcc; redw[threadIdx.x >> 6][2] = ssq; redw[threadIdx.x >> 6][3] = xsq; }
        __syncthreads();
        if (threadIdx.x < 4) {       // one atomic per workgroup and scalar, spread over BA_NSLOT lines
            double v = 0.0;
            for (int w = 0; w < (int)(blockDim.x >> 6); w++) v += redw[w][threadIdx.x];
            atomicAdd(&b.pt_scal[((size_t)set * BA_NSLOT + ((size_t)vb & (BA_NSLOT - 1))) * BA_SLOT_STRIDE + threadIdx.x], v);
        }
        K78_STAMP(b, 4);
    }
}

