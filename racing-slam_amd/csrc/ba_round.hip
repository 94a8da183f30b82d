// ba_round.hip — one launch per LM round of the local-window bundle adjustment: K5, K7 and K8 of a round in ONE grid.
//
// A round used to be two launches, K5 (linearisation + Schur complement, one workgroup per item of landmarks) and the
// fused K7 + K8 launch (ba_solve.hip).  What separates them is a kernel boundary plus the part of each kernel that only
// re-establishes what the previous one already knew: K7's workgroups cannot do anything before S is complete, K8's
// workgroups idle until delta_c exists — and K5's workgroups are gone by then.  Here the same workgroups stay:
//
//   workgroups [0, ns)            K7, one per speculative radius (ba_solve_body.h, ROUND): decide the round, store the
//                                 state block (set 0), wait until every item workgroup has counted itself on BA_SDONE,
//                                 read the accumulators with L1-bypassing loads, solve, publish delta_c.
//   workgroups [ns, ns + items)   one item of landmarks each (ba_schur_body.h, ROUND): decision, linearisation, Schur
//                                 products into the accumulators with memory-side atomics, s_waitcnt vmcnt(0), barrier,
//                                 one count on BA_SDONE.  Then the SAME workgroup becomes a K8 workgroup
//                                 (ba_backsub_body.h, FUSED): its loads go out while K7 solves, it clears the accumulators
//                                 once every K7 has taken them and finishes behind its set's hand-off word.
//
// Hand-offs: the guide's one-producer rows (MI355X_MICROARCH.md, inter-workgroup visibility) — payload produced by
// memory-side atomics (the accumulators) or sc1 stores (delta_c), drained (vmcnt(0)) in every producing wave, a workgroup
// barrier, ONE lane signals (agent-scope atomic add / sc1 store); consumers poll with an sc1 load from one lane, follow
// behind a workgroup barrier and touch the payload with sc1 loads only.  No agent-scope fences anywhere.
// Residency: every workgroup of the grid must be resident at once (K7 waits for all items, the items' K8 phase waits for
// K7): the host launches this form only when ns + max(items, ns * nblk) <= number of CUs (one workgroup per CU: the launch
// carries K7's LDS).  Every wait is bounded by the wall clock (BaBufs::hand_timeout); a lost hand-off makes the solve
// unusable and the host runs it again as separate launches (ba.hip, ba_solve_impl), so the grid always drains.
#include "ba_common.h"
#include "ba_backsub_body.h"
#include "ba_schur_body.h"
#include "ba_solve_body.h"

// K8 of an item workgroup: back-substitution, candidate points and candidate cost of the item's OWN landmarks for every
// active speculative set — what it has just linearised.  (The K8 launch partitions the landmarks differently; inside one
// launch a workgroup may only consume per-landmark results — V^-1, damping, gradient — that it stored itself.)  Four lanes
// per (landmark, set) pair as in ba_backsub_body.h, 128 pairs per pass of the 512 threads; the arithmetic per pair is that
// body's, in the same order.  All sets' delta_c and candidate camera blocks live in LDS; the step scalars are summed per
// set through LDS atomics (a wave holds pairs of different sets).
static __device__ __forceinline__ void ba_backsub_item_body(const BaDims& d, const BaBufs& b, const BaGroup& g, const int item, const BaState& st)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    if (st.done) return;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & 63, l = lane & 15, sub = lane >> 4;
    const int C = d.C, n = d.n, nact = st.nact, it_l = g.it_l;
    double* cprep = lds;                                        // [C][BA_PREP_LDS] current cameras' blocks
    double* cprepn = cprep + (size_t)C * BA_PREP_LDS;           // [BA_MAXSETS][C][BA_PREP_LDS] candidates'
    double* dcl = cprepn + (size_t)BA_MAXSETS * C * BA_PREP_LDS;   // [BA_MAXSETS][n] delta_c
    double* xcl = dcl + (size_t)BA_MAXSETS * n;                 // [C][6] current cameras
    double* sums = xcl + (size_t)C * 6;                         // [BA_MAXSETS][4] cost, mcc, step^2, x^2
    int* sll = (int*)(sums + BA_MAXSETS * 4);                   // [C] slot map
    __shared__ unsigned hand_code[BA_MAXSETS];
    const int npairs = it_l * nact;
    const int ppp = nt >> 2;                                    // pairs per pass
    // ---- pass 0's loads go out first (this workgroup stored most of them a moment ago: cache hits)
    struct Pair { int p, set, o0, nobs; bool valid; double X[3], g[3], I[6], lamp[3]; int cs_pre[K8_PRE]; float2 uv_pre[K8_PRE]; };
    auto load_pair = [&](int pair, Pair& r) {
        r.valid = false; r.p = -1; r.set = 0; r.o0 = 0; r.nobs = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) { r.X[k] = 0.0; r.g[k] = 0.0; r.lamp[k] = 0.0; }
#pragma unroll
        for (int k = 0; k < 6; k++) r.I[k] = 0.0;
#pragma unroll
        for (int q = 0; q < K8_PRE; q++) { r.cs_pre[q] = 0; r.uv_pre[q] = make_float2(0.f, 0.f); }
        if (pair >= npairs) return;
        const int slot = pair % it_l;
        r.set = pair / it_l;
        const int q = item * it_l + slot;
        if (q >= d.P) return;
        const int4 lm = g.lm[q];
        r.p = lm.x; r.o0 = lm.y; r.nobs = lm.z; r.valid = true;
        const size_t p = (size_t)r.p;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            r.X[k] = b.Xp[((size_t)st.cur * d.P + p) * 3 + k];
            r.g[k] = b.gp[3 * p + k];
            r.lamp[k] = b.lamp[((size_t)r.set * d.P + p) * 3 + k];
        }
#pragma unroll
        for (int k = 0; k < 6; k++) r.I[k] = b.Vinv[((size_t)r.set * d.P + p) * 6 + k];
#pragma unroll
        for (int q2 = 0; q2 < K8_PRE; q2++) {
            const int j = sub + 4 * q2;
            if (j < r.nobs) { r.cs_pre[q2] = b.obs_cs[r.o0 + j]; r.uv_pre[q2] = b.obs_uv[r.o0 + j]; }
        }
    };
    Pair pr;
    load_pair((tid >> 6) * 16 + l, pr);
    if (tid < BA_MAXSETS * 4) sums[tid] = 0.0;
    // ---- every active set's K7 has taken the accumulators: clear them for the next linearisation
    if (tid < nact) hand_code[tid] = ba_hand_wait(b.dbg + BA_HAND_TAKEN, tid, (unsigned)st.n_rounds, b.hand_timeout);
    __syncthreads();
    {
        bool lost = false, conv = false;
        for (int k = 0; k < nact; k++) { lost = lost || hand_code[k] == 4u; conv = conv || (hand_code[k] & 2u); }
        if (lost) { if (tid == 0) atomicAdd(b.dbg + BA_HAND_ERR, 1ull); return; }
        if (conv) return;                           // converged in K7's gradient test: the solve is over
    }
#if RS_STAMPS
    if (tid == 0 && item == g.n_items / 2) b.dbg[34] = wall_clock64();
#endif
    {
        const size_t gtid = (size_t)item * nt + tid, gnth = (size_t)g.n_items * nt;
        for (size_t i = gtid; i < b.acc_count; i += gnth) b.acc[i] = 0.0;
        for (size_t i = gtid; i < BA_NSLOT * BA_SLOT_STRIDE; i += gnth) b.gmax[i] = 0.0;
    }
    const double* gprep = b.prep + (size_t)st.cur * C * BA_PREP;
    for (int i = tid; i < C * BA_PREP; i += nt) cprep[(i / BA_PREP) * BA_PREP_LDS + i % BA_PREP] = gprep[i];
    for (int i = tid; i < C * 6; i += nt) xcl[i] = b.Xc[(size_t)st.cur * C * 6 + i];
    for (int i = tid; i < C; i += nt) sll[i] = b.slot[i];
    __syncthreads();                                // hand_code is reused
    if (tid < nact) hand_code[tid] = ba_hand_wait(b.dbg + BA_HAND, tid, (unsigned)st.n_rounds, b.hand_timeout);
    __syncthreads();
    {
        bool lost = false;
        for (int k = 0; k < nact; k++) lost = lost || hand_code[k] == 4u;
        if (lost) { if (tid == 0) atomicAdd(b.dbg + BA_HAND_ERR, 1ull); return; }
    }
#if RS_STAMPS
    if (tid == 0 && item == g.n_items / 2) b.dbg[35] = wall_clock64();
#endif
    for (int i = tid; i < nact * n; i += nt) {
        const int k = i / n, e = i - k * n;
        dcl[i] = (hand_code[k] & 1u) ? 0.0 : ba_load_sc1(b.dc + (size_t)k * BA_DC_STRIDE(n) + e);
    }
    __syncthreads();
    // the candidates' cameras and their blocks, exactly as K7's epilogue forms them for the next linearisation
    for (int i = tid; i < nact * C; i += nt) {
        const int k = i / C, c = i - k * C, s = sll[c];
        double xn[6];
#pragma unroll
        for (int a = 0; a < 6; a++) xn[a] = s >= 0 ? xcl[6 * c + a] + dcl[k * n + 6 * s + a] : xcl[6 * c + a];
        cam_prepare(xn, cprepn + ((size_t)k * C + c) * BA_PREP_LDS);
    }
    __syncthreads();
    for (int base = 0; base < npairs; base += ppp) {
        if (base > 0) load_pair(base + (tid >> 6) * 16 + l, pr);
        const bool live = pr.valid && !(hand_code[pr.set] & 1u);        // (a set whose solver failed contributes nothing)
        const int set = pr.set, o0 = pr.o0, nobs = live ? pr.nobs : 0;
        const double* dc_set = dcl + (size_t)set * n;
        const double* prepn = cprepn + (size_t)set * C * BA_PREP_LDS;
        double t[3] = {0, 0, 0};
        ObsLin o;
        for (int j = sub, r = 0; j < nobs; j += 4, r++) {
            int cs;
            float2 uvv;
            if (r < K8_PRE) { cs = r == 0 ? pr.cs_pre[0] : r == 1 ? pr.cs_pre[1] : pr.cs_pre[2]; uvv = r == 0 ? pr.uv_pre[0] : r == 1 ? pr.uv_pre[1] : pr.uv_pre[2]; }
            else { cs = b.obs_cs[o0 + j]; uvv = b.obs_uv[o0 + j]; }
            const int c = cs & 0xFFFF, s = (cs >> 16) - 1;
            if (s < 0) continue;
            obs_eval<true>(cprep + (size_t)c * BA_PREP_LDS, pr.X, uvv, d, o);
            double m0 = 0.0, m1 = 0.0;
#pragma unroll
            for (int a = 0; a < 6; a++) { const double dc = dc_set[6 * s + a]; m0 += o.jc[a] * dc; m1 += o.jc[6 + a] * dc; }
#pragma unroll
            for (int k = 0; k < 3; k++) t[k] += o.w * (o.jp[k] * m0 + o.jp[3 + k] * m1);   // W_i^T delta_c
        }
#pragma unroll
        for (int k = 0; k < 3; k++) { t[k] += __shfl_xor(t[k], 16, 64); t[k] += __shfl_xor(t[k], 32, 64); }
        double cost = 0.0, mcc = 0.0, ssq = 0.0, xsq = 0.0;
        double Xc[3] = {0, 0, 0};
        if (live) {
            const double* I = pr.I;
            const double tt[3] = {t[0] + pr.g[0], t[1] + pr.g[1], t[2] + pr.g[2]};
            const double dp[3] = {-(I[0] * tt[0] + I[1] * tt[1] + I[2] * tt[2]), -(I[1] * tt[0] + I[3] * tt[1] + I[4] * tt[2]),
                                  -(I[2] * tt[0] + I[4] * tt[1] + I[5] * tt[2])};
            double* Xn = b.Xp + (size_t)((st.cur + 1 + set) % (b.ns + 1)) * d.P * 3;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                Xc[k] = pr.X[k] + dp[k];
                if (sub == 0) {
                    Xn[3 * (size_t)pr.p + k] = Xc[k];
                    mcc += 0.5 * (dp[k] * dp[k] * pr.lamp[k] - dp[k] * pr.g[k]);
                    ssq += (pr.X[k] - Xc[k]) * (pr.X[k] - Xc[k]);
                    xsq += pr.X[k] * pr.X[k];
                }
            }
        }
        for (int j = sub, r = 0; j < nobs; j += 4, r++) {
            int c;
            float2 uvv;
            if (r < K8_PRE) { c = (r == 0 ? pr.cs_pre[0] : r == 1 ? pr.cs_pre[1] : pr.cs_pre[2]) & 0xFFFF; uvv = r == 0 ? pr.uv_pre[0] : r == 1 ? pr.uv_pre[1] : pr.uv_pre[2]; }
            else { c = b.obs_cs[o0 + j] & 0xFFFF; uvv = b.obs_uv[o0 + j]; }
            obs_eval<false>(prepn + (size_t)c * BA_PREP_LDS, Xc, uvv, d, o);
            cost += 0.5 * o.rho;
        }
        cost += __shfl_xor(cost, 16, 64); cost += __shfl_xor(cost, 32, 64);      // the pair's four lanes
        if (live && sub == 0) {
            atomicAdd(&sums[set * 4 + 0], cost);
            atomicAdd(&sums[set * 4 + 1], mcc);
            atomicAdd(&sums[set * 4 + 2], ssq);
            atomicAdd(&sums[set * 4 + 3], xsq);
        }
    }
    __syncthreads();
    if (tid < nact * 4) {           // one atomic per workgroup, set and scalar, spread over BA_NSLOT lines
        const int set = tid >> 2, k = tid & 3;
        if (!(hand_code[set] & 1u))
            atomicAdd(&b.pt_scal[((size_t)set * BA_NSLOT + ((size_t)item & (BA_NSLOT - 1))) * BA_SLOT_STRIDE + k], sums[tid]);
    }
}

// K7 is compiled as a function of its own (not inlined into the kernel): inlined, the kernel's combined register pressure
// spilled 500 SGPRs and the factorisation chain ran 16 % slower than in ba_solve_backsub (taken -> delta_c 34.6 us against
// 29.8).  The item roles stay inlined: as functions their argument structures live in scratch and the linearisation went
// from 33 to 45 us.
static __device__ __attribute__((noinline)) void ba_round_k7(const BaDims d, const BaBufs b, const BaOpt opt, const int it, const int n_items)
{   // (arguments BY VALUE: references would force the kernel's own copies of these structures into scratch as well)
    ba_reduced_solve_lds_body<true, true>(d, b, opt, it, n_items);
}
template <bool PREP_LDS>
static __device__ __forceinline__ void ba_round_k5(const BaDims& d, const BaBufs& b, const BaOpt& opt, const BaGroup& g, const int it, const int item, BaState* st)
{
    ba_schur_body<PREP_LDS, true>(d, b, opt, g, it, item, st);
}
static __device__ __forceinline__ void ba_round_k8(const BaDims& d, const BaBufs& b, const BaGroup& g, const int item, const BaState* st)
{
    ba_backsub_item_body(d, b, g, item, *st);
}

template <bool PREP_LDS>
static __device__ __forceinline__ void ba_round_body(const BaDims& d, const BaBufs& b, const BaOpt& opt, const BaGroup& g, const int it, const int nblk)
{
    if ((int)blockIdx.x < b.ns) {
        ba_round_k7(d, b, opt, it, g.n_items);
        return;
    }
    __shared__ BaState st_round;
    const int v = (int)blockIdx.x - b.ns;
#if RS_STAMPS
    const bool stamp = threadIdx.x == 0 && v == g.n_items / 2;
    if (stamp) b.dbg[32] = wall_clock64();
#endif
    ba_round_k5<PREP_LDS>(d, b, opt, g, it, v, &st_round);
#if RS_STAMPS
    if (stamp) b.dbg[33] = wall_clock64();
#endif
    __syncthreads();                                                  // the linearisation's LDS image is dead from here
    ba_round_k8(d, b, g, v, &st_round);
#if RS_STAMPS
    if (stamp) b.dbg[36] = wall_clock64();
#endif
    (void)nblk;
}

__global__ __launch_bounds__(K7_THREADS) void ba_round(BaDims d, BaBufs b, BaOpt opt, BaGroup g, int it, int nblk)
{
    ba_round_body<true>(d, b, opt, g, it, nblk);
}

size_t ba_round_lds_bytes(const BaDims& d)
{
    size_t v = ba_reduced_solve_lds_bytes(d.n);
    const size_t k8 = sizeof(double) * ((size_t)(1 + BA_MAXSETS) * d.C * BA_PREP_LDS + (size_t)BA_MAXSETS * d.n + 6 * (size_t)d.C + 4 * BA_MAXSETS + (size_t)d.C);
    if (k8 > v) v = k8;
    if (ba_schur_lds_bytes(d.C, d.Cf) > v) v = ba_schur_lds_bytes(d.C, d.Cf);
    return v;
}

int ba_round_workgroups(const BaDims& d, const BaBufs& b, const BaGroup& g)
{
    (void)d;
    return b.ns + g.n_items;
}

// the window fits this form: camera blocks in LDS (<= SCH_MAXC_LDS cameras); the caller has checked the LDS solve,
// the LDS back-substitution and the residency (ba_round_workgroups <= CUs)
bool ba_round_eligible(const BaDims& d) { return d.C <= SCH_MAXC_LDS; }

int ba_prepare_round(const BaDims& d) { return (int)rs_lds_attr((const void*)ba_round, ba_round_lds_bytes(d)); }

void ba_launch_round(hipStream_t s, const BaDims& d, const BaBufs& b, const BaOpt& opt, const BaGroup& g, int it)
{
    const int per = K7_THREADS / 4, nblk = (d.P + per - 1) / per;
    hipLaunchKernelGGL(ba_round, dim3(ba_round_workgroups(d, b, g)), dim3(K7_THREADS), ba_round_lds_bytes(d), s, d, b, opt, g, it, nblk);
}
