// ba_solve_body.h — the body of K7 (the reduced camera system of the local-window BA in one workgroup; see ba_solve.hip
// for the layout and the roles), shared by the K7 / K7 + K8 kernels (ba_solve.hip) and by the one-launch-per-round kernel
// (ba_round.hip).
#pragma once
#include "ba_common.h"
#include "ba_backsub_body.h"

#define K7_THREADS 512
// K7_CHAIN_SIMDS 1 (default): chain waves 0 and 4 share SIMD 0, six tile waves on SIMDs 1-3.  2: chain waves 0 and 1 have a
// SIMD each (while both work — more than 64 rows left — two chain waves on one SIMD take turns: 1.4 us per block step
// against 1.0), waves 2, 3, 6, 7 are the tile waves on SIMDs 2 and 3 with nine tiles each, waves 4 and 5 only keep the
// barriers.  Built, parity-green and measured in round 3: the fused launch 45.0 -> 51.0 us, the two-launch K7 40.6 -> 46.9 —
// four tile waves do not keep up with the chain (eighteen MFMAs, nine operand pairs and up to nine publishes per wave and
// step), so the trailing update becomes the critical path.  Kept as a compile-time switch.
#ifndef K7_CHAIN_SIMDS
#define K7_CHAIN_SIMDS 1
#endif
// K7_V2 1 (round 4; built, parity-green, NOT faster — profiles/round4_k7_step_breakdown.txt): the chain waves keep a FIXED row -> lane map (wave 0: rows 0 .. 63, wave 4: rows 64 .. n), so the
// rank-6 fix-up of the next block column stays in registers from step to step instead of travelling through LDS; the six
// rows of the diagonal block go through a 36-double scratch, with a word per chain wave where the block straddles row 64
// (no workgroup barrier); the tile waves derive their MFMA operands from the factor panel itself (A, row-major) and the
// pivots; a block step costs ONE workgroup barrier instead of two.  Sized by tools/microbench/f64_latency.hip: one wave
// issues a vector instruction every ~6 cycles whether or not it depends on the one before (f64 fma 6.0 dependent, 5.5
// independent; v_rcp_f64 26; a double through v_readlane 48; dependent ds_read 60), and two waves on one SIMD each keep that
// rate — the chain is bound by the instruction COUNT per wave, so the step's row work is split over two waves and nothing
// is computed or moved twice.  K7_V2 2: the same without ANY workgroup barrier inside the loop — progress words in LDS between
// the chain waves and the tile waves (every spin bounded).  K7_V2 0 (default) is round 1's two-barrier form.
#ifndef K7_V2
#define K7_V2 0
#endif
#if K7_CHAIN_SIMDS == 2
#define K7_TPW 9       // tiles per tile wave: 4 tile waves x 9 = the 36 lower-triangle tiles of a 128 x 128 matrix
#define K7_NTW 4
#else
#define K7_TPW 6       // 6 tile waves x 6
#define K7_NTW 6
#endif

typedef __attribute__((ext_vector_type(4))) double d4;

// broadcast one lane's double to the wave (lane is wave-uniform)
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// 1/x from v_rcp_f64 + one Newton step
__device__ __forceinline__ double fast_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r, 1.0);
    return fma(r, e, r);
}


// HANDOFF: this workgroup is a producer of the fused launch (ba_solve_backsub below).  It publishes two words per set
// (ba_backsub_body.h): BA_HAND_TAKEN once the accumulators are in registers / LDS and the gradient test is done (K8 clears
// them while the factorisation runs), and BA_HAND when delta_c — stored write-through (sc1) — is complete, or the solver
// has failed.  Every exit past the common early-out publishes what the consumers wait for.
// ROUND (implies HANDOFF): the workgroup is a K7 workgroup of ba_round (ba_round.hip, one launch per LM round).  It owns the
// state block (set 0 decides the round and stores the state), waits until all `n_items` item workgroups of the launch have
// counted themselves on BA_SDONE — their atomics into the accumulators are then performed — and reads the accumulators with
// L1-bypassing loads (they were produced inside this launch).
template <bool HANDOFF, bool ROUND = false>
static __device__ __forceinline__ void ba_reduced_solve_lds_body(const BaDims& d, const BaBufs& b, const BaOpt& opt, const int it = 0,
                                                                 const int n_items = 0)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int n = d.n, tid = threadIdx.x, nt = K7_THREADS;
    const int set = blockIdx.x;                    // speculative radius evaluated by this workgroup (ba_common.h)
    const int LD = n + 1 + ((n & 1) ? 1 : 0);     // odd row stride (n is a multiple of 6)
    double* A = sm;                                // (n+1) x LD: published panels + rhs row n
    double* lam = A + (size_t)(n + 1) * LD;        // [n] camera damping
    double* xs = lam + n;                          // [n] solution
    double* Minv = xs + n;                         // [n/6][36] inverses of the diagonal blocks of L
    double* Us = Minv + 6 * n;                     // [Cf*36] U folded over the BA_UREP replicas
    double* gcs = Us + 6 * n;                      // [n] gc folded
    double* grs = gcs + n;                         // [n] gc + rhs folded: the reduced right-hand side
    // MFMA operand panels of the current step, k-major: Pd[e][i] = F[i][e] d_e, Nf[e][i] = -F[i][e];
    // rows e = 6, 7 stay zero (K = 6 padded to 8).  16-byte aligned.
    double* Pd = sm + (((size_t)(n + 1) * LD + 17 * (size_t)n + 1) & ~(size_t)1);
    double* Nf = Pd + 8 * 128;
    __shared__ BaState st;
    __shared__ int s_fail;
    __shared__ double red[K7_THREADS / 64];
    __shared__ double red3[K7_THREADS / 64][3];
    BA_STAMP_DECL;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const int NTL = (n + 1 + 15) / 16;             // tile rows/cols
    // Roles.  f64 MFMA and f64 VALU share one datapath per SIMD (measured: they do not overlap, neither
    // within a wave nor between two waves of one SIMD), and waves w, w+4 of a workgroup land on the same
    // SIMD.  So the latency chain of the factorisation (diagonal block -> panel -> next diagonal block)
    // gets a SIMD of its own: waves 0 and 4 are the CHAIN waves (one matrix row per lane, 128 >= n+1-6
    // rows), waves 1,2,3,5,6,7 are the TILE waves that keep the trailing matrix in MFMA accumulators.
#if K7_CHAIN_SIMDS == 2
    const bool chain = wave < 2;                   // waves 0 and 1
    const bool idle = wave == 4 || wave == 5;      // (they share the chain waves' SIMDs)
    const int tw = (wave & 1) + 2 * (wave >> 2);   // tile waves 2, 3, 6, 7 -> 0 .. 3
    const int cw = wave;                           // chain wave index
#elif K7_V2
    const bool chain = (wave & 3) == 0;            // waves 0 and 4: rows 0 .. 63 and 64 .. n (one row per lane, fixed)
    const bool idle = false;
    const int tw = chain ? 0 : wave - 1 - (wave >> 2);     // tile wave index 0..5
#else
    const bool chain = (wave & 3) == 0;            // waves 0 and 4
    const bool idle = false;
    const int tw = chain ? 0 : wave - 1 - (wave >> 2);     // tile wave index 0..5
    const int cw = wave >> 2;
#endif
    // slot s of tile wave tw holds lower-triangle tile number t = K7_NTW s + tw, tiles numbered row by row
    int tr[K7_TPW], tc[K7_TPW];
    bool tv[K7_TPW];
#pragma unroll
    for (int s = 0; s < K7_TPW; s++) {
        const int t = K7_NTW * s + tw;
        int r = 0;
#pragma unroll
        for (int q = 1; q < 8; q++) r += (t >= q * (q + 1) / 2) ? 1 : 0;
        tr[s] = r;
        tc[s] = t - r * (r + 1) / 2;
        tv[s] = !chain && !idle && r < NTL;
    }

    // accumulators: plain loads behind a kernel boundary, L1-bypassing ones when they were produced inside this launch
    auto acc_ld = [&](const double* p) -> double { return ROUND ? ba_load_sc1(p) : *p; };
    if (ROUND) {
        // the round's decision (every workgroup of the launch computes the same; set 0 owns the state block), then the
        // wait for the item workgroups: one lane polls, the others follow behind the barrier
#if RS_STAMPS
        if (tid == 0 && set == 0) b.dbg[45] = wall_clock64();
#endif
        if (tid == 0) s_fail = 0;
        const BaState s0 = ba_round_state(b, opt, it, &st, set == 0);
        if (s0.done || set >= s0.nact) return;
        __shared__ unsigned s_lost;
        if (tid == 0) {
            const unsigned long long want = (unsigned long long)(unsigned)s0.n_rounds * (unsigned long long)n_items;
            const unsigned long long t0 = wall_clock64();
            unsigned lost = 0;
            while (ba_load_word_sc1(b.dbg + BA_SDONE) < want) {
                if (wall_clock64() - t0 > b.hand_timeout) { lost = 1; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            s_lost = lost;
#if RS_STAMPS
            if (set == 0) b.dbg[44] = wall_clock64();
#endif
        }
        __syncthreads();
        if (s_lost) {
            // an item workgroup never arrived (not resident, or preempted): nothing can be solved.  Release the consumers
            // ("solve over") and flag the launch; the host re-runs the solve as separate launches.
            if (tid == 0) {
                atomicAdd(b.dbg + BA_HAND_ERR, 1ull);
                const unsigned long long w = ((unsigned long long)(unsigned)s0.n_rounds << 2) | 2ull;
                ba_store_word_sc1(b.dbg + BA_HAND_TAKEN + set, w);
            }
            return;
        }
    }
    // ---- every global load of the prologue is issued here, up front, so that the ~1 us latencies of the
    // state block, K5's slot sums, the Jacobi scale, the accumulator replicas and S overlap
    if (!ROUND && tid == 0) { st = *b.st; s_fail = 0; }
    double pre_cost = 0.0, pre_fail = 0.0, pre_gm = 0.0;
    if (tid < 64) {
        pre_cost = acc_ld(&b.scal[(size_t)tid * BA_SLOT_STRIDE + 0]);
        pre_fail = acc_ld(&b.scal[(size_t)tid * BA_SLOT_STRIDE + 1 + set]);
        for (int r = 0; r < b.gmax_blocks; r++)      // every rank's block (they arrive through the SUM all-reduce)
            pre_gm = fmax(pre_gm, acc_ld(&b.gmax_all[((size_t)r * BA_NSLOT + tid) * BA_SLOT_STRIDE]));
    }
    const double pre_sc = tid < n ? b.sc[tid] : 1.0;                   // n <= 126: one entry per thread
    double fold[2] = {0.0, 0.0}, keep[2] = {0.0, 0.0};                 // 8 n <= 1008 entries: two per thread
    // replica layout: rhs[ns][n] U[Cf*36] gc[n]; this set's view of it is entry i < n -> its rhs, i >= n -> U | gc
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int i = tid + h * nt;
        if (i < 8 * n) {
            const size_t addr = i < n ? (size_t)set * n + i : (size_t)(b.ns - 1) * n + i;
#pragma unroll
            for (int r = 0; r < BA_UREP; r++) fold[h] += acc_ld(&b.rhs[(size_t)r * b.cam_stride + addr]);
            if (i >= n) keep[h] = b.Ukeep[i - n];                      // U | gc of the last fresh linearisation
        }
    }
    // (Tried twice: S through LDS — all 512 threads read S with consecutive lanes on consecutive entries of a row into the
    // panel area, the tile waves gather from there.  As a rolled loop of 8-byte loads: K7 37.9 -> 47.5 us; as 16 16-byte
    // loads per thread all in flight: 40.6 -> 44.1 us (64 more live registers, two more barriers, a 2- to 4-way bank
    // conflicted LDS gather).  The gather below keeps 24 independent loads per lane in flight and costs ~2 us.)
    double sv[K7_TPW][4];                                              // tile waves: their entries of S
#pragma unroll
    for (int s = 0; s < K7_TPW; s++) {
#pragma unroll
        for (int q = 0; q < 4; q++) sv[s][q] = 0.0;
        if (!tv[s]) continue;                                          // wave-uniform
        const int kc = min(16 * tc[s] + lr, n - 1);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int ic = min(16 * tr[s] + lq + 4 * q, n - 1);
            const double* sp = b.S + ((size_t)set * n + kc) * n + ic;   // only k <= i is used: (k, i) is S's upper triangle
            double v = acc_ld(sp);
            for (int r = 1; r < b.srep; r++) v += acc_ld(sp + (size_t)r * b.s_rep_stride);      // K5 scattered into srep replicas
            sv[s][q] = v;
        }
    }
    __syncthreads();
    if (st.done || set >= st.nact) return;        // (the consumers of the fused launch take the same exit on the same state)
#if RS_STAMPS
    if (HANDOFF && tid == 0 && set == 0) b.dbg[42] = wall_clock64();
#endif
    const unsigned long long epoch = (unsigned long long)(unsigned)st.n_rounds << 2;
    auto publish = [&](int word, unsigned long long code) {   // reached by every thread of the workgroup
        if (HANDOFF) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's loads have arrived, its stores are done
            __syncthreads();
            if (tid == 0) {
#if RS_STAMPS
                if (set == 0) b.dbg[word == BA_HAND ? 40 : 43] = wall_clock64();
#endif
                ba_store_word_sc1(b.dbg + word + set, epoch | code);
            }
        }
    };
    BA_STAMP(b, 2);
    for (int i = tid; i < BA_NSLOT * BA_SLOT_STRIDE; i += nt) {                       // K8 of this round accumulates here
        double* z = b.pt_scal + (size_t)set * BA_NSLOT * BA_SLOT_STRIDE + i;
        // fused launch: K8's atomics execute at the memory side within this launch; a plain store would sit in this
        // XCD's L2 until the kernel ends and then overwrite their sums.  Written through, and complete before publish().
        if (HANDOFF) ba_store_sc1(z, 0.0);
        else *z = 0.0;
    }
    for (int i = tid; i < 2 * 8 * 128; i += nt) Pd[i] = 0.0;
#if K7_V2
    // the chain reads whole block columns, entries above the diagonal included (they only reach dead values, but must be
    // finite), and the tile waves read operand rows of A that were never published
    for (int i = tid; i < (n + 1) * LD; i += nt) A[i] = 0.0;
#endif
    // fold the replicas of the camera-side accumulators (replica layout: rhs[n] U[Cf*36] gc[n])
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int i = tid + h * nt;
        if (i < 8 * n) {
            // U and gc are only accumulated on fresh iterations (K5 skips its first pass after a rejected step)
            const double v = (i < n || st.fresh) ? fold[h] : keep[h];
            if (i >= n && st.fresh && set == 0) b.Ukeep[i - n] = v;
            if (i < n) grs[i] = v;                         // rhs part
            else if (i < n + 6 * n) Us[i - n] = v;
            else gcs[i - 7 * n] = v;
        }
    }
    __syncthreads();
    for (int i = tid; i < n; i += nt) grs[i] += gcs[i];
    // Jacobi scaling of the camera blocks (first iteration) and the LM damping
    if (tid < n) {
        const double h = Us[(tid / 6) * 36 + (tid % 6) * 7];
        double sc = pre_sc;
        if (!st.have_scale) {
            sc = opt.jacobi ? 1.0 / (1.0 + sqrt(h)) : 1.0;
            if (set == 0) b.sc[tid] = sc;
        }
        const double s2 = sc * sc;
        lam[tid] = clampd(s2 * h, opt.dmin, opt.dmax) / (ba_set_radius(st, set) * s2);
    }
    BA_STAMP(b, 3);

    // (1) fresh linearisation: cost at x, gradient test
    const double fail_sum = wave_sum(pre_fail);          // only wave 0 holds real values
    if (st.fresh) {
        const double c = wave_sum(pre_cost);
        const double gslots = wave_max_nonneg(pre_gm);
        double gm = 0.0;
        for (int i = tid; i < n; i += nt) gm = fmax(gm, fabs(gcs[i]));
        gm = wave_max_nonneg(gm);
        if ((tid & 63) == 0) red[tid >> 6] = gm;
        __syncthreads();
        if (tid == 0) {
            st.x_cost = c;
            if (st.iter == 0) st.initial_cost = st.x_cost;
            double g = gslots;
            for (int w = 0; w < nt / 64; w++) g = fmax(g, red[w]);
            if (!isfinite(st.x_cost)) { st.done = 1; st.termination = RS_BA_FAILURE; }
            else if (g <= opt.gtol) { st.done = 1; st.termination = RS_BA_CONVERGENCE_GRADIENT; }
        }
        __syncthreads();
        if (st.done) { if (tid == 0 && set == 0) *b.st = st; publish(BA_HAND_TAKEN, 2ull); return; }      // every set reaches the same verdict
    } else {
        __syncthreads();
    }
    publish(BA_HAND_TAKEN, 0ull);      // the accumulators are in registers / LDS from here on: K8 may clear them
    if (tid == 0 && fail_sum > 0.0) s_fail = 1;          // K5 saw a non-finite landmark block
    BA_STAMP(b, 0);

    const int NB = n / 6;
    // (3) block L D L^T, one camera (6 columns) per step, two barriers per step.  The two roles run
    // DIFFERENT loops with the same barrier count (s_barrier only counts arrivals), so neither role's
    // registers are live in the other's code:
    //   phase 1 (after barrier A: block column J is final in LDS)
    //       chain: (a) factor + invert the 6x6 diagonal block (per lane, redundantly: no cross-lane traffic)
    //              (b) one lane per row: F_i = row_i L^-T D^-1 -> LDS (factor panel + the MFMA operand panels)
    //       tiles: rank-6 trailing update of step J-1 (operands in registers) on the live tiles, then
    //              publish block column J+1 RAW (it has the updates of steps <= J-1)
    //   phase 2 (after barrier B)
    //       chain: apply step J's update to the 6 entries of block column J+1 of its row (36 FMAs), so
    //              the next diagonal block never waits for the matrix cores
    //       tiles: load the MFMA operands of step J
#if K7_V2
    // (3) block L D L^T, one camera (6 columns) per step, ONE barrier per step.  Interval J (between barriers J - 1 and J):
    //   chain wave: raw block column J of its two rows per lane (published by the tile waves one interval earlier: it has
    //               the trailing updates of steps <= J - 2), minus step J - 1's rank-6 update with the panel values it still
    //               holds in registers -> final block column J; the six lanes of the diagonal block pass it through a
    //               scratch in LDS (the same wave reads it back: no barrier); L D L^T of the block per lane (redundantly);
    //               F_i = row_i L^-T D^-1 for both rows; F -> A (row-major factor panel), pivots -> dd[J & 1], and
    //               F d of the NEXT diagonal block's rows -> Pds for the next fix-up
    //   tile waves: rank-6 trailing update of step J - 1 with operands read from the factor panel A (columns of step J - 1,
    //               written one interval earlier) and dd[(J - 1) & 1], then publish block column J + 1 raw
    // Nothing is read in the interval in which it is written except by the wave that wrote it.
#if K7_V2 == 2
    // barrier-free form: the two chain waves may be a step apart, so the scratch blocks are three deep (step J uses J % 3)
#define K7_RING(J_) ((J_) % 3)
#else
#define K7_RING(J_) 0
#endif
    double* const Dg0 = Pd;                // [3][36] final diagonal block of a step (rows written by the lanes that own them)
    double* const Pds0 = Pd + 108;         // [3][36] Pds[e * 6 + k] = F[r0 + k][e] d_e of a step, read by the fix-up of the next
    double* const ddv = Pd + 216;          // [2][8] pivots of step J in ddv[(J & 1) * 8 + e]; entries 6, 7 stay zero
    volatile int* const dflag = (volatile int*)(Pd + 232);   // [2] dflag[w] = J + 1: chain wave w has written its rows of block J
    volatile int* const cflag = dflag + 2;                   // [2] cflag[w] = J + 1: chain wave w has finished step J (panel in A, pivots)
    volatile int* const tflag = dflag + 4;                   // [6] tflag[t] = J + 1: tile wave t has published block column J + 1
    // bounded spin on a word in LDS (the barrier-free form): a wave that gives up marks the solve failed and goes on
    auto wait_for = [&](volatile int* word, int want) {
        int spins = 0;
        while (*word < want) {
            if (++spins > (1 << 20)) { s_fail = 1; break; }
            __builtin_amdgcn_s_sleep(1);                   // (back-to-back polls of eight waves starve the LDS pipe the chain needs)
        }
    };
    if (chain) {
        BA_STAMP(b, 1);
        const int sl = wave >> 2;                          // this wave's rows: 64 sl + lane
        const int i = 64 * sl + lane, ic = min(i, n);
        double Fp[6];
#pragma unroll
        for (int e = 0; e < 6; e++) Fp[e] = 0.0;
#if RS_STAMPS
        const unsigned long long cpre_ = wall_clock64();
#endif
        __syncthreads();                                   // block column 0 published by the tile waves
#if RS_STAMPS
        if (tid == 0 && blockIdx.x == 0) b.dbg[22] += wall_clock64() - cpre_;      // wait for the tile waves' set-up
#endif
#if RS_STAMPS >= 2      // per-phase stamps: every one drains the wave's LDS queue (s_memrealtime returns through lgkmcnt), so the
                        // phases are indicative only; the coarse stamps (RS_STAMPS 1) do not disturb the loop
        unsigned long long cs_[6] = {0, 0, 0, 0, 0, 0}, ct_ = wall_clock64();
#define K7_CSTAMP(i) do { if (tid == 0 && blockIdx.x == 0) { const unsigned long long t__ = wall_clock64(); cs_[i] += t__ - ct_; ct_ = t__; } } while (0)
#else
#define K7_CSTAMP(i) do { } while (0)
#endif
        for (int J = 0; J < NB; J++) {
            const int c0 = 6 * J, r0 = c0 + 6;
            double* const Dg = Dg0 + 36 * K7_RING(J);
            double* const Pds = Pds0 + 36 * K7_RING(J);
            const double* const Pdp = Pds0 + 36 * K7_RING(J + 2);           // the step before (J - 1 = J + 2 mod 3)
            if (64 * sl + 63 < c0 || 64 * sl > n) {        // none of this wave's rows is left (or it never had any)
#if K7_V2 == 2
                if (lane == 0) cflag[sl] = NB + 1;         // nobody waits for this wave any more
                break;
#else
                __syncthreads();
                continue;
#endif
            }
#if K7_V2 == 2
            {
                // block column J is published (every tile wave has finished its step J - 1); the other chain wave is at most
                // one step behind (the scratch rings are three deep) and, where it owns rows of block J, has finished step J - 1
                // (its part of Pds)
                const int other = sl ^ 1;
#pragma unroll
                for (int t = 0; t < K7_NTW; t++) wait_for(tflag + t, J);
                const bool other_owns = (c0 >> 6) == other || ((c0 + 5) >> 6) == other;
                wait_for(cflag + other, other_owns ? J : J - 1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
#endif
            double x[6];
#pragma unroll
            for (int k = 0; k < 6; k++) x[k] = A[ic * LD + c0 + k];
            if (J > 0) {
                // step J - 1's update of block column J: A[i][c0 + k] -= sum_e F_i[e] d_e F[c0 + k][e]
                const double2* g2 = reinterpret_cast<const double2*>(Pdp);
#pragma unroll
                for (int e = 0; e < 6; e++) {
                    const double2 g01 = g2[3 * e], g23 = g2[3 * e + 1], g45 = g2[3 * e + 2];
                    x[0] -= Fp[e] * g01.x; x[1] -= Fp[e] * g01.y; x[2] -= Fp[e] * g23.x;
                    x[3] -= Fp[e] * g23.y; x[4] -= Fp[e] * g45.x; x[5] -= Fp[e] * g45.y;
                }
            }
            K7_CSTAMP(0);
            // the diagonal block: its rows belong to one chain wave or, where it straddles row 64, to both.  Owners write
            // their rows into the scratch and raise their word; a wave that needs rows of the other one polls that word
            // (both waves are in the same step: the barrier below separates the steps).
            {
                const int a = i - c0;
                const int o_lo = c0 >> 6, o_hi = (c0 + 5) >> 6;                    // owner waves of the first / last row
                if (a >= 0 && a < 6) {
#pragma unroll
                    for (int k = 0; k < 6; k++) Dg[a * 6 + k] = x[k];
                }
                if (sl == o_lo || sl == o_hi) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_s_waitcnt(0xc07f);                           // lgkmcnt(0): the rows are in LDS
                    if (lane == 0) dflag[sl] = J + 1;
                }
                const int other = sl ^ 1;
                if (other == o_lo || other == o_hi) {
                    while (dflag[other] < J + 1) { }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_s_waitcnt(0xc07f);
            }
            double Lm[6][6];
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
                for (int e = 0; e <= a; e++) Lm[a][e] = Dg[a * 6 + e];
            K7_CSTAMP(1);
            // (a) L D L^T of the diagonal block (every lane the same arithmetic on the same values)
            double dinv[6], dpiv[6];
            bool fbad = false;
#pragma unroll
            for (int c = 0; c < 6; c++) {
                const double piv = Lm[c][c];
                if (!(piv > 0.0) || !isfinite(piv)) fbad = true;
                const double rd = fast_rcp(piv);
                dinv[c] = rd;
                dpiv[c] = piv;
                double lc[6];
#pragma unroll
                for (int a = c + 1; a < 6; a++) lc[a] = Lm[a][c] * rd;              // l_ac; Lm[a][c] still holds l_ac * d_c
#pragma unroll
                for (int a = c + 1; a < 6; a++)
#pragma unroll
                    for (int e = c + 1; e <= a; e++) Lm[a][e] -= lc[a] * Lm[e][c];  // a_ae -= l_ac d_c l_ec
#pragma unroll
                for (int a = c + 1; a < 6; a++) Lm[a][c] = lc[a];
            }
            K7_CSTAMP(2);
            // (b) panel row: t = row L^-T by forward substitution (unit lower L), F = t D^-1.  A row of the diagonal block
            // itself comes out as its row of L (entries e < a; the rest is never read): the backward substitution reads the
            // unit-lower blocks from A.
            double F[6];
            {
                double t[6];
#pragma unroll
                for (int r = 0; r < 6; r++) {
                    double sacc = x[r];
#pragma unroll
                    for (int e = 0; e < r; e++) sacc -= t[e] * Lm[r][e];
                    t[r] = sacc;
                }
#pragma unroll
                for (int r = 0; r < 6; r++) F[r] = t[r] * dinv[r];
            }
            K7_CSTAMP(3);
            if (i >= c0 && i <= n) {
#pragma unroll
                for (int r = 0; r < 6; r++) A[i * LD + c0 + r] = F[r];
            }
            if (J + 1 < NB) {
                const int k0 = i - r0;                     // rows of the next diagonal block: their F d for the next fix-up
                if (k0 >= 0 && k0 < 6) {
#pragma unroll
                    for (int e = 0; e < 6; e++) Pds[e * 6 + k0] = F[e] * dpiv[e];
                }
                if (lane == 0 && sl == (c0 >> 6)) {
#pragma unroll
                    for (int e = 0; e < 6; e++) ddv[(J & 1) * 8 + e] = dpiv[e];
                }
            }
#pragma unroll
            for (int e = 0; e < 6; e++) Fp[e] = F[e];
            if (lane == 0 && fbad) s_fail = 1;
            K7_CSTAMP(4);
#if K7_V2 == 2
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_s_waitcnt(0xc07f);                            // lgkmcnt(0): panel, Pds and pivots are in LDS
            if (lane == 0) cflag[sl] = J + 1;
#else
            __syncthreads();                                               // barrier J
#endif
            K7_CSTAMP(5);
        }
#if K7_V2 == 2
        if (lane == 0 && cflag[sl] < NB) cflag[sl] = NB + 1;               // (left the loop after the last step)
#endif
#if RS_STAMPS >= 2
        if (tid == 0 && blockIdx.x == 0) { for (int q = 0; q < 6; q++) b.dbg[16 + q] += cs_[q]; }
#endif
#if RS_STAMPS
        if (tid == 0 && blockIdx.x == 0) { const unsigned long long t__ = wall_clock64(); b.dbg[30] += t__ - cpre_; b.dbg[23] = t__; b.dbg[31] += 1ull; }   // set-up wait + loop
#endif
    } else {
        // (2) the lower triangle + rhs row in the accumulator tiles: S (prefetched) + U + damping
        d4 acc[K7_TPW];
#pragma unroll
        for (int s = 0; s < K7_TPW; s++) {
            acc[s] = d4{0.0, 0.0, 0.0, 0.0};
            if (!tv[s]) continue;                       // wave-uniform
            const int k = 16 * tc[s] + lr;
            if (tr[s] >= tc[s] + 2 && 16 * tr[s] + 15 < n) {
                // a tile at least two tile rows below the diagonal and above the rhs row: |i - k| >= 17, so no
                // camera block and no diagonal entry falls into it — it is S alone (wave-uniform shortcut)
#pragma unroll
                for (int q = 0; q < 4; q++) acc[s][q] = (k < n) ? sv[s][q] : 0.0;
                continue;
            }
            const int kc = min(k, n - 1), kb = kc / 6;
            const double gv = grs[kc];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int i = 16 * tr[s] + lq + 4 * q;
                const int ic = min(i, n - 1);
                const double uv = Us[kb * 36 + (kc - 6 * kb) * 6 + ic % 6];
                double val = sv[s][q] + ((kb == ic / 6) ? uv : 0.0) + ((i == k) ? lam[ic] : 0.0);
                val = (k < n && i < n && k <= i) ? val : ((i == n && k < n) ? gv : 0.0);
                acc[s][q] = val;
            }
        }
        // loop-invariant publish addressing: LDS index of this lane's first entry of slot s and the
        // mask of its 4 rows that lie in the stored lower triangle
        int paddr[K7_TPW];
        unsigned pmask[K7_TPW];
#pragma unroll
        for (int s = 0; s < K7_TPW; s++) {
            const int k = 16 * tc[s] + lr;
            paddr[s] = (16 * tr[s] + lq) * LD + k;
            unsigned m = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int i = 16 * tr[s] + lq + 4 * q;
                m |= (i >= k && i <= n) ? (1u << q) : 0u;
            }
            pmask[s] = m;
        }
        const int LD4 = 4 * LD;
        // publish block column 0 (raw == final)
#pragma unroll
        for (int s = 0; s < K7_TPW; s++) {
            if (!tv[s] || tc[s] != 0) continue;
            if (lr >= 6) continue;
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (pmask[s] >> q & 1) A[paddr[s] + q * LD4] = acc[s][q];
        }
        // operand addressing: lane (lr, lq) feeds k = 4 kc + lq of the step's six columns (k = 6, 7: zero padding)
        int arow[K7_TPW], brow[K7_TPW];
#pragma unroll
        for (int s = 0; s < K7_TPW; s++) { arow[s] = min(16 * tr[s] + lr, n) * LD; brow[s] = min(16 * tc[s] + lr, n) * LD; }
        const int e1 = min(4 + lq, 5);                     // second K chunk: columns 4, 5 (lanes lq >= 2 are padding)
        const bool pad1 = lq >= 2;
        __syncthreads();
        for (int J = 0; J < NB; J++) {
            const int c0 = 6 * J, r0 = c0 + 6;
            if (J > 0) {
#if K7_V2 == 2
                wait_for(cflag + 0, J);                                    // both chain waves have finished step J - 1
                wait_for(cflag + 1, J);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
                // trailing update of step J - 1; live region: columns >= c0.  No row masks: rows that are already factored
                // only put garbage into accumulator entries that are never read again.
                const int cm = c0 - 6;
                const double* dv = ddv + ((J - 1) & 1) * 8;
                const double d0 = dv[lq], d1 = dv[4 + lq];             // d1 = 0 in the padding lanes
#pragma unroll
                for (int s = 0; s < K7_TPW; s++) {
                    if (!tv[s] || 16 * tc[s] + 15 < c0) continue;                  // wave-uniform
                    const double a0 = A[arow[s] + cm + lq], a1 = A[arow[s] + cm + e1];
                    const double b0 = A[brow[s] + cm + lq], b1 = A[brow[s] + cm + e1];
                    acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a0, b0 * d0, acc[s], 0, 0, 0);
                    acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(pad1 ? 0.0 : -a1, b1 * d1, acc[s], 0, 0, 0);
                }
            }
            // publish block column J + 1 raw (k in [r0, r0 + 6), rows i >= k) from the owning tiles
            if (J + 1 < NB) {
#pragma unroll
                for (int s = 0; s < K7_TPW; s++) {
                    if (!tv[s] || 16 * tc[s] + 15 < r0 || 16 * tc[s] >= r0 + 6) continue;   // wave-uniform
                    const int k = 16 * tc[s] + lr;
                    if (k < r0 || k >= r0 + 6) continue;
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        if (pmask[s] >> q & 1) A[paddr[s] + q * LD4] = acc[s][q];
                }
            }
#if K7_V2 == 2
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_s_waitcnt(0xc07f);                            // lgkmcnt(0): the column is in LDS
            if (lane == 0) tflag[tw] = J + 1;
#else
            __syncthreads();                                               // barrier J
#endif
        }
    }
#else
    if (chain) {
        BA_STAMP(b, 1);
        const int crow = cw * 64 + lane;           // row slot 0..127 (n + 1 - 6 <= 121 rows)
#if RS_STAMPS
        const unsigned long long cpre_ = wall_clock64();
#endif
        __syncthreads();                           // block column 0 published by the tile waves
#if RS_STAMPS
        if (tid == 0 && blockIdx.x == 0) b.dbg[22] += wall_clock64() - cpre_;
#endif
        for (int J = 0; J < NB; J++) {
            const int c0 = 6 * J, r0 = c0 + 6;
            const int irow = r0 + crow;
            const bool has_row = irow <= n;
            if (wave != 0 && r0 + 64 > n) {        // the second chain wave has no rows left
                __syncthreads();
                __syncthreads();
                continue;
            }
            double* row = A + min(irow, n) * LD + c0;
            // loads first: the diagonal block and this lane's panel row (clamped address, no branch)
            double L[6][6], rr[6];
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
                for (int e = 0; e <= a; e++) L[a][e] = A[(c0 + a) * LD + c0 + e];
#pragma unroll
            for (int e = 0; e < 6; e++) rr[e] = row[e];
            // (a) L D L^T of the diagonal block
            double dinv[6], dpiv[6], F[6];
            bool fbad = false;
#pragma unroll
            for (int c = 0; c < 6; c++) {
                const double piv = L[c][c];
                if (!(piv > 0.0) || !isfinite(piv)) fbad = true;
                const double rd = fast_rcp(piv);
                dinv[c] = rd;
                dpiv[c] = piv;
                double lc[6];
#pragma unroll
                for (int a = c + 1; a < 6; a++) lc[a] = L[a][c] * rd;              // l_ac; L[a][c] still holds l_ac * d_c
#pragma unroll
                for (int a = c + 1; a < 6; a++)
#pragma unroll
                    for (int e = c + 1; e <= a; e++) L[a][e] -= lc[a] * L[e][c];   // a_ae -= l_ac d_c l_ec
#pragma unroll
                for (int a = c + 1; a < 6; a++) L[a][c] = lc[a];
            }
            // (b) panel row: t = row L^-T by forward substitution (unit lower L), F = t D^-1
            {
                double t[6];
#pragma unroll
                for (int r = 0; r < 6; r++) {
                    double sacc = rr[r];
#pragma unroll
                    for (int e = 0; e < r; e++) sacc -= t[e] * L[r][e];
                    t[r] = sacc;
                }
#pragma unroll
                for (int r = 0; r < 6; r++) F[r] = t[r] * dinv[r];
            }
            if (has_row) {
#pragma unroll
                for (int r = 0; r < 6; r++) {
                    row[r] = F[r];
                    Nf[r * 128 + irow] = -F[r];
                    Pd[r * 128 + irow] = F[r] * dpiv[r];
                }
            }
            __syncthreads();                                               // barrier B
            if (J + 1 < NB) {
                // step J's update of block column J+1: A[i][r0+k] -= sum_e F_i[e] d_e F[r0+k][c0+e]
                double* nxt = row + 6;
                double x[6];
#pragma unroll
                for (int k = 0; k < 6; k++) x[k] = nxt[k];
#pragma unroll
                for (int e = 0; e < 6; e++) {
                    const double2* g2 = reinterpret_cast<const double2*>(Pd + e * 128 + r0);
                    const double2 g01 = g2[0], g23 = g2[1], g45 = g2[2];
                    x[0] -= F[e] * g01.x; x[1] -= F[e] * g01.y;
                    x[2] -= F[e] * g23.x; x[3] -= F[e] * g23.y;
                    x[4] -= F[e] * g45.x; x[5] -= F[e] * g45.y;
                }
                if (has_row) {
#pragma unroll
                    for (int k = 0; k < 6; k++) nxt[k] = x[k];
                }
            }
            if (tid == 63) {         // the unit-lower diagonal blocks, for the backward substitution
#pragma unroll
                for (int a = 1; a < 6; a++)
#pragma unroll
                    for (int e = 0; e < a; e++) Minv[J * 36 + a * 6 + e] = L[a][e];
                if (fbad) s_fail = 1;
            }
            __syncthreads();                                               // barrier A of step J+1
        }
#if RS_STAMPS
        if (tid == 0 && blockIdx.x == 0) { const unsigned long long t__ = wall_clock64(); b.dbg[30] += t__ - cpre_; b.dbg[23] = t__; b.dbg[31] += 1ull; }
#endif
    } else {
        // (2) the lower triangle + rhs row in the accumulator tiles: S (prefetched) + U + damping
        d4 acc[K7_TPW];
        double opA[K7_TPW][2], opB[K7_TPW][2];
#pragma unroll
        for (int s = 0; s < K7_TPW; s++) {
            acc[s] = d4{0.0, 0.0, 0.0, 0.0};
            opA[s][0] = opA[s][1] = opB[s][0] = opB[s][1] = 0.0;
            if (!tv[s]) continue;                       // wave-uniform
            const int k = 16 * tc[s] + lr;
            if (tr[s] >= tc[s] + 2 && 16 * tr[s] + 15 < n) {
                // a tile at least two tile rows below the diagonal and above the rhs row: |i - k| >= 17, so no
                // camera block and no diagonal entry falls into it — it is S alone (wave-uniform shortcut)
#pragma unroll
                for (int q = 0; q < 4; q++) acc[s][q] = (k < n) ? sv[s][q] : 0.0;
                continue;
            }
            const int kc = min(k, n - 1), kb = kc / 6;
            const double gv = grs[kc];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int i = 16 * tr[s] + lq + 4 * q;
                const int ic = min(i, n - 1);
                const double uv = Us[kb * 36 + (kc - 6 * kb) * 6 + ic % 6];
                double val = sv[s][q] + ((kb == ic / 6) ? uv : 0.0) + ((i == k) ? lam[ic] : 0.0);
                val = (k < n && i < n && k <= i) ? val : ((i == n && k < n) ? gv : 0.0);
                acc[s][q] = val;
            }
        }
        // publish block column 0 (raw == final)
#pragma unroll
        for (int s = 0; s < K7_TPW; s++) {
            if (!tv[s] || tc[s] != 0) continue;
            const int k = lr;
            if (k >= 6) continue;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int i = 16 * tr[s] + lq + 4 * q;
                if (i >= k && i <= n) A[i * LD + k] = acc[s][q];
            }
        }
        // loop-invariant publish addressing: LDS index of this lane's first entry of slot s and the
        // mask of its 4 rows that lie in the stored lower triangle
        int paddr[K7_TPW];
        unsigned pmask[K7_TPW];
#pragma unroll
        for (int s = 0; s < K7_TPW; s++) {
            const int k = 16 * tc[s] + lr;
            paddr[s] = (16 * tr[s] + lq) * LD + k;
            unsigned m = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int i = 16 * tr[s] + lq + 4 * q;
                m |= (i >= k && i <= n) ? (1u << q) : 0u;
            }
            pmask[s] = m;
        }
        const int LD4 = 4 * LD;
        __syncthreads();
        for (int J = 0; J < NB; J++) {
            const int c0 = 6 * J, r0 = c0 + 6;
            // phase 1: the odd slots' half of the trailing update of step J-1 (the even slots' half ran
            // in phase 2 of step J-1, while the chain did its fix-up); live region: rows/cols >= c0
            if (J > 0) {
#pragma unroll
                for (int s = 1; s < K7_TPW; s += 2) {
                    if (!tv[s] || 16 * tc[s] + 15 < c0) continue;                  // wave-uniform
                    acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(opA[s][0], opB[s][0], acc[s], 0, 0, 0);
                    acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(opA[s][1], opB[s][1], acc[s], 0, 0, 0);
                }
            }
            // publish block column J+1 raw (k in [r0, r0+6), rows i >= k) from the owning tiles
            if (J + 1 < NB) {
#pragma unroll
                for (int s = 0; s < K7_TPW; s++) {
                    if (!tv[s] || 16 * tc[s] + 15 < r0 || 16 * tc[s] >= r0 + 6) continue;   // wave-uniform
                    const int k = 16 * tc[s] + lr;
                    if (k < r0 || k >= r0 + 6) continue;
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        if (pmask[s] >> q & 1) A[paddr[s] + q * LD4] = acc[s][q];
                }
            }
            __syncthreads();                                               // barrier B
            if (J + 1 < NB) {
                // operands of step J for the matrix cores.  No masks: rows that are already factored only
                // put garbage into accumulator entries that are never read again.
#pragma unroll
                for (int s = 0; s < K7_TPW; s++) {
                    if (!tv[s] || 16 * tc[s] + 15 < r0) continue;                  // wave-uniform
#pragma unroll
                    for (int kc = 0; kc < 2; kc++) {
                        opA[s][kc] = Nf[(4 * kc + lq) * 128 + 16 * tr[s] + lr];
                        opB[s][kc] = Pd[(4 * kc + lq) * 128 + 16 * tc[s] + lr];
                    }
                }
                // phase 2: the even slots' half of the trailing update of step J
#pragma unroll
                for (int s = 0; s < K7_TPW; s += 2) {
                    if (!tv[s] || 16 * tc[s] + 15 < r0) continue;                  // wave-uniform
                    acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(opA[s][0], opB[s][0], acc[s], 0, 0, 0);
                    acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(opA[s][1], opB[s][1], acc[s], 0, 0, 0);
                }
            }
            __syncthreads();                                               // barrier A of step J+1
        }
    }
#endif
    __syncthreads();
    if (s_fail) {
        if (tid == 0) {
            if (set == 0) { st.solver_failed = 1; *b.st = st; }
            else b.set_out[set].solver_failed = 1;
        }
        publish(BA_HAND, 1ull);
        return;
    }
    // (4) backward substitution L^T x = y in ONE wave without block barriers: y (row n of the panels)
    // lives in registers, two entries per lane; per block column the 6 entries of y_J are broadcast with
    // v_readlane, every lane solves the 6x6 unit-lower block redundantly, and each lane updates its own
    // entries with the panel column loaded ahead of the dependency chain.
    if (wave == 0) {
        const double* yrow = A + (size_t)n * LD;
        double y0 = yrow[min(lane, n)], y1 = yrow[min(64 + lane, n)];
        const int i0 = min(lane, n), i1 = min(64 + lane, n);
        // One step.  Operands (the unit-lower diagonal block and this lane's two entries of the panel
        // column) do not depend on the chain: the caller loads them one step ahead into the other register
        // set (the loop is unrolled by two so that the sets never have to be copied).  The second entry
        // (rows 64..) is only touched while the block column lies beyond row 64.
#define K7_BS_LOAD(J_, L_, P0_, P1_)                                                                       \
    {                                                                                                      \
        const int jj_ = max((J_), 0), cc_ = 6 * jj_;                                                       \
        _Pragma("unroll") for (int e = 1; e < 6; e++)                                                      \
            _Pragma("unroll") for (int t = 0; t < e; t++) L_[e][t] = K7_V2 ? A[(cc_ + e) * LD + cc_ + t] : Minv[jj_ * 36 + e * 6 + t]; \
        _Pragma("unroll") for (int e = 0; e < 6; e++) P0_[e] = A[(cc_ + e) * LD + i0];                     \
        if (cc_ > 64) { _Pragma("unroll") for (int e = 0; e < 6; e++) P1_[e] = A[(cc_ + e) * LD + i1]; }   \
    }
#define K7_BS_STEP(J_, L_, P0_, P1_)                                                                       \
    {                                                                                                      \
        const int c0_ = 6 * (J_);                                                                          \
        double x[6];                                                                                       \
        _Pragma("unroll") for (int t = 5; t >= 0; t--) {                                                   \
            const int idx = c0_ + t;                                                                       \
            double sacc = readlane_f64(idx >= 64 ? y1 : y0, idx & 63);                                     \
            _Pragma("unroll") for (int e = 5; e > t; e--) sacc -= L_[e][t] * x[e];                         \
            x[t] = sacc;                                                                                   \
        }                                                                                                  \
        _Pragma("unroll") for (int e = 0; e < 6; e++) y0 -= P0_[e] * x[e];                                 \
        if (c0_ > 64) { _Pragma("unroll") for (int e = 0; e < 6; e++) y1 -= P1_[e] * x[e]; }               \
        if (lane == 0) { _Pragma("unroll") for (int t = 0; t < 6; t++) xs[c0_ + t] = x[t]; }               \
    }
        double La[6][6], Pa0[6], Pa1[6], Lb[6][6], Pb0[6], Pb1[6];
        K7_BS_LOAD(NB - 1, La, Pa0, Pa1);
        for (int J = NB - 1; J >= 0; J -= 2) {
            K7_BS_LOAD(J - 1, Lb, Pb0, Pb1);
            K7_BS_STEP(J, La, Pa0, Pa1);
            if (J - 1 >= 0) {
                K7_BS_LOAD(J - 2, La, Pa0, Pa1);
                K7_BS_STEP(J - 1, Lb, Pb0, Pb1);
            }
        }
#undef K7_BS_LOAD
#undef K7_BS_STEP
    }
    __syncthreads();
    BA_STAMP(b, 6);
#if RS_STAMPS
    if (tid == 0 && blockIdx.x == 0) { b.dbg[25] += wall_clock64() - b.dbg[23]; }      // end of the loop -> end of the backward substitution
#endif
    double* dc_set = b.dc + (size_t)set * BA_DC_STRIDE(n);
    if (HANDOFF) {
        // delta_c leaves first, written through, and the set's word is published: K8's workgroups derive the candidate
        // cameras and their blocks themselves (the same arithmetic as below) while this workgroup finishes its epilogue
        int badi = 0;
        for (int i = tid; i < n; i += nt) {
            const double dlt = -xs[i];
            if (!isfinite(dlt)) badi = 1;
            ba_store_sc1(&dc_set[i], dlt);
        }
        publish(BA_HAND, __syncthreads_or(badi) ? 1ull : 0ull);
    }
    // (5) delta_c = -x, candidate cameras, camera part of the step scalars
    double mcc = 0.0, ssq = 0.0, xsq = 0.0;
    bool bad = false;
    const double* Xc = b.Xc + (size_t)st.cur * d.C * 6;
    const int cand = (st.cur + 1 + set) % (b.ns + 1);       // this set's candidate buffer
    double* Xn = b.Xc + (size_t)cand * d.C * 6;
    for (int c = tid; c < d.C; c += nt) {
        const int s = b.slot[c];
        bool active = false;
        if (s >= 0)
            for (int k = 0; k < 6; k++) active = active || Us[s * 36 + k * 7] > 0.0;
        for (int k = 0; k < 6; k++) {
            const double x = Xc[6 * c + k];
            if (s >= 0) {
                const double dlt = -xs[6 * s + k];
                if (!isfinite(dlt)) bad = true;
                mcc += 0.5 * (dlt * dlt * lam[6 * s + k] - dlt * gcs[6 * s + k]);
                const double xn = x + dlt;
                if (active) { ssq += (x - xn) * (x - xn); xsq += x * x; }
                Xn[6 * c + k] = xn;
                if (!HANDOFF) dc_set[6 * s + k] = dlt;
            } else {
                Xn[6 * c + k] = x;
            }
        }
        cam_prepare(Xn + 6 * c, b.prep + ((size_t)cand * d.C + c) * BA_PREP);      // (for the next round's K5)
    }
    mcc = wave_sum(mcc); ssq = wave_sum(ssq); xsq = wave_sum(xsq);
    if (__any(bad) && (tid & 63) == 0) s_fail = 1;
    if ((tid & 63) == 0) { red3[tid >> 6][0] = mcc; red3[tid >> 6][1] = ssq; red3[tid >> 6][2] = xsq; }
    __syncthreads();
    if (tid == 0) {
        double a0 = 0, a1 = 0, a2 = 0;
        for (int w = 0; w < nt / 64; w++) { a0 += red3[w][0]; a1 += red3[w][1]; a2 += red3[w][2]; }
        if (set == 0) {
            st.cam_scal[0] = a0; st.cam_scal[1] = a1; st.cam_scal[2] = a2;
            st.solver_failed = s_fail;
            *b.st = st;
        } else {
            BaSetOut so;
            so.cam_scal[0] = a0; so.cam_scal[1] = a1; so.cam_scal[2] = a2; so.cam_scal[3] = 0.0;
            so.solver_failed = s_fail;
#pragma unroll
            for (int q = 0; q < 7; q++) so.pad[q] = 0;
            b.set_out[set] = so;
        }
    }
    BA_STAMP(b, 7);
    BA_STAMP_FLUSH(b, 0);
}

