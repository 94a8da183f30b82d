// ba_solve.hip — K7: the reduced camera system of the local-window BA.
//
// Solves (U + Lambda_c - sum Y Y^T) y = g~ for the <= 21 free cameras of a
// local window (n = 6 Cf <= 126) in ONE workgroup.  This is the step Ceres
// hands to a sparse Cholesky after Schur elimination (reference
// src/Optimization.cpp:360, SPARSE_SCHUR); at n ~ 108 the matrix is dense and
// tiny, and the solve is a latency problem, not a bandwidth problem.
//
// Layout.  The lower triangle — plus the right-hand side as an extra ROW n, so
// that the forward substitution falls out of the factorisation — lives in
// REGISTERS in the accumulator layout of v_mfma_f64_16x16x4_f64: the
// (n+1) x (n+1) matrix is cut into 16x16 tiles, the 36 lower-triangle tiles
// are dealt to six TILE waves.  LDS carries the published block columns, the
// factor panels and the k-major MFMA operand panels.
//
// f64 MFMA and f64 VALU share one datapath per SIMD (tools/microbench/overlap.hip),
// and waves w, w+4 of a workgroup share a SIMD, so the work is split by ROLE:
//   chain waves (0, 4)   right-looking block L D L^T, one camera (6 columns) per step:
//                        6x6 diagonal block per lane (v_rcp_f64 + Newton, no sqrt),
//                        panel row by forward substitution, rank-6 fix-up of the next
//                        block column on the VALU — never waits for the matrix cores
//   tile waves (others)  rank-6 trailing update, 2 MFMAs per live tile (K = 6 padded
//                        to 8), publish of the next block column
// two barriers per step; the roles run different loops with equal barrier counts.
// Then a backward substitution in one wave with y in registers, and the camera
// step / candidate cameras / step scalars.  Measurements: DESIGN.md 4.2.
#include "ba_common.h"
#include "ba_backsub_body.h"
#include "ba_solve_body.h"

__global__ __launch_bounds__(K7_THREADS) void ba_reduced_solve_lds(BaDims d, BaBufs b, BaOpt opt) { ba_reduced_solve_lds_body<false>(d, b, opt); }

// K7 + K8 in ONE launch (single local window, vision only, one rank): workgroups [0, ns) are K7, one per speculative
// set; the others are K8's, `nblk` landmark blocks of K7_THREADS / 4 landmarks for each of `rs` = min(ns, BA_CALIBRATED_SETS)
// sets, and wait for their set's hand-off word; in a round deeper than `rs` a K8 workgroup evaluates set s + rs after set s.  A K8 launch behind K7 costs the launch gap plus K8's own prologue (state, landmark records, Jacobi
// scale, V^-1, the first observations: two dependent round trips) AFTER the solve; here all of that is in flight
// while K7 is solving, and what remains behind the hand-off is one round trip for delta_c / the candidate's camera
// blocks and the arithmetic.  Producers never wait for consumers and are dispatched first, and a consumer's wait
// is bounded (ba_hand_wait), so the grid always drains.
__global__ __launch_bounds__(K7_THREADS) void ba_solve_backsub(BaDims d, BaBufs b, BaOpt opt, int nblk)
{
    if ((int)blockIdx.x < b.ns) {
        ba_reduced_solve_lds_body<true>(d, b, opt);
    } else {
        const int v = (int)blockIdx.x - b.ns, rs = ((int)gridDim.x - b.ns) / nblk;
        ba_backsub_cost4_body<true>(d, b, v % nblk, v / nblk, rs, (size_t)v, (size_t)gridDim.x - b.ns);
    }
}
// batched: blockIdx.x = speculative set, blockIdx.z = window
__global__ __launch_bounds__(K7_THREADS) void ba_reduced_solve_lds_batch(const BaWin* w, BaOpt opt, int it)
{
    const BaWin& x = w[blockIdx.z];
    const BaBufs b = ba_win_round(x, it, false);
    ba_reduced_solve_lds_body<false>(x.d, b, opt);
}

size_t ba_reduced_solve_lds_bytes(int n)
{
    const int LD = n + 1 + ((n & 1) ? 1 : 0);
    return sizeof(double) * ((size_t)(n + 1) * LD + 2 * (size_t)n + 6 * (size_t)n + 6 * (size_t)n + 3 * (size_t)n + 8 + 2 * 8 * 128);
}

// workgroups of the fused launch, all resident at once: the K7 workgroups of every set (producers: dispatched first, they
// never wait) and the K8 workgroups of `rs` sets: as many of the sets a calibrated round evaluates (<= 3) as the chip holds
// beside them (3 up to 10.7 k landmarks, 2 up to 16 k, 1 up to 32 k).  In a deeper round the K8 workgroups of set s go on to
// sets s + rs, s + 2 rs, ..., whose hand-off is published by then.
static inline int ba_backsub_resident_sets(const BaDims& d, const BaBufs& b, int n_cu)
{
    const int per = K7_THREADS / 4, nblk = (d.P + per - 1) / per;
    int rs = b.ns < BA_CALIBRATED_SETS ? b.ns : BA_CALIBRATED_SETS;
    while (rs > 1 && b.ns + rs * nblk > n_cu) rs--;
    return rs;
}
int ba_solve_backsub_workgroups(const BaDims& d, const BaBufs& b, int n_cu)
{
    const int per = K7_THREADS / 4;
    return b.ns + ba_backsub_resident_sets(d, b, n_cu) * ((d.P + per - 1) / per);
}

void ba_launch_solve_backsub(hipStream_t s, const BaDims& d, const BaBufs& b, const BaOpt& opt, int n_cu)
{
    const size_t lds = max(ba_reduced_solve_lds_bytes(d.n), ba_backsub_lds_bytes(d.C, d.n));
    (void)rs_lds_attr((const void*)ba_solve_backsub, lds);
    const int per = K7_THREADS / 4, nblk = (d.P + per - 1) / per;
    hipLaunchKernelGGL(ba_solve_backsub, dim3(b.ns + nblk * ba_backsub_resident_sets(d, b, n_cu)), dim3(K7_THREADS), lds, s, d, b, opt, nblk);
}

void ba_launch_reduced_solve_lds(hipStream_t s, const BaDims& d, const BaBufs& b, const BaOpt& opt)
{
    const size_t lds = ba_reduced_solve_lds_bytes(d.n);
    hipLaunchKernelGGL(ba_reduced_solve_lds, dim3(b.ns), dim3(K7_THREADS), lds, s, d, b, opt);
}

int ba_prepare_reduced_solve_lds(int n)
{
    const size_t lds = ba_reduced_solve_lds_bytes(n);
    return (int)rs_lds_attr((const void*)ba_reduced_solve_lds, lds);
}

int ba_prepare_reduced_solve_lds_batch(int max_n)
{
    return (int)rs_lds_attr((const void*)ba_reduced_solve_lds_batch, ba_reduced_solve_lds_bytes(max_n));
}

void ba_launch_reduced_solve_lds_batch(hipStream_t s, const BaWin* d_wins, int B, const BaOpt& opt, int it, int ns, int max_n)
{
    hipLaunchKernelGGL(ba_reduced_solve_lds_batch, dim3(ns, 1, B), dim3(K7_THREADS), ba_reduced_solve_lds_bytes(max_n), s, d_wins, opt, it);
}
