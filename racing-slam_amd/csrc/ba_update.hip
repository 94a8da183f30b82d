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
#include "ba_backsub_body.h"

__global__ __launch_bounds__(K8_THREADS) void ba_backsub_cost4(BaDims d, BaBufs b)
{
    ba_backsub_cost4_body<false>(d, b, (int)blockIdx.x, (int)blockIdx.y, 0, (size_t)blockIdx.y * gridDim.x + blockIdx.x, (size_t)gridDim.x * gridDim.y);
}
// batched: blockIdx.x = landmark block, blockIdx.y = speculative set, blockIdx.z = window
__global__ __launch_bounds__(K8_THREADS) void ba_backsub_cost4_batch(const BaWin* w, int it)
{
    const BaWin& x = w[blockIdx.z];
    const BaBufs b = ba_win_round(x, it, false);
    ba_backsub_cost4_body<false>(x.d, b, (int)blockIdx.x, (int)blockIdx.y, 0, (size_t)blockIdx.y * gridDim.x + blockIdx.x, (size_t)gridDim.x * gridDim.y);
}

size_t ba_backsub_lds_bytes(int C, int n)
{
    return sizeof(double) * (2 * (size_t)C * BA_PREP_LDS + (size_t)n + 8 + 7 * (size_t)C);      // (+ cameras and slot map of the fused form)
}

void ba_launch_backsub(hipStream_t s, const BaDims& d, const BaBufs& b)
{
    hipLaunchKernelGGL(ba_backsub_cost4, dim3(d.P > 0 ? (d.P + 63) / 64 : 1, b.ns), dim3(K8_THREADS), ba_backsub_lds_bytes(d.C, d.n), s, d, b);
}

void ba_launch_backsub_batch(hipStream_t s, const BaWin* d_wins, int B, int it, int ns, int max_P, size_t lds)
{
    if (lds > 48 * 1024) (void)rs_lds_attr((const void*)ba_backsub_cost4_batch, lds);
    hipLaunchKernelGGL(ba_backsub_cost4_batch, dim3((max_P + 63) / 64, ns, B), dim3(K8_THREADS), lds, s, d_wins, it);
}
