// ba_init_body.h — K0: the state a bundle adjustment starts from (copies of the cameras / points into the state buffers,
// camera blocks, cleared accumulators, the initial state block).  Shared by ba_init (ba.hip) and by ba_init_count
// (ba_schur.hip: the same work as the first part of the landmark grouping's count launch).
#pragma once
#include "ba_common.h"
#include "ba_backsub_body.h"

static __device__ __forceinline__ void ba_init_body(const BaDims& d, const BaBufs& b, const BaOpt& opt, const double* __restrict__ cams_in,
                                                    const double* __restrict__ pts_in, unsigned long long free_mask, int from_mask,
                                                    uint8_t* __restrict__ cam_free, int32_t* __restrict__ zero_i32, int zero_n)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    if (from_mask && tid < d.C) {     // C <= 64: the free-camera table arrives as a kernel argument, not as two copies
        const int fr = (int)(free_mask >> tid & 1ull);
        b.slot[tid] = fr ? __popcll(free_mask & ((1ull << tid) - 1ull)) : -1;
        cam_free[tid] = (uint8_t)fr;
    }
    for (int i = tid; i < d.C * 6; i += nth) {
        const double v = cams_in[i];
        for (int q = 0; q <= b.ns; q++) b.Xc[(size_t)q * d.C * 6 + i] = v;
    }
    for (int i = tid; i < d.P * 3; i += nth) b.Xp[i] = pts_in[i];
    for (int c = tid; c < d.C; c += nth) cam_prepare(cams_in + 6 * c, b.prep + (size_t)c * BA_PREP);
    for (size_t i = tid; i < b.acc_count; i += nth) b.acc[i] = 0.0;
    for (int i = tid; i < 2 * b.ns * BA_NSLOT * BA_SLOT_STRIDE; i += nth) b.pt_scal[i] = 0.0;
    for (int i = tid; i < BA_NSLOT * BA_SLOT_STRIDE; i += nth) b.gmax[i] = 0.0;
    for (int i = tid; i < zero_n; i += nth) zero_i32[i] = 0;      // histogram of the landmark grouping
    if (tid == 0) {
        b.dbg[62] = 0ull;         // workgroups of ba_finalize that have finished (completion flag for the host)
        for (int k = BA_HAND; k <= BA_HAND_ERR; k++) b.dbg[k] = 0ull;     // K7 -> K8 hand-off words and their error counter (ba_backsub_body.h)
        b.dbg[BA_SDONE] = 0ull;
        b.dbg[37] = 0ull; b.dbg[38] = 0ull; b.dbg[26] = 0ull; b.dbg[27] = ~0ull; b.dbg[28] = 0ull; b.dbg[29] = ~0ull;
        BaState s;
        s.radius = opt.r0; s.decrease_factor = 2.0; s.x_cost = 0.0; s.initial_cost = 0.0;
        s.cam_scal[0] = s.cam_scal[1] = s.cam_scal[2] = s.cam_scal[3] = 0.0;
        s.iter = 0; s.successful = 0; s.invalid_steps = 0; s.done = 0;
        s.termination = 0; s.cur = 0; s.have_scale = 0; s.solver_failed = 0;
        s.fresh = 1; s.usable = 0; s.consec_accepts = 0; s.nact = 1;
        s.n_rounds = 0; s.n_fresh = 0; s.n_sets = 0; s.hand_lost = 0; s.calibrated = 0; s.pad_[0] = s.pad_[1] = s.pad_[2] = 0;
        b.st[1] = s;      // the state iteration 0 starts from (st[(0 + 1) & 1])
    }
}

