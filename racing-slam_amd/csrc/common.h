// common.h — internal declarations shared by the HIP translation units of librsgpu.so.
// gfx950 (MI355X) only: 64-lane wavefronts, no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/rsgpu.h"

#define RS_WAVE 64

struct rs_prof_slot {
    char name[32];
    int launches;
    std::vector<hipEvent_t> ev;   // pairs (start, stop)
};

// grow-only bump arena (device or pinned host memory); slabs that became too small are retired and freed at the
// next reset, after the stream has drained
struct rs_arena {
    char* base = nullptr;
    size_t cap = 0, used = 0;
    std::vector<void*> retired;
};
struct rs_pending_download { const void* pinned; void* user; size_t bytes; };

struct rs_context {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // grow-only device workspace (never freed inside an enqueue path)
    void* ws = nullptr;
    size_t ws_bytes = 0;
    // "known zero" region of the workspace (the landmark grouping's histogram / cursors of the last bundle adjustment, left
    // at zero by its finalize kernel) and the highest workspace byte any OTHER caller may have written since (rs_workspace)
    int32_t* grp_zero_ptr = nullptr;
    int grp_zero_n = 0;
    size_t ws_dirty_hi = 0;
    // pinned host scratch for small result read-back
    void* pinned = nullptr;
    size_t pinned_bytes = 0;
    // staging pool of the boundary (rs_stage_*): inputs flattened by a shim travel host -> pinned -> device with
    // asynchronous copies on the context stream; outputs come back the same way
    rs_arena stage_dev, stage_pin;
    std::vector<rs_pending_download> stage_down;
    // profiling
    bool prof_on = false;
    std::vector<rs_prof_slot> prof;
    // RCCL (loaded with dlopen; see comm.hip)
    void* comm = nullptr;
    int n_ranks = 1;
    int rank = 0;
    // in-process group of contexts (rs_comm_init_local): the same exchange step without RCCL
    struct rs_local_group* local = nullptr;
    // cached BA graph / buffers live in ba.hip
    void* ba_cache = nullptr;
    // per-iteration record of the last rs_bundle_adjust (points into the pinned block; rs_ba_get_trace)
    const void* ba_trace = nullptr;
    int ba_trace_n = 0;
    int ba_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // rs_ba_get_stats
    int ba_item = 0;                    // landmarks per item of a single solve: 0 = by window size (40 / 64), or 32 / 40 / 64
    int ba_batch_item = 0;              // landmarks per item of rs_bundle_adjust_batch's grid mode: 0 = default (32), or 32 / 40 / 64
    int ba_band_mode = 0;               // blocked reduced solve: 0 = the banded factorisation (two-sided) when S is block-banded, 1 = never, 2 = banded in one workgroup
    int ba_s_replicas = 0;              // replicas of S on the local-window path (0 = library default; "ba_s_replicas")
    int ba_handoff_timeout_us = 4000;   // fused K7 + K8 launch: how long a K8 workgroup waits for its hand-off word ("ba_handoff_timeout_us")
    const double* ba_cams = nullptr;    // cameras after the last rs_bundle_adjust, mirrored in the pinned block
    int ba_cams_n = 0;
    // speculative trust-region radii per BA round (0 = library default; rs_context_set_int "ba_speculative_sets")
    int ba_sets = 0;
    int n_cu = 256;                     // compute units of the device
    int ba_fuse_mode = 0;               // local-window BA: 0 = K7 + K8 in one launch when this is the only solve in flight, 1 = never,
                                        // 2 = K7 + K8 in one launch wherever possible, 3 = the whole round (K5 + K7 + K8) wherever possible
    int k2_mode = 0;                    // rs_reproj_match: 0 = eight lanes per map point where the KD-tree fits in LDS, 1 = always one lane per point
    int ba_imu_mode = 0;                // inertial solves: 0 = z blocks eliminated around the LDS K7 where possible, 1 = always the N x N blocked solve
    int ba_batch_mode = 0;              // rs_bundle_adjust_batch: 0 = one grid for all windows where possible, 1 = lanes only
    // child contexts of rs_bundle_adjust_batch (own stream / workspace each), created on first use
    std::vector<rs_context*> batch_lanes;
    std::vector<hipStream_t> batch_streams;
    // proposal table of rs_reproj_match: persistent, always left at all-ones by the accept tail
    unsigned long long* prop = nullptr;
    size_t prop_cap = 0;
    hipEvent_t sync_ev = nullptr;       // rs_context_wait_for: "everything enqueued on this context so far"
    void* tri_pin = nullptr;            // pinned in / out block + completion flag of rs_triangulate_host's one-launch path
    int tri_ticket = 0;
    void* k1_top = nullptr;             // [batch][nq] {best, second} packed keys of rs_match_descriptors; all-ones between calls
    size_t k1_top_cap = 0;
};

int rs_fail(rs_context* ctx, int code, const char* fmt, ...);

#define RS_HIP(ctx, call)                                                                  \
    do {                                                                                   \
        hipError_t e__ = (call);                                                           \
        if (e__ != hipSuccess)                                                             \
            return rs_fail((ctx), RS_ERR_HIP, "%s failed: %s (%s:%d)", #call,              \
                           hipGetErrorString(e__), __FILE__, __LINE__);                    \
    } while (0)

// workspace: returns a device pointer with at least `bytes` bytes, 256-B aligned.
int rs_workspace(rs_context* ctx, size_t bytes, void** out);
int rs_workspace_quiet(rs_context* ctx, size_t bytes, void** out);   // does not mark the bytes as possibly written (ba.hip keeps its own account)
int rs_pinned(rs_context* ctx, size_t bytes, void** out);
// hipFuncSetAttribute(fn, MaxDynamicSharedMemorySize, bytes), but only when `bytes` exceeds what was already set for
// `fn` in this process: the attribute is sticky and the driver call costs microseconds of host time per launch.
hipError_t rs_lds_attr(const void* fn, size_t bytes);


// profiling brackets around a kernel launch
void rs_prof_start(rs_context* ctx, const char* name);
void rs_prof_stop(rs_context* ctx, const char* name);

struct rs_prof_scope {
    rs_context* c;
    const char* n;
    rs_prof_scope(rs_context* ctx, const char* name) : c(ctx), n(name) { if (c->prof_on) rs_prof_start(c, n); }
    ~rs_prof_scope() { if (c->prof_on) rs_prof_stop(c, n); }
};

// Exclusive prefix sum of one int per thread over the workgroup (any size up to 1024, multiple of 64);
// returns this thread's offset and the workgroup total.
__device__ __forceinline__ int rs_block_exclusive_scan(int v, int* total)
{
    __shared__ int rs_scan_w[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    int x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(x, off, 64);
        if (lane >= off) x += t;
    }
    if (lane == 63) rs_scan_w[wave] = x;
    __syncthreads();
    int pre = 0, tot = 0;
    for (int w = 0; w < nw; w++) { const int c = rs_scan_w[w]; if (w < wave) pre += c; tot += c; }
    __syncthreads();
    *total = tot;
    return pre + x - v;
}

// sum all-reduce of f64 on the context stream over the attached communicator (RCCL or the in-process group);
// no-op without one
int rs_allreduce_f64(rs_context* ctx, double* d_buf, size_t count, bool is_max);
int rs_allreduce_min_u64(rs_context* ctx, unsigned long long* d_buf, size_t count);   // element-wise minimum of unsigned 64-bit keys
static inline bool rs_comm_active(const rs_context* ctx) { return ctx->comm != nullptr || ctx->local != nullptr; }

