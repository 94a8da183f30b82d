// context.hip — context lifetime, workspace, HIP-event profiling and the RCCL
// communicator of librsgpu.so.
#include <dlfcn.h>
#include <stdarg.h>
#include <stdlib.h>

#include "common.h"

int rs_fail(rs_context* ctx, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    else fprintf(stderr, "rsgpu: %s\n", buf);
    return code;
}

extern "C" int rs_abi_version(void) { return RSGPU_ABI_VERSION; }

extern "C" int rs_context_create(int device_id, rs_context** out)
{
    if (!out) return RS_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        fprintf(stderr, "rsgpu: no HIP device visible; librsgpu has no CPU fallback\n");
        return RS_ERR_NO_DEVICE;
    }
    if (device_id < 0 || device_id >= n) return RS_ERR_INVALID;
    if (hipSetDevice(device_id) != hipSuccess) return RS_ERR_HIP;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return RS_ERR_HIP;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fprintf(stderr, "rsgpu: device %d is %s; this library is built for gfx950 only\n", device_id,
                prop.gcnArchName);
        return RS_ERR_NO_DEVICE;
    }
    rs_context* c = new rs_context();
    c->device = device_id;
    c->n_cu = prop.multiProcessorCount;
    if (const char* e = getenv("RS_BA_SETS")) { const int v = atoi(e); if (v >= 1 && v <= 3) c->ba_sets = v; }
    *out = c;
    return RS_OK;
}

void rs_ba_cache_free(rs_context* ctx);
static void arena_free(rs_arena& a, bool pinned);

extern "C" int rs_context_destroy(rs_context* ctx)
{
    if (!ctx) return RS_OK;
    (void)hipSetDevice(ctx->device);
    rs_comm_destroy(ctx);
    rs_ba_cache_free(ctx);
    for (rs_context* lane : ctx->batch_lanes) rs_context_destroy(lane);
    for (hipStream_t st : ctx->batch_streams) (void)hipStreamDestroy(st);
    ctx->batch_lanes.clear();
    ctx->batch_streams.clear();
    arena_free(ctx->stage_dev, false);
    arena_free(ctx->stage_pin, true);
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->prop) (void)hipFree(ctx->prop);
    if (ctx->k1_top) (void)hipFree(ctx->k1_top);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->tri_pin) (void)hipHostFree(ctx->tri_pin);
    if (ctx->sync_ev) (void)hipEventDestroy(ctx->sync_ev);
    for (auto& s : ctx->prof)
        for (auto e : s.ev) (void)hipEventDestroy(e);
    delete ctx;
    return RS_OK;
}

extern "C" int rs_context_set_stream(rs_context* ctx, void* s)
{
    if (!ctx) return RS_ERR_INVALID;
    ctx->stream = (hipStream_t)s;
    return RS_OK;
}

extern "C" int rs_context_wait_for(rs_context* ctx, rs_context* const* others, int n)
{
    if (!ctx || n < 0 || (n > 0 && !others)) return rs_fail(ctx, RS_ERR_INVALID, "rs_context_wait_for: bad arguments");
    for (int i = 0; i < n; i++) {
        rs_context* o = others[i];
        if (!o) return rs_fail(ctx, RS_ERR_INVALID, "rs_context_wait_for: null context");
        if (o == ctx || o->stream == ctx->stream) continue;             // same stream: ordered already
        if (!o->sync_ev) {
            RS_HIP(ctx, hipSetDevice(o->device));
            RS_HIP(ctx, hipEventCreateWithFlags(&o->sync_ev, hipEventDisableTiming));
        }
        RS_HIP(ctx, hipEventRecord(o->sync_ev, o->stream));
        RS_HIP(ctx, hipStreamWaitEvent(ctx->stream, o->sync_ev, 0));
    }
    return RS_OK;
}

extern "C" int rs_context_fork(rs_context* ctx, rs_context* const* others, int n)
{
    if (!ctx || n < 0 || (n > 0 && !others)) return rs_fail(ctx, RS_ERR_INVALID, "rs_context_fork: bad arguments");
    bool recorded = false;
    for (int i = 0; i < n; i++) {
        rs_context* o = others[i];
        if (!o) return rs_fail(ctx, RS_ERR_INVALID, "rs_context_fork: null context");
        if (o == ctx || o->stream == ctx->stream) continue;
        if (!recorded) {
            if (!ctx->sync_ev) {
                RS_HIP(ctx, hipSetDevice(ctx->device));
                RS_HIP(ctx, hipEventCreateWithFlags(&ctx->sync_ev, hipEventDisableTiming));
            }
            RS_HIP(ctx, hipEventRecord(ctx->sync_ev, ctx->stream));
            recorded = true;
        }
        RS_HIP(ctx, hipStreamWaitEvent(o->stream, ctx->sync_ev, 0));
    }
    return RS_OK;
}

#define BA_MAXSETS_KNOB 5        /* = BA_MAXSETS (ba_common.h) */
extern "C" int rs_context_set_int(rs_context* ctx, const char* name, int value)
{
    if (!ctx || !name) return RS_ERR_INVALID;
    if (strcmp(name, "ba_speculative_sets") == 0) {
        if (value < 0 || value > BA_MAXSETS_KNOB) return rs_fail(ctx, RS_ERR_INVALID, "ba_speculative_sets must be 0 (default) .. 5");
        ctx->ba_sets = value;
        return RS_OK;
    }
    if (strcmp(name, "ba_imu_mode") == 0) {
        if (value < 0 || value > 1) return rs_fail(ctx, RS_ERR_INVALID, "ba_imu_mode must be 0 (eliminate around the LDS solve) or 1 (blocked N x N solve)");
        ctx->ba_imu_mode = value;
        return RS_OK;
    }
    if (strcmp(name, "ba_fuse_mode") == 0) {
        if (value < 0 || value > 3) return rs_fail(ctx, RS_ERR_INVALID, "ba_fuse_mode must be 0 (solve + back-substitution in one launch when no other solve of the process is in flight), 1 (separate launches), 2 (solve + back-substitution in one launch wherever possible) or 3 (the whole round in one launch wherever possible)");
        ctx->ba_fuse_mode = value;
        return RS_OK;
    }
    if (strcmp(name, "ba_item_landmarks") == 0) {
        if (value != 0 && (value < 32 || value > 64 || value % 8)) return rs_fail(ctx, RS_ERR_INVALID, "ba_item_landmarks must be 0 (default) or 32, 40, 48, 56, 64");
        ctx->ba_item = value;
        return RS_OK;
    }
    if (strcmp(name, "ba_batch_item_landmarks") == 0) {
        if (value != 0 && (value < 32 || value > 64 || value % 8)) return rs_fail(ctx, RS_ERR_INVALID, "ba_batch_item_landmarks must be 0 (default) or 32, 40, 48, 56, 64");
        ctx->ba_batch_item = value;
        return RS_OK;
    }
    if (strcmp(name, "ba_band_mode") == 0) {
        if (value < 0 || value > 2) return rs_fail(ctx, RS_ERR_INVALID, "ba_band_mode must be 0 (banded factorisation where the reduced matrix is block-banded), 1 (always the general blocked one) or 2 (banded, one workgroup instead of two sides)");
        ctx->ba_band_mode = value;
        return RS_OK;
    }
    if (strcmp(name, "ba_s_replicas") == 0) {
        if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8) return rs_fail(ctx, RS_ERR_INVALID, "ba_s_replicas must be 0 (default), 1, 2, 4 or 8");
        ctx->ba_s_replicas = value;
        return RS_OK;
    }
    if (strcmp(name, "ba_handoff_timeout_us") == 0) {
        if (value < 1 || value > 1000000) return rs_fail(ctx, RS_ERR_INVALID, "ba_handoff_timeout_us must be 1 .. 1000000");
        ctx->ba_handoff_timeout_us = value;
        return RS_OK;
    }
    if (strcmp(name, "k2_mode") == 0) {
        if (value < 0 || value > 1) return rs_fail(ctx, RS_ERR_INVALID, "k2_mode must be 0 (eight lanes per map point where possible) or 1 (one lane per point)");
        ctx->k2_mode = value;
        return RS_OK;
    }
    if (strcmp(name, "ba_batch_mode") == 0) {
        if (value < 0 || value > 1) return rs_fail(ctx, RS_ERR_INVALID, "ba_batch_mode must be 0 (grid where possible) or 1 (lanes)");
        ctx->ba_batch_mode = value;
        return RS_OK;
    }
    return rs_fail(ctx, RS_ERR_INVALID, "unknown option %s", name);
}

extern "C" int rs_context_synchronize(rs_context* ctx)
{
    if (!ctx) return RS_ERR_INVALID;
    RS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RS_OK;
}

extern "C" const char* rs_last_error(const rs_context* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int rs_workspace(rs_context* ctx, size_t bytes, void** out)
{
    if (bytes > ctx->ws_dirty_hi) ctx->ws_dirty_hi = bytes;      // the caller may write [0, bytes)
    return rs_workspace_quiet(ctx, bytes, out);
}

int rs_workspace_quiet(rs_context* ctx, size_t bytes, void** out)
{
    if (bytes > ctx->ws_bytes) {
        // growing is a synchronising event (first call / larger problem only)
        RS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->ws) RS_HIP(ctx, hipFree(ctx->ws));
        ctx->ws = nullptr;
        ctx->grp_zero_ptr = nullptr;
        ctx->ws_bytes = 0;
        size_t want = (bytes + (1u << 20)) & ~((size_t)(1u << 20) - 1);
        if (hipMalloc(&ctx->ws, want) != hipSuccess) return rs_fail(ctx, RS_ERR_NOMEM, "workspace of %zu bytes", want);
        ctx->ws_bytes = want;
    }
    *out = ctx->ws;
    return RS_OK;
}

int rs_pinned(rs_context* ctx, size_t bytes, void** out)
{
    if (bytes > ctx->pinned_bytes) {
        if (ctx->pinned) RS_HIP(ctx, hipHostFree(ctx->pinned));
        ctx->pinned = nullptr;
        ctx->ba_trace = nullptr;
        ctx->ba_trace_n = 0;
        ctx->ba_cams = nullptr;
        ctx->ba_cams_n = 0;
        size_t want = bytes < 4096 ? 4096 : bytes;
        if (hipHostMalloc(&ctx->pinned, want, hipHostMallocDefault) != hipSuccess)
            return rs_fail(ctx, RS_ERR_NOMEM, "pinned buffer of %zu bytes", want);
        ctx->pinned_bytes = want;
    }
    *out = ctx->pinned;
    return RS_OK;
}

// ------------------------------------------------------------- staging pool
// What a drop-in shim needs around every call: upload a handful of freshly flattened host arrays, get device scratch
// for the outputs, read a few results back.  hipMalloc / hipFree per array (the first version of the shims) costs
// tens of microseconds each and synchronises the device; here both sides are bump allocators over grow-only slabs
// and every copy is asynchronous on the context stream.
static int arena_take(rs_context* ctx, rs_arena& a, bool pinned, size_t bytes, void** out)
{
    const size_t need = (bytes + 255) & ~(size_t)255;
    if (a.used + need > a.cap) {
        size_t want = a.cap ? a.cap * 2 : (size_t)1 << 20;
        while (want < a.used + need) want *= 2;
        void* p = nullptr;
        const hipError_t e = pinned ? hipHostMalloc(&p, want, hipHostMallocDefault) : hipMalloc(&p, want);
        if (e != hipSuccess) return rs_fail(ctx, RS_ERR_NOMEM, "staging slab of %zu bytes", want);
        if (a.base) a.retired.push_back(a.base);      // earlier takes of this group stay valid until the reset
        a.base = (char*)p;
        a.cap = want;
        a.used = 0;
    }
    *out = a.base + a.used;
    a.used += need;
    return RS_OK;
}

static void arena_reset(rs_arena& a, bool pinned)
{
    for (void* p : a.retired) { if (pinned) (void)hipHostFree(p); else (void)hipFree(p); }
    a.retired.clear();
    a.used = 0;
}

static void arena_free(rs_arena& a, bool pinned)
{
    arena_reset(a, pinned);
    if (a.base) { if (pinned) (void)hipHostFree(a.base); else (void)hipFree(a.base); }
    a.base = nullptr;
    a.cap = 0;
}

extern "C" int rs_stage_begin(rs_context* ctx)
{
    if (!ctx) return RS_ERR_INVALID;
    RS_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->stage_dev.used || ctx->stage_pin.used || !ctx->stage_down.empty())
        RS_HIP(ctx, hipStreamSynchronize(ctx->stream));      // copies of the previous group may still be in flight
    ctx->stage_down.clear();
    arena_reset(ctx->stage_dev, false);
    arena_reset(ctx->stage_pin, true);
    return RS_OK;
}

extern "C" int rs_stage_alloc(rs_context* ctx, size_t bytes, void** d_out)
{
    if (!ctx || !d_out) return RS_ERR_INVALID;
    return arena_take(ctx, ctx->stage_dev, false, bytes ? bytes : 1, d_out);
}

extern "C" int rs_stage_upload(rs_context* ctx, const void* h_src, size_t bytes, void** d_out)
{
    if (!ctx || !d_out || (bytes && !h_src)) return RS_ERR_INVALID;
    int rc = arena_take(ctx, ctx->stage_dev, false, bytes ? bytes : 1, d_out);
    if (rc || !bytes) return rc;
    void* pin = nullptr;
    rc = arena_take(ctx, ctx->stage_pin, true, bytes, &pin);
    if (rc) return rc;
    memcpy(pin, h_src, bytes);
    RS_HIP(ctx, hipMemcpyAsync(*d_out, pin, bytes, hipMemcpyHostToDevice, ctx->stream));
    return RS_OK;
}

extern "C" int rs_stage_download(rs_context* ctx, const void* d_src, size_t bytes, void* h_dst)
{
    if (!ctx || (bytes && (!d_src || !h_dst))) return RS_ERR_INVALID;
    if (!bytes) return RS_OK;
    void* pin = nullptr;
    const int rc = arena_take(ctx, ctx->stage_pin, true, bytes, &pin);
    if (rc) return rc;
    RS_HIP(ctx, hipMemcpyAsync(pin, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    ctx->stage_down.push_back({pin, h_dst, bytes});
    return RS_OK;
}

extern "C" int rs_stage_sync(rs_context* ctx)
{
    if (!ctx) return RS_ERR_INVALID;
    RS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (const auto& d : ctx->stage_down) memcpy(d.user, d.pinned, d.bytes);
    ctx->stage_down.clear();
    return RS_OK;
}

// ------------------------------------------------------------------ profiling
static rs_prof_slot* prof_slot(rs_context* ctx, const char* name)
{
    for (auto& s : ctx->prof)
        if (strcmp(s.name, name) == 0) return &s;
    if (ctx->prof.size() >= RS_PROF_MAX) return nullptr;
    rs_prof_slot s;
    memset(s.name, 0, sizeof s.name);
    strncpy(s.name, name, sizeof s.name - 1);
    s.launches = 0;
    ctx->prof.push_back(s);
    return &ctx->prof.back();
}

void rs_prof_start(rs_context* ctx, const char* name)
{
    rs_prof_slot* s = prof_slot(ctx, name);
    if (!s) return;
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
    s->ev.push_back(a);
    s->ev.push_back(b);
    (void)hipEventRecord(a, ctx->stream);
}

void rs_prof_stop(rs_context* ctx, const char* name)
{
    rs_prof_slot* s = prof_slot(ctx, name);
    if (!s || s->ev.size() < 2) return;
    (void)hipEventRecord(s->ev.back(), ctx->stream);
    s->launches++;
}

extern "C" int rs_prof_begin(rs_context* ctx)
{
    if (!ctx) return RS_ERR_INVALID;
    for (auto& s : ctx->prof)
        for (auto e : s.ev) (void)hipEventDestroy(e);
    ctx->prof.clear();
    ctx->prof_on = true;
    return RS_OK;
}

extern "C" int rs_prof_end(rs_context* ctx, rs_prof_entry* entries, int* count)
{
    if (!ctx || !entries || !count) return RS_ERR_INVALID;
    ctx->prof_on = false;
    RS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int n = 0;
    for (auto& s : ctx->prof) {
        rs_prof_entry& e = entries[n++];
        memcpy(e.name, s.name, sizeof e.name);
        e.launches = s.launches;
        e.total_ms = 0.0;
        for (size_t i = 0; i + 1 < s.ev.size(); i += 2) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, s.ev[i], s.ev[i + 1]) == hipSuccess) e.total_ms += ms;
        }
        for (auto ev : s.ev) (void)hipEventDestroy(ev);
        s.ev.clear();
    }
    ctx->prof.clear();
    *count = n;
    return RS_OK;
}

// ----------------------------------------------------------------------- RCCL
// RCCL is bound at run time: a torch process has already loaded librccl.so.1
// (soname match => the same library instance), a plain C++ host gets ROCm's.
typedef struct { char internal[128]; } rs_nccl_uid;
typedef int (*fn_get_uid)(rs_nccl_uid*);
typedef int (*fn_init_rank)(void**, int, rs_nccl_uid, int);
typedef int (*fn_destroy)(void*);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*fn_errstr)(int);
typedef int (*fn_count)(const void*, int*);

static struct {
    void* h = nullptr;
    fn_get_uid get_uid = nullptr;
    fn_init_rank init_rank = nullptr;
    fn_destroy destroy = nullptr;
    fn_allreduce allreduce = nullptr;
    fn_errstr errstr = nullptr;
    fn_count count = nullptr;
} g_rccl;

static int rccl_load(rs_context* ctx)
{
    if (g_rccl.h) return RS_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        g_rccl.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.h) break;
    }
    if (!g_rccl.h) return rs_fail(ctx, RS_ERR_RCCL, "cannot dlopen librccl: %s", dlerror());
    g_rccl.get_uid = (fn_get_uid)dlsym(g_rccl.h, "ncclGetUniqueId");
    g_rccl.init_rank = (fn_init_rank)dlsym(g_rccl.h, "ncclCommInitRank");
    g_rccl.destroy = (fn_destroy)dlsym(g_rccl.h, "ncclCommDestroy");
    g_rccl.allreduce = (fn_allreduce)dlsym(g_rccl.h, "ncclAllReduce");
    g_rccl.errstr = (fn_errstr)dlsym(g_rccl.h, "ncclGetErrorString");
    g_rccl.count = (fn_count)dlsym(g_rccl.h, "ncclCommCount");
    if (!g_rccl.get_uid || !g_rccl.init_rank || !g_rccl.destroy || !g_rccl.allreduce)
        return rs_fail(ctx, RS_ERR_RCCL, "librccl lacks the expected symbols");
    return RS_OK;
}

extern "C" int rs_comm_get_unique_id(uint8_t id[RS_COMM_ID_BYTES])
{
    int rc = rccl_load(nullptr);
    if (rc) return rc;
    rs_nccl_uid u;
    memset(&u, 0, sizeof u);
    int e = g_rccl.get_uid(&u);
    if (e != 0) return rs_fail(nullptr, RS_ERR_RCCL, "ncclGetUniqueId: %d", e);
    memcpy(id, &u, RS_COMM_ID_BYTES);
    return RS_OK;
}

extern "C" int rs_comm_init_rank(rs_context* ctx, const uint8_t id[RS_COMM_ID_BYTES], int n_ranks, int rank)
{
    if (!ctx || !id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return RS_ERR_INVALID;
    int rc = rccl_load(ctx);
    if (rc) return rc;
    if (ctx->comm) rs_comm_destroy(ctx);
    RS_HIP(ctx, hipSetDevice(ctx->device));
    rs_nccl_uid u;
    memcpy(&u, id, RS_COMM_ID_BYTES);
    void* comm = nullptr;
    int e = g_rccl.init_rank(&comm, n_ranks, u, rank);
    if (e != 0) return rs_fail(ctx, RS_ERR_RCCL, "ncclCommInitRank: %s", g_rccl.errstr ? g_rccl.errstr(e) : "?");
    ctx->comm = comm;
    ctx->n_ranks = n_ranks;
    ctx->rank = rank;
    return RS_OK;
}

// ------------------------------------------------------- in-process group
// Several contexts of ONE process (one host thread each, own stream, same or peer-accessible device) form a group
// whose exchange step needs no RCCL: the landmark-sharded BA then runs, for instance, as two shards on one GPU.
// The all-reduce is a deterministic sum in rank order (every member computes the identical result, as the
// redundant reduced solves require):
//   record "my buffer is ready" -> host barrier -> every stream waits for all ready events -> each member sums all
//   buffers into its own scratch -> record "I have read" -> host barrier -> wait for all -> scratch -> buffer.
#include <condition_variable>
#include <mutex>
#define RS_LOCAL_MAX 8
struct rs_local_group {
    int n = 0;
    int refs = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    unsigned long generation = 0;
    bool failed = false;             // a member hit an error inside an exchange step: every member returns RS_ERR_HIP from it
    double* buf[RS_LOCAL_MAX] = {};
    hipEvent_t ready[RS_LOCAL_MAX] = {}, read_done[RS_LOCAL_MAX] = {};
    double* scratch[RS_LOCAL_MAX] = {};
    size_t scratch_cap[RS_LOCAL_MAX] = {};
};

// Host barrier of the group's member threads.  A member that never arrives (its call failed before the exchange step, or
// its thread died) must not strand the others: the wait is bounded (RS_LOCAL_BARRIER_SECONDS), and a timed-out barrier
// marks the group failed — every later exchange step of every member then returns an error.
#include <chrono>
#define RS_LOCAL_BARRIER_SECONDS 30
static void local_barrier(rs_local_group* g)
{
    std::unique_lock<std::mutex> lk(g->m);
    const unsigned long gen = g->generation;
    if (++g->arrived == g->n) { g->arrived = 0; g->generation++; g->cv.notify_all(); return; }
    if (!g->cv.wait_for(lk, std::chrono::seconds(RS_LOCAL_BARRIER_SECONDS), [&] { return g->generation != gen; })) {
        g->failed = true;                 // give up: release whoever else waits in this generation
        g->arrived = 0;
        g->generation++;
        g->cv.notify_all();
    }
}

struct LocalBufs { double* p[RS_LOCAL_MAX]; };
__global__ void rs_local_sum(LocalBufs b, int n, size_t count, double* __restrict__ out)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        double v = 0.0;
        for (int r = 0; r < n; r++) v += b.p[r][i];
        out[i] = v;
    }
}
__global__ void rs_local_min_u64(LocalBufs b, int n, size_t count, double* __restrict__ out)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long v = ~0ull;
        for (int r = 0; r < n; r++) { const unsigned long long x = ((const unsigned long long*)b.p[r])[i]; v = x < v ? x : v; }
        ((unsigned long long*)out)[i] = v;
    }
}

// (8-byte elements: f64 sums, or unsigned 64-bit minima when min_u64 is set)
static int local_allreduce(rs_context* ctx, double* d_buf, size_t count, bool min_u64 = false)
{
    // A member that hits a HIP error must still arrive at BOTH host barriers — the others would otherwise wait for it for
    // ever — so errors are collected (first one kept in ctx->err), flagged on the group, and every member returns after the
    // second barrier.
    rs_local_group* g = ctx->local;
    const int r = ctx->rank;
    {
        // a group that has failed once (a member's error, or a barrier that timed out) stays failed: its barrier generations
        // may be mixed up by a straggler.  Later exchange steps return at once instead of waiting 2 x 30 s again; the group
        // must be destroyed (rs_comm_destroy on every member) and created anew.
        std::lock_guard<std::mutex> lk(g->m);
        if (g->failed)
            return rs_fail(ctx, RS_ERR_HIP, "the in-process group failed in an earlier exchange step: destroy it (rs_comm_destroy on every member) and create it again");
    }
    bool bad = false;
    auto chk = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && !bad) { bad = true; rs_fail(ctx, RS_ERR_HIP, "%s failed in the in-process exchange step: %s", what, hipGetErrorString(e)); }
    };
    if (count > g->scratch_cap[r]) {
        if (g->scratch[r]) chk(hipFree(g->scratch[r]), "hipFree");
        g->scratch[r] = nullptr;
        g->scratch_cap[r] = 0;
        chk(hipMalloc(&g->scratch[r], sizeof(double) * count), "hipMalloc");
        if (!bad) g->scratch_cap[r] = count;
    }
    g->buf[r] = d_buf;
    chk(hipEventRecord(g->ready[r], ctx->stream), "hipEventRecord");
    if (bad) { std::lock_guard<std::mutex> lk(g->m); g->failed = true; }
    local_barrier(g);
    bool group_bad;
    { std::lock_guard<std::mutex> lk(g->m); group_bad = g->failed; }
    if (!group_bad) {
        LocalBufs lb;
        for (int q = 0; q < RS_LOCAL_MAX; q++) lb.p[q] = q < g->n ? g->buf[q] : nullptr;
        for (int q = 0; q < g->n; q++)
            if (q != r) chk(hipStreamWaitEvent(ctx->stream, g->ready[q], 0), "hipStreamWaitEvent");
        const int blocks = (int)((count + 255) / 256 < 512 ? (count + 255) / 256 : 512);
        if (!bad) {
            if (min_u64) hipLaunchKernelGGL(rs_local_min_u64, dim3(blocks), dim3(256), 0, ctx->stream, lb, g->n, count, g->scratch[r]);
            else hipLaunchKernelGGL(rs_local_sum, dim3(blocks), dim3(256), 0, ctx->stream, lb, g->n, count, g->scratch[r]);
        }
        chk(hipEventRecord(g->read_done[r], ctx->stream), "hipEventRecord");
        if (bad) { std::lock_guard<std::mutex> lk(g->m); g->failed = true; }
    }
    local_barrier(g);
    { std::lock_guard<std::mutex> lk(g->m); group_bad = g->failed; }
    if (group_bad) return bad ? RS_ERR_HIP : rs_fail(ctx, RS_ERR_HIP, "another member of the in-process group failed in the exchange step");
    for (int q = 0; q < g->n; q++)
        if (q != r) chk(hipStreamWaitEvent(ctx->stream, g->read_done[q], 0), "hipStreamWaitEvent");
    chk(hipMemcpyAsync(d_buf, g->scratch[r], sizeof(double) * count, hipMemcpyDeviceToDevice, ctx->stream), "hipMemcpyAsync");
    return bad ? RS_ERR_HIP : RS_OK;
}

extern "C" int rs_comm_init_local(rs_context** ctxs, int n)
{
    if (!ctxs || n < 1 || n > RS_LOCAL_MAX) return RS_ERR_INVALID;
    for (int i = 0; i < n; i++)
        if (!ctxs[i] || rs_comm_active(ctxs[i])) return RS_ERR_INVALID;
    rs_local_group* g = new rs_local_group();
    g->n = n;
    g->refs = n;
    for (int i = 0; i < n; i++) {
        RS_HIP(ctxs[i], hipSetDevice(ctxs[i]->device));
        RS_HIP(ctxs[i], hipEventCreateWithFlags(&g->ready[i], hipEventDisableTiming));
        RS_HIP(ctxs[i], hipEventCreateWithFlags(&g->read_done[i], hipEventDisableTiming));
        ctxs[i]->local = g;
        ctxs[i]->n_ranks = n;
        ctxs[i]->rank = i;
    }
    return RS_OK;
}

extern "C" int rs_comm_destroy(rs_context* ctx)
{
    if (!ctx) return RS_ERR_INVALID;
    if (ctx->comm && g_rccl.destroy) g_rccl.destroy(ctx->comm);
    ctx->comm = nullptr;
    if (ctx->local) {
        rs_local_group* g = ctx->local;
        const int r = ctx->rank;
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        if (g->scratch[r]) (void)hipFree(g->scratch[r]);
        (void)hipEventDestroy(g->ready[r]);
        (void)hipEventDestroy(g->read_done[r]);
        bool last;
        { std::lock_guard<std::mutex> lk(g->m); last = --g->refs == 0; }
        if (last) delete g;
        ctx->local = nullptr;
    }
    ctx->n_ranks = 1;
    ctx->rank = 0;
    return RS_OK;
}

int rs_allreduce_min_u64(rs_context* ctx, unsigned long long* d_buf, size_t count)
{
    if (ctx->local) return local_allreduce(ctx, (double*)d_buf, count, true);
    if (!ctx->comm) return RS_OK;
    // ncclUint64 = 5, ncclMin = 3
    int e = g_rccl.allreduce(d_buf, d_buf, count, 5, 3, ctx->comm, ctx->stream);
    if (e != 0) return rs_fail(ctx, RS_ERR_RCCL, "ncclAllReduce(min, u64): %s", g_rccl.errstr ? g_rccl.errstr(e) : "?");
    return RS_OK;
}

extern "C" int rs_comm_count(rs_context* ctx, int* h_ranks, int* h_kind)
{
    if (!ctx || !h_ranks || !h_kind) return RS_ERR_INVALID;
    *h_ranks = 1;
    *h_kind = 0;
    if (ctx->local) { *h_ranks = ctx->local->n; *h_kind = 2; return RS_OK; }
    if (!ctx->comm) return RS_OK;
    *h_kind = 1;
    if (!g_rccl.count) return rs_fail(ctx, RS_ERR_RCCL, "librccl lacks ncclCommCount");
    int n = 0;
    const int e = g_rccl.count(ctx->comm, &n);
    if (e != 0) return rs_fail(ctx, RS_ERR_RCCL, "ncclCommCount: %s", g_rccl.errstr ? g_rccl.errstr(e) : "?");
    *h_ranks = n;
    return RS_OK;
}

__global__ void rs_empty_kernel() {}

extern "C" int rs_prof_empty_launch(rs_context* ctx, int n, double* h_us)
{
    if (!ctx || !h_us || n < 1 || n > 100000) return RS_ERR_INVALID;
    RS_HIP(ctx, hipSetDevice(ctx->device));
    hipEvent_t a, b;
    RS_HIP(ctx, hipEventCreate(&a));
    RS_HIP(ctx, hipEventCreate(&b));
    for (int i = 0; i < 8; i++) hipLaunchKernelGGL(rs_empty_kernel, dim3(1), dim3(64), 0, ctx->stream);
    RS_HIP(ctx, hipEventRecord(a, ctx->stream));
    for (int i = 0; i < n; i++) hipLaunchKernelGGL(rs_empty_kernel, dim3(1), dim3(64), 0, ctx->stream);
    RS_HIP(ctx, hipEventRecord(b, ctx->stream));
    RS_HIP(ctx, hipEventSynchronize(b));
    float ms = 0.f;
    RS_HIP(ctx, hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    *h_us = 1e3 * (double)ms / n;
    return RS_OK;
}

int rs_allreduce_f64(rs_context* ctx, double* d_buf, size_t count, bool is_max)
{
    if (ctx->local) return is_max ? rs_fail(ctx, RS_ERR_UNSUPPORTED, "max over the in-process group") : local_allreduce(ctx, d_buf, count);
    if (!ctx->comm) return RS_OK;     // a 1-rank communicator still goes through RCCL (exercised by the tests)
    // ncclFloat64 = 8, ncclSum = 0, ncclMax = 2
    int e = g_rccl.allreduce(d_buf, d_buf, count, 8, is_max ? 2 : 0, ctx->comm, ctx->stream);
    if (e != 0) return rs_fail(ctx, RS_ERR_RCCL, "ncclAllReduce: %s", g_rccl.errstr ? g_rccl.errstr(e) : "?");
    return RS_OK;
}

#include <mutex>
#include <unordered_map>
hipError_t rs_lds_attr(const void* fn, size_t bytes)
{
    // the attribute belongs to the function ON THE CURRENT DEVICE: the cache is keyed by (device, function), so that a
    // process with contexts on several devices sets it on each of them
    static std::mutex mu;
    static std::unordered_map<unsigned long long, size_t> have;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    size_t& cur = have[(unsigned long long)(uintptr_t)fn * 64ull + (unsigned long long)(dev & 63)];
    if (bytes <= cur) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) cur = bytes;
    return e;
}
