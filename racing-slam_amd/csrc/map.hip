// map.hip — the map, resident on the device across frames.
//
// The reference's map is a pointer graph on the host (Map owns MapPoint objects, each with an
// unordered_map<KeyFrame*, size_t> of observations; key frames own their descriptor matrices — reference
// src/Map.cpp:63-124, src/MapPoint.cpp, src/Frame.cpp:80-116), and MapMatcher::match walks ALL of it twice per frame
// (src/MapMatcher.cpp:165-175).  A drop-in that re-flattens and re-uploads that graph on every call spends 100x the
// kernel time doing so (bench.py `boundary`: 5.4 ms per match_map for a 45 us kernel).  rs_map keeps the flat form:
//
//   host mirror (this file)      per point: position, alive flag, observation list (key frame, keypoint) in insertion
//                                order; per key frame: pose, keypoints, keypoint -> point table (Frame::map_matches),
//                                row offset of its descriptors in the device pool.  O(1) updates, mirroring the calls
//                                the reference makes: create_point / remove_point / associate / disassociate /
//                                set_position / set_pose.
//   device image                 positions, alive flags, observation CSR (key frame, descriptor row), key-frame centres,
//                                descriptor pool (append only: a key frame's rows are uploaded ONCE).  Re-flattened from
//                                the mirror and uploaded only when the map changed since the last use — i.e. once per
//                                KEY FRAME, not per frame; positions / centres alone when only those moved (after BA).
//
// Point slots are never reused, so ascending slot = creation order = the reference's map order (Map::remove_point erases
// in place, src/Map.cpp:63-76), which is what decides ties between points (src/MapMatcher.cpp:95-97).  Observation order
// within a point = insertion order (the reference iterates an unordered_map: unspecified upstream).
//
// rs_frame is the per-frame counterpart: keypoints, descriptors and the flattened KD-tree (built once, like the
// reference builds it in the Frame constructor, src/Frame.cpp:8-15) uploaded once and shared by the two match calls of a
// frame; a frame that becomes a key frame hands its descriptor rows to the pool device-to-device.
#include <algorithm>

#include "common.h"

struct MapObs { int32_t kf, kp; };

struct MapKeyFrame {
    int n = 0;
    int pool_row = 0;                  // first row of its descriptors in the device pool
    float pose[16];
    std::vector<float> kp;             // [n][2]
    std::vector<int32_t> kp_point;     // [n] point slot matched by keypoint i or -1 (Frame::map_matches)
};

struct rs_map {
    rs_context* ctx = nullptr;
    // host mirror
    std::vector<float> pos;                         // [P][3]
    std::vector<uint8_t> alive;                     // [P]
    std::vector<std::vector<MapObs>> obs;           // [P]
    std::vector<MapKeyFrame> kfs;
    int n_alive = 0;
    bool dirty_topology = true, dirty_positions = true, dirty_centres = true;
    // device image
    float* d_pos = nullptr; uint8_t* d_alive = nullptr; int32_t* d_obs_ptr = nullptr;
    int32_t* d_obs_kf = nullptr; int32_t* d_obs_desc = nullptr; float* d_centres = nullptr;
    uint8_t* d_pool = nullptr; uint8_t* d_elig = nullptr; uint8_t* d_flag = nullptr;
    float* d_kp_pool = nullptr;                     // [pool rows][2] keypoints of the key frames, row for row with the descriptor pool
    size_t cap_kp_pool = 0;
    int32_t* d_win = nullptr; size_t cap_win = 0;   // scratch of rs_map_bundle_adjust: pid [P] | obs offset [P]
    size_t cap_points = 0, cap_obs = 0, cap_kf = 0, cap_pool_bytes = 0, pool_rows = 0, n_obs = 0;
    // scratch for match results
    int32_t* d_out = nullptr; size_t cap_out = 0;
    std::vector<int32_t> h_obs_ptr, h_obs_kf, h_obs_desc;
    std::vector<float> h_centres;
};

struct rs_frame {
    rs_context* ctx = nullptr;
    int n = 0;
    float* d_kp = nullptr; uint8_t* d_desc = nullptr; int32_t* d_kd = nullptr;   // node_kp | left | right
    uint8_t* d_matched = nullptr;
    void* d_packed = nullptr;           // {x, y, left, right}[n] + keypoint[n]: what K2 stages in LDS (rs_kdtree_pack layout)
    int kd_root = -1;
    std::vector<float> kp;
};

template <typename T>
static int grow(rs_context* ctx, T** p, size_t* cap, size_t need, size_t keep_bytes)
{
    if (need <= *cap) return RS_OK;
    size_t want = *cap ? *cap * 2 : 1024;
    while (want < need) want *= 2;
    T* q = nullptr;
    if (hipMalloc((void**)&q, sizeof(T) * want) != hipSuccess) return rs_fail(ctx, RS_ERR_NOMEM, "map buffer of %zu entries", want);
    if (*p) {
        RS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (keep_bytes) RS_HIP(ctx, hipMemcpy(q, *p, keep_bytes, hipMemcpyDeviceToDevice));
        RS_HIP(ctx, hipFree(*p));
    }
    *p = q;
    *cap = want;
    return RS_OK;
}

extern "C" int rs_map_create(rs_context* ctx, rs_map** out)
{
    if (!ctx || !out) return RS_ERR_INVALID;
    rs_map* m = new rs_map();
    m->ctx = ctx;
    *out = m;
    return RS_OK;
}

extern "C" int rs_map_destroy(rs_map* m)
{
    if (!m) return RS_OK;
    (void)hipSetDevice(m->ctx->device);
    (void)hipStreamSynchronize(m->ctx->stream);
    for (void* p : {(void*)m->d_pos, (void*)m->d_alive, (void*)m->d_obs_ptr, (void*)m->d_obs_kf, (void*)m->d_obs_desc,
                    (void*)m->d_centres, (void*)m->d_pool, (void*)m->d_elig, (void*)m->d_flag, (void*)m->d_out, (void*)m->d_kp_pool,
                    (void*)m->d_win})
        if (p) (void)hipFree(p);
    delete m;
    return RS_OK;
}

// ------------------------------------------------------------------------------------------------ frames
extern "C" int rs_frame_create(rs_context* ctx, const float* h_kp, const uint8_t* h_desc, int n, rs_frame** out)
{
    if (!ctx || !out || n < 0 || (n > 0 && (!h_kp || !h_desc))) return RS_ERR_INVALID;
    RS_HIP(ctx, hipSetDevice(ctx->device));
    rs_frame* f = new rs_frame();
    f->ctx = ctx;
    f->n = n;
    f->kp.assign(h_kp, h_kp + 2 * (size_t)n);
    const size_t m = n > 0 ? (size_t)n : 1;
    std::vector<int32_t> kd(3 * m);
    int32_t root = -1;
    if (n > 0) rs_kdtree_build(h_kp, n, kd.data(), kd.data() + m, kd.data() + 2 * m, &root);     // src/Frame.cpp:8-15
    f->kd_root = root;
    if (hipMalloc((void**)&f->d_kp, sizeof(float) * 2 * m) != hipSuccess || hipMalloc((void**)&f->d_desc, 32 * m) != hipSuccess ||
        hipMalloc((void**)&f->d_kd, sizeof(int32_t) * 3 * m) != hipSuccess || hipMalloc((void**)&f->d_matched, m) != hipSuccess) {
        delete f;
        return rs_fail(ctx, RS_ERR_NOMEM, "frame buffers");
    }
    if (n > 0) {
        RS_HIP(ctx, hipMemcpyAsync(f->d_kp, h_kp, sizeof(float) * 2 * m, hipMemcpyHostToDevice, ctx->stream));
        RS_HIP(ctx, hipMemcpyAsync(f->d_desc, h_desc, 32 * m, hipMemcpyHostToDevice, ctx->stream));
        RS_HIP(ctx, hipMemcpyAsync(f->d_kd, kd.data(), sizeof(int32_t) * 3 * m, hipMemcpyHostToDevice, ctx->stream));
        std::vector<int32_t> packed(5 * m);
        for (size_t i = 0; i < m; i++) {
            const int32_t kpi = kd[i];
            memcpy(&packed[4 * i], &h_kp[2 * (size_t)kpi], 8);
            packed[4 * i + 2] = kd[m + i];
            packed[4 * i + 3] = kd[2 * m + i];
            packed[4 * m + i] = kpi;
        }
        if (hipMalloc(&f->d_packed, 20 * m) != hipSuccess) { delete f; return rs_fail(ctx, RS_ERR_NOMEM, "frame buffers"); }
        RS_HIP(ctx, hipMemcpyAsync(f->d_packed, packed.data(), 20 * m, hipMemcpyHostToDevice, ctx->stream));
        RS_HIP(ctx, hipStreamSynchronize(ctx->stream));       // the sources are the caller's / this function's memory
    }
    *out = f;
    return RS_OK;
}

extern "C" int rs_frame_destroy(rs_frame* f)
{
    if (!f) return RS_OK;
    (void)hipSetDevice(f->ctx->device);
    (void)hipStreamSynchronize(f->ctx->stream);
    for (void* p : {(void*)f->d_kp, (void*)f->d_desc, (void*)f->d_kd, (void*)f->d_matched, f->d_packed})
        if (p) (void)hipFree(p);
    delete f;
    return RS_OK;
}

// ------------------------------------------------------------------------------------------------ updates
static void centre_of(const float* T, float c[3])      // Frame::camera_center = -R^T t, src/Frame.cpp:39-42
{
    for (int i = 0; i < 3; i++) c[i] = (-T[i] * T[3] + -T[4 + i] * T[7]) + -T[8 + i] * T[11];
}

extern "C" int rs_map_add_keyframe(rs_map* m, const rs_frame* f, const float h_pose[16], int* out_kf)
{
    if (!m || !f || !h_pose || !out_kf || f->ctx != m->ctx) return RS_ERR_INVALID;
    rs_context* ctx = m->ctx;
    RS_HIP(ctx, hipSetDevice(ctx->device));
    const size_t rows = m->pool_rows + (size_t)f->n;
    int rc = grow(ctx, &m->d_pool, &m->cap_pool_bytes, 32 * (rows > 0 ? rows : 1), 32 * m->pool_rows);   // keeps the rows already there
    if (rc) return rc;
    rc = grow(ctx, &m->d_kp_pool, &m->cap_kp_pool, 2 * (rows > 0 ? rows : 1), sizeof(float) * 2 * m->pool_rows);
    if (rc) return rc;
    MapKeyFrame k;
    k.n = f->n;
    k.pool_row = (int)m->pool_rows;
    memcpy(k.pose, h_pose, sizeof k.pose);
    k.kp = f->kp;
    k.kp_point.assign((size_t)f->n, -1);
    if (f->n > 0)
        RS_HIP(ctx, hipMemcpyAsync(m->d_pool + 32 * m->pool_rows, f->d_desc, 32 * (size_t)f->n, hipMemcpyDeviceToDevice, ctx->stream));
    if (f->n > 0)
        RS_HIP(ctx, hipMemcpyAsync(m->d_kp_pool + 2 * m->pool_rows, f->d_kp, sizeof(float) * 2 * (size_t)f->n, hipMemcpyDeviceToDevice, ctx->stream));
    m->pool_rows = rows;
    m->kfs.push_back(std::move(k));
    m->dirty_centres = true;
    *out_kf = (int)m->kfs.size() - 1;
    return RS_OK;
}

extern "C" int rs_map_set_keyframe_pose(rs_map* m, int kf, const float h_pose[16])
{
    if (!m || !h_pose || kf < 0 || kf >= (int)m->kfs.size()) return RS_ERR_INVALID;
    memcpy(m->kfs[(size_t)kf].pose, h_pose, sizeof(float) * 16);
    m->dirty_centres = true;
    return RS_OK;
}

extern "C" int rs_map_add_point(rs_map* m, const float xyz[3], int* out_point)
{
    if (!m || !xyz || !out_point) return RS_ERR_INVALID;
    m->pos.insert(m->pos.end(), xyz, xyz + 3);
    m->alive.push_back(1);
    m->obs.emplace_back();
    m->n_alive++;
    m->dirty_topology = m->dirty_positions = true;
    *out_point = (int)m->alive.size() - 1;
    return RS_OK;
}

static bool point_ok(const rs_map* m, int p) { return p >= 0 && p < (int)m->alive.size() && m->alive[(size_t)p]; }

extern "C" int rs_map_set_position(rs_map* m, int point, const float xyz[3])
{
    if (!m || !xyz || !point_ok(m, point)) return RS_ERR_INVALID;
    memcpy(&m->pos[3 * (size_t)point], xyz, sizeof(float) * 3);
    m->dirty_positions = true;
    return RS_OK;
}

extern "C" int rs_map_remove_observation(rs_map* m, int point, int kf)
{
    if (!m || !point_ok(m, point) || kf < 0 || kf >= (int)m->kfs.size()) return RS_ERR_INVALID;
    auto& v = m->obs[(size_t)point];
    for (size_t i = 0; i < v.size(); i++)
        if (v[i].kf == kf) {
            auto& tab = m->kfs[(size_t)kf].kp_point;
            if (tab[(size_t)v[i].kp] == point) tab[(size_t)v[i].kp] = -1;
            v.erase(v.begin() + (long)i);
            m->dirty_topology = true;
            return RS_OK;
        }
    return RS_OK;       // MapPoint::remove_observation of an absent key frame is a no-op (src/Map.cpp:117-124)
}

// Map::associate (src/Map.cpp:95-113): the key frame's keypoint and the point end up matched to each other; whatever
// either was matched to before (in that key frame) is disassociated first.
extern "C" int rs_map_add_observation(rs_map* m, int point, int kf, int keypoint)
{
    if (!m || !point_ok(m, point) || kf < 0 || kf >= (int)m->kfs.size()) return RS_ERR_INVALID;
    MapKeyFrame& k = m->kfs[(size_t)kf];
    if (keypoint < 0 || keypoint >= k.n) return RS_ERR_INVALID;
    const int existing = k.kp_point[(size_t)keypoint];
    auto& v = m->obs[(size_t)point];
    bool seen = false;
    for (const auto& o : v) seen = seen || o.kf == kf;
    if (existing == point && seen) return RS_OK;                          // :97-100
    if (existing >= 0 && existing != point) rs_map_remove_observation(m, existing, kf);     // :101-106
    if (seen) rs_map_remove_observation(m, point, kf);                    // :107-109
    v.push_back({kf, keypoint});
    k.kp_point[(size_t)keypoint] = point;
    m->dirty_topology = true;
    return RS_OK;
}

extern "C" int rs_map_remove_point(rs_map* m, int point)
{
    if (!m || !point_ok(m, point)) return RS_ERR_INVALID;
    for (const auto& o : m->obs[(size_t)point]) {                         // Frame::remove_map_match for every observer
        auto& tab = m->kfs[(size_t)o.kf].kp_point;
        if (tab[(size_t)o.kp] == point) tab[(size_t)o.kp] = -1;
    }
    m->obs[(size_t)point].clear();
    m->obs[(size_t)point].shrink_to_fit();
    m->alive[(size_t)point] = 0;
    m->n_alive--;
    m->dirty_topology = true;
    return RS_OK;
}

extern "C" int rs_map_counts(const rs_map* m, int h_out[4])
{
    if (!m || !h_out) return RS_ERR_INVALID;
    size_t no = 0;
    for (const auto& v : m->obs) no += v.size();
    h_out[0] = (int)m->alive.size(); h_out[1] = m->n_alive; h_out[2] = (int)no; h_out[3] = (int)m->kfs.size();
    return RS_OK;
}

extern "C" int rs_map_get_positions(const rs_map* m, int first, int count, float* h_xyz)
{
    if (!m || first < 0 || count < 0 || (size_t)first + (size_t)count > m->alive.size() || (count > 0 && !h_xyz)) return RS_ERR_INVALID;
    if (count) memcpy(h_xyz, &m->pos[3 * (size_t)first], sizeof(float) * 3 * (size_t)count);
    return RS_OK;
}

// ------------------------------------------------------------------------------------------------ device image
static int map_sync_device(rs_map* m)
{
    rs_context* ctx = m->ctx;
    const size_t P = m->alive.size(), KF = m->kfs.size();
    hipStream_t s = ctx->stream;
    if (m->dirty_topology) {
        m->h_obs_ptr.resize(P + 1);
        m->h_obs_kf.clear();
        m->h_obs_desc.clear();
        for (size_t p = 0; p < P; p++) {
            m->h_obs_ptr[p] = (int32_t)m->h_obs_kf.size();
            for (const auto& o : m->obs[p]) {
                m->h_obs_kf.push_back(o.kf);
                m->h_obs_desc.push_back(m->kfs[(size_t)o.kf].pool_row + o.kp);
            }
        }
        m->h_obs_ptr[P] = (int32_t)m->h_obs_kf.size();
        m->n_obs = m->h_obs_kf.size();
        if (P > m->cap_points) {
            // every per-point array is re-uploaded / recomputed / zeroed below: nothing to preserve
            RS_HIP(ctx, hipStreamSynchronize(s));
            for (void* q : {(void*)m->d_alive, (void*)m->d_obs_ptr, (void*)m->d_pos, (void*)m->d_elig, (void*)m->d_flag})
                if (q) RS_HIP(ctx, hipFree(q));
            m->d_alive = nullptr; m->d_obs_ptr = nullptr; m->d_pos = nullptr; m->d_elig = nullptr; m->d_flag = nullptr;
            size_t cap = m->cap_points ? m->cap_points : 4096;
            while (cap < P) cap *= 2;
            if (hipMalloc((void**)&m->d_alive, cap) != hipSuccess || hipMalloc((void**)&m->d_obs_ptr, sizeof(int32_t) * (cap + 1)) != hipSuccess ||
                hipMalloc((void**)&m->d_pos, sizeof(float) * 3 * cap) != hipSuccess || hipMalloc((void**)&m->d_elig, cap) != hipSuccess ||
                hipMalloc((void**)&m->d_flag, cap) != hipSuccess)
                return rs_fail(ctx, RS_ERR_NOMEM, "map buffers for %zu points", cap);
            RS_HIP(ctx, hipMemsetAsync(m->d_flag, 0, cap, s));
            m->cap_points = cap;
            m->dirty_positions = true;
        }
        if (m->n_obs > m->cap_obs) {
            RS_HIP(ctx, hipStreamSynchronize(s));
            if (m->d_obs_kf) RS_HIP(ctx, hipFree(m->d_obs_kf));
            if (m->d_obs_desc) RS_HIP(ctx, hipFree(m->d_obs_desc));
            m->d_obs_kf = nullptr; m->d_obs_desc = nullptr;
            size_t cap = m->cap_obs ? m->cap_obs : 16384;
            while (cap < m->n_obs) cap *= 2;
            if (hipMalloc((void**)&m->d_obs_kf, sizeof(int32_t) * cap) != hipSuccess || hipMalloc((void**)&m->d_obs_desc, sizeof(int32_t) * cap) != hipSuccess)
                return rs_fail(ctx, RS_ERR_NOMEM, "map buffers for %zu observations", cap);
            m->cap_obs = cap;
        }
        // pageable sources: these three copies complete before returning only because of the synchronisation below
        if (P) RS_HIP(ctx, hipMemcpyAsync(m->d_alive, m->alive.data(), P, hipMemcpyHostToDevice, s));
        RS_HIP(ctx, hipMemcpyAsync(m->d_obs_ptr, m->h_obs_ptr.data(), sizeof(int32_t) * (P + 1), hipMemcpyHostToDevice, s));
        if (m->n_obs) {
            RS_HIP(ctx, hipMemcpyAsync(m->d_obs_kf, m->h_obs_kf.data(), sizeof(int32_t) * m->n_obs, hipMemcpyHostToDevice, s));
            RS_HIP(ctx, hipMemcpyAsync(m->d_obs_desc, m->h_obs_desc.data(), sizeof(int32_t) * m->n_obs, hipMemcpyHostToDevice, s));
        }
    }
    if (m->dirty_positions && P) RS_HIP(ctx, hipMemcpyAsync(m->d_pos, m->pos.data(), sizeof(float) * 3 * P, hipMemcpyHostToDevice, s));
    if (m->dirty_centres) {
        m->h_centres.resize(3 * (KF ? KF : 1));
        for (size_t k = 0; k < KF; k++) centre_of(m->kfs[k].pose, &m->h_centres[3 * k]);
        size_t ck = m->cap_kf;
        int rc = grow(ctx, &m->d_centres, &ck, 3 * (KF ? KF : 1), 0);
        if (rc) return rc;
        m->cap_kf = ck;
        if (KF) RS_HIP(ctx, hipMemcpyAsync(m->d_centres, m->h_centres.data(), sizeof(float) * 3 * KF, hipMemcpyHostToDevice, s));
    }
    if (m->dirty_topology || m->dirty_positions || m->dirty_centres) RS_HIP(ctx, hipStreamSynchronize(s));
    m->dirty_topology = m->dirty_positions = m->dirty_centres = false;
    return RS_OK;
}

// per-call eligibility (src/MapMatcher.cpp:53, :169): alive, not already matched by the frame, and — for
// match_key_frame — observed by the required key frame
// flag table [P], all zero between calls: bit 0 = the frame already matches the point
__global__ __launch_bounds__(256) void k_map_flag(const int32_t* __restrict__ idx, int n, uint8_t* __restrict__ flag, uint8_t bit, int set)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        unsigned int* w = (unsigned int*)(flag + (idx[i] & ~3));
        const unsigned int msk = (unsigned int)bit << (8 * (idx[i] & 3));
        if (set) atomicOr(w, msk); else atomicAnd(w, ~msk);
    }
}

__global__ __launch_bounds__(256) void k_map_eligible(int P, const uint8_t* __restrict__ alive, const uint8_t* __restrict__ flag,
                                                      const int32_t* __restrict__ obs_ptr, const int32_t* __restrict__ obs_kf,
                                                      int required_kf, int listed_only, uint8_t* __restrict__ elig)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    bool e = alive[p] != 0 && (flag[p] & 1) == 0 && (!listed_only || (flag[p] & 2) != 0);
    if (e && required_kf >= 0) {
        bool seen = false;
        for (int o = obs_ptr[p]; o < obs_ptr[p + 1]; o++) seen = seen || obs_kf[o] == required_kf;
        e = seen;
    }
    elig[p] = e ? 1 : 0;
}

// MapMatcher::match_map / match_key_frame / match_for_fuse against the resident map.
//   h_kp_matched [n]       Frame::is_matched(keypoint)                    (src/MapMatcher.cpp:81)
//   h_matched_points       the point slots the frame already matches      (:53)
//   required_observer_kf   >= 0: only points observed by that key frame (:169); -1: none
//   h_only_points          match_for_fuse (:117-127): only these slots take part (n_only < 0: the whole map).  They
//                          compete for a keypoint in LIST order, as the reference's loop over its vector does (:120-126)
//                          and as the flattened path does (the shim flattens the vector in order): the listed points are
//                          gathered from the mirror into a small view of their own, point i of it = h_only_points[i].
//                          Dead slots in the list are skipped like the reference's null pointers (:121-123).
// Every host list is validated and every buffer is grown BEFORE the first launch, and the flag table is cleared by the
// same launches that follow the eligibility kernel whatever happens later: it is all-zero between calls.
extern "C" int rs_map_match(rs_context* ctx, rs_map* m, rs_frame* f, const float h_pose[16], const float h_intrinsics[4],
                            int width, int height, const uint8_t* h_kp_matched, const int32_t* h_matched_points,
                            int n_matched_points, int required_observer_kf, const int32_t* h_only_points, int n_only,
                            int replace, int max_distance, int32_t* h_match_kp, int32_t* h_match_point, int* h_count)
{
    if (!ctx || !m || !f || m->ctx != ctx || f->ctx != ctx || !h_pose || !h_intrinsics || !h_count) return RS_ERR_INVALID;
    if (n_matched_points < 0 || (n_matched_points > 0 && !h_matched_points)) return rs_fail(ctx, RS_ERR_INVALID, "matched point list");
    if (required_observer_kf >= (int)m->kfs.size()) return rs_fail(ctx, RS_ERR_INVALID, "unknown key frame");
    *h_count = 0;
    const int N = f->n, P = (int)m->alive.size();
    if (N == 0 || P == 0) return RS_OK;
    if (!h_match_kp || !h_match_point) return rs_fail(ctx, RS_ERR_INVALID, "null output");
    if (n_only > 0 && !h_only_points) return rs_fail(ctx, RS_ERR_INVALID, "null point list");
    for (int i = 0; i < n_matched_points; i++)
        if (h_matched_points[i] < 0 || h_matched_points[i] >= P) return rs_fail(ctx, RS_ERR_INVALID, "matched point %d out of range", i);
    for (int i = 0; i < n_only; i++)
        if (h_only_points[i] < 0 || h_only_points[i] >= P) return rs_fail(ctx, RS_ERR_INVALID, "listed point %d out of range", i);
    if (n_only == 0) return RS_OK;                       // an empty list: nothing takes part
    RS_HIP(ctx, hipSetDevice(ctx->device));
    int rc = map_sync_device(m);
    if (rc) return rc;
    const bool listed = n_only > 0;
    const int Pv = listed ? n_only : P;                  // points of the view the matcher runs on
    const size_t need = 2 * (size_t)Pv + 4 * (size_t)N + 1;
    size_t co = m->cap_out;
    rc = grow(ctx, &m->d_out, &co, need, 0);
    if (rc) return rc;
    m->cap_out = co;
    rc = rs_stage_begin(ctx);
    if (rc) return rc;
    hipStream_t s = ctx->stream;
    // the frame's per-call state: which keypoints / points it already matches
    uint8_t* d_matched = nullptr;
    if (h_kp_matched) { rc = rs_stage_upload(ctx, h_kp_matched, (size_t)N, (void**)&d_matched); if (rc) return rc; }
    else { rc = rs_stage_alloc(ctx, (size_t)N, (void**)&d_matched); if (rc) return rc; RS_HIP(ctx, hipMemsetAsync(d_matched, 0, (size_t)N, s)); }
    rs_map_view mv{P, m->d_pos, m->d_elig, m->d_obs_ptr, m->d_obs_kf, m->d_obs_desc, m->d_centres, m->d_pool};
    if (listed) {
        // the listed points as a view of their own, in list order, from the mirror (map_sync_device has just rebuilt the
        // observation tables): a fuse list is a few hundred points
        std::vector<uint8_t> taken((size_t)P, 0), elig((size_t)n_only);
        for (int i = 0; i < n_matched_points; i++) taken[(size_t)h_matched_points[i]] = 1;
        std::vector<float> pos(3 * (size_t)n_only);
        std::vector<int32_t> optr((size_t)n_only + 1), okf, odesc;
        for (int i = 0; i < n_only; i++) {
            const size_t p = (size_t)h_only_points[i];
            memcpy(&pos[3 * (size_t)i], &m->pos[3 * p], sizeof(float) * 3);
            optr[(size_t)i] = (int32_t)okf.size();
            bool e = m->alive[p] != 0 && !taken[p];
            bool seen = required_observer_kf < 0;
            for (int32_t o = m->h_obs_ptr[p]; o < m->h_obs_ptr[p + 1]; o++) {
                okf.push_back(m->h_obs_kf[(size_t)o]);
                odesc.push_back(m->h_obs_desc[(size_t)o]);
                seen = seen || m->h_obs_kf[(size_t)o] == required_observer_kf;
            }
            elig[(size_t)i] = (e && seen) ? 1 : 0;
        }
        optr[(size_t)n_only] = (int32_t)okf.size();
        if (okf.empty()) { okf.push_back(0); odesc.push_back(0); }
        float* d_p = nullptr; uint8_t* d_e = nullptr; int32_t *d_op = nullptr, *d_ok = nullptr, *d_od = nullptr;
        if ((rc = rs_stage_upload(ctx, pos.data(), sizeof(float) * pos.size(), (void**)&d_p))) return rc;
        if ((rc = rs_stage_upload(ctx, elig.data(), elig.size(), (void**)&d_e))) return rc;
        if ((rc = rs_stage_upload(ctx, optr.data(), sizeof(int32_t) * optr.size(), (void**)&d_op))) return rc;
        if ((rc = rs_stage_upload(ctx, okf.data(), sizeof(int32_t) * okf.size(), (void**)&d_ok))) return rc;
        if ((rc = rs_stage_upload(ctx, odesc.data(), sizeof(int32_t) * odesc.size(), (void**)&d_od))) return rc;
        mv = rs_map_view{n_only, d_p, d_e, d_op, d_ok, d_od, m->d_centres, m->d_pool};
    } else {
        int32_t* d_mp = nullptr;
        if (n_matched_points > 0) {
            rc = rs_stage_upload(ctx, h_matched_points, sizeof(int32_t) * (size_t)n_matched_points, (void**)&d_mp);
            if (rc) return rc;
            hipLaunchKernelGGL(k_map_flag, dim3((n_matched_points + 255) / 256), dim3(256), 0, s, d_mp, n_matched_points, m->d_flag, (uint8_t)1, 1);
        }
        {
            rs_prof_scope ps(ctx, "K2p_map_eligible");
            hipLaunchKernelGGL(k_map_eligible, dim3((P + 255) / 256), dim3(256), 0, s, P, m->d_alive, m->d_flag, m->d_obs_ptr, m->d_obs_kf,
                               required_observer_kf, 0, m->d_elig);
        }
        if (n_matched_points > 0)       // leave the flag table all-zero for the next call (enqueued before anything below can fail)
            hipLaunchKernelGGL(k_map_flag, dim3((n_matched_points + 255) / 256), dim3(256), 0, s, d_mp, n_matched_points, m->d_flag, (uint8_t)1, 0);
    }
    int32_t* pk = m->d_out, *pd = pk + Pv, *pp = pd + Pv, *pdist = pp + N, *mkp = pdist + N, *mpt = mkp + N, *cnt = mpt + N;
    rs_frame_view fv{};
    memcpy(fv.pose, h_pose, sizeof fv.pose);
    fv.fx = h_intrinsics[0]; fv.fy = h_intrinsics[1]; fv.cx = h_intrinsics[2]; fv.cy = h_intrinsics[3];
    fv.width = width; fv.height = height; fv.n_keypoints = N;
    fv.d_keypoints = f->d_kp; fv.d_descriptors = f->d_desc; fv.d_kp_matched = d_matched;
    fv.d_kd_node_kp = f->d_kd; fv.d_kd_left = f->d_kd + N; fv.d_kd_right = f->d_kd + 2 * (size_t)N; fv.kd_root = f->kd_root;
    fv.d_kd_packed = f->d_packed;
    rc = rs_reproj_match(ctx, &fv, &mv, replace, max_distance, pk, pd, pp, pdist, mkp, mpt, cnt);
    if (rc) return rc;
    int32_t n_out = 0;
    rc = rs_stage_download(ctx, cnt, sizeof(int32_t), &n_out);
    if (rc) return rc;
    rc = rs_stage_download(ctx, mkp, sizeof(int32_t) * (size_t)N, h_match_kp);      // at most N matches: one read-back
    if (rc) return rc;
    rc = rs_stage_download(ctx, mpt, sizeof(int32_t) * (size_t)N, h_match_point);
    if (rc) return rc;
    rc = rs_stage_sync(ctx);
    if (rc) return rc;
    if (listed)
        for (int i = 0; i < n_out; i++) h_match_point[i] = h_only_points[h_match_point[i]];      // view index -> map slot
    *h_count = n_out;
    return RS_OK;
}

// ------------------------------------------------------------------------------------------------ local BA
// optimization::bundle_adjust on the resident map (reference src/Optimization.cpp:269-374).  The window's problem — free
// points, observation CSR by landmark, observation pixels, f64 points — is built ON THE DEVICE from the resident image
// (observation CSR by point, key-point pool): three small kernels over the map's point slots instead of three host passes
// over 20 x 2000 key-point tables plus 1 MB of uploads.  The host packs the <= ~40 poses, reads back two counters, calls
// rs_bundle_adjust, and on a usable solve a write-back kernel puts the positions into the device image while the mirror
// and the caller get them through one download.
//   free points      alive, >= 2 observations in the whole map, matched by an optimised frame of the list (:287-302), in
//                    ascending SLOT order (the reference walks its frames' match tables; the set is the same, and the
//                    order only decides f64 summation order);
//   residual blocks  every listed frame x its matched free points (:304-315), CSR by point, frames in list order.
//   h_out_poses [n_kfs][16]           (rows of fixed key frames are their unchanged poses)
//   h_out_points [cap], h_out_xyz [cap][3], *h_n_points    the free points (slots) and their new positions
__global__ __launch_bounds__(256) void k_win_count(int P, const uint8_t* __restrict__ alive, const int32_t* __restrict__ obs_ptr,
                                                   const int32_t* __restrict__ obs_kf, const int32_t* __restrict__ win_of_kf,
                                                   int32_t* __restrict__ flag, int32_t* __restrict__ cnt)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const int o0 = obs_ptr[p], o1 = obs_ptr[p + 1];
    int c = 0, fr = 0;
    for (int o = o0; o < o1; o++) {
        const int w = win_of_kf[obs_kf[o]];       // -1: not in the list; else list index | free << 16
        if (w >= 0) { c++; fr |= w >> 16; }
    }
    const int f = (alive[p] != 0 && o1 - o0 >= 2 && fr) ? 1 : 0;
    flag[p] = f;
    cnt[p] = f ? c : 0;
}

// exclusive scans of flag -> pid and cnt -> obs offset in place, totals to h_tot (one workgroup; the map has 1e4 - 1e5 slots)
__global__ __launch_bounds__(1024) void k_win_scan(int P, int32_t* __restrict__ flag, int32_t* __restrict__ cnt, int32_t* __restrict__ tot)
{
    __shared__ int carry[2];
    if (threadIdx.x == 0) { carry[0] = 0; carry[1] = 0; }
    __syncthreads();
    for (int base = 0; base < P; base += 1024) {
        const int i = base + (int)threadIdx.x;
        const int f = i < P ? flag[i] : 0, c = i < P ? cnt[i] : 0;
        int tf, tc;
        const int of = rs_block_exclusive_scan(f, &tf);
        const int oc = rs_block_exclusive_scan(c, &tc);
        if (i < P) { flag[i] = f ? carry[0] + of : -1; cnt[i] = carry[1] + oc; }
        __syncthreads();
        if (threadIdx.x == 0) { carry[0] += tf; carry[1] += tc; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { tot[0] = carry[0]; tot[1] = carry[1]; }
}

__global__ __launch_bounds__(256) void k_win_fill(int P, int C, const int32_t* __restrict__ pid, const int32_t* __restrict__ off,
                                                  const int32_t* __restrict__ obs_ptr, const int32_t* __restrict__ obs_kf,
                                                  const int32_t* __restrict__ obs_desc, const int32_t* __restrict__ win_of_kf,
                                                  const float* __restrict__ pos, const float2* __restrict__ kp_pool,
                                                  double* __restrict__ pts, int32_t* __restrict__ list, int32_t* __restrict__ optr,
                                                  int32_t* __restrict__ ocam, float2* __restrict__ ouv, int n_free, int n_obs)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p == 0) optr[n_free] = n_obs;
    if (p >= P) return;
    const int q = pid[p];
    if (q < 0) return;
    list[q] = p;
#pragma unroll
    for (int k = 0; k < 3; k++) pts[3 * (size_t)q + k] = (double)pos[3 * (size_t)p + k];
    int w = off[p];
    optr[q] = w;
    const int o0 = obs_ptr[p], o1 = obs_ptr[p + 1];
    // frames in LIST order (a point has at most one observation per key frame)
    for (int c = 0; c < C; c++)
        for (int o = o0; o < o1; o++) {
            const int wk = win_of_kf[obs_kf[o]];
            if (wk >= 0 && (wk & 0xFFFF) == c) { ocam[w] = c; ouv[w] = kp_pool[obs_desc[o]]; w++; }
        }
}

__global__ __launch_bounds__(256) void k_win_writeback(int n_free, const int32_t* __restrict__ list, const double* __restrict__ pts,
                                                       float* __restrict__ pos, float* __restrict__ oxyz)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_free) return;
    const int p = list[q];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float v = (float)pts[3 * (size_t)q + k];                              // :369-372
        pos[3 * (size_t)p + k] = v;
        oxyz[3 * (size_t)q + k] = v;
    }
}

extern "C" int rs_map_bundle_adjust(rs_context* ctx, rs_map* m, const int32_t* h_kfs, const uint8_t* h_free, int n_kfs,
                                    const float h_intrinsics[4], const rs_ba_options* options, rs_ba_summary* h_summary,
                                    float* h_out_poses, int32_t* h_out_points, float* h_out_xyz, int capacity, int* h_n_points)
{
    if (!ctx || !m || m->ctx != ctx || !h_kfs || !h_free || n_kfs < 0 || !h_intrinsics || !h_summary || !h_n_points) return RS_ERR_INVALID;
    *h_n_points = 0;
    memset(h_summary, 0, sizeof *h_summary);
    const size_t C = (size_t)n_kfs, KF = m->kfs.size(), P = m->alive.size();
    if (C > 65535) return rs_fail(ctx, RS_ERR_UNSUPPORTED, "more than 65535 frames in a window");
    std::vector<int32_t> win(KF ? KF : 1, -1);
    for (size_t c = 0; c < C; c++) {
        if (h_kfs[c] < 0 || h_kfs[c] >= (int)KF) return rs_fail(ctx, RS_ERR_INVALID, "unknown key frame");
        if (win[(size_t)h_kfs[c]] >= 0) return rs_fail(ctx, RS_ERR_INVALID, "key frame %d listed twice", h_kfs[c]);
        win[(size_t)h_kfs[c]] = (int32_t)c | (h_free[c] ? 1 << 16 : 0);
    }
    auto keep_poses = [&]() {
        if (h_out_poses) for (size_t c = 0; c < C; c++) memcpy(h_out_poses + 16 * c, m->kfs[(size_t)h_kfs[c]].pose, sizeof(float) * 16);
    };
    if (P == 0 || C == 0) { h_summary->termination = RS_BA_FAILURE; keep_poses(); return RS_OK; }
    RS_HIP(ctx, hipSetDevice(ctx->device));
    int rc = map_sync_device(m);
    if (rc) return rc;
    size_t cw = m->cap_win;
    rc = grow(ctx, &m->d_win, &cw, 2 * P + 2, 0);
    if (rc) return rc;
    m->cap_win = cw;
    if ((rc = rs_stage_begin(ctx))) return rc;
    hipStream_t s = ctx->stream;
    int32_t* d_winkf = nullptr;
    if ((rc = rs_stage_upload(ctx, win.data(), sizeof(int32_t) * win.size(), (void**)&d_winkf))) return rc;
    int32_t *d_pid = m->d_win, *d_off = d_pid + P, *d_tot = d_off + P;
    const int pb = (int)((P + 255) / 256);
    {
        rs_prof_scope ps(ctx, "K9_window_build");
        hipLaunchKernelGGL(k_win_count, dim3(pb), dim3(256), 0, s, (int)P, m->d_alive, m->d_obs_ptr, m->d_obs_kf, d_winkf, d_pid, d_off);
        hipLaunchKernelGGL(k_win_scan, dim3(1), dim3(1024), 0, s, (int)P, d_pid, d_off, d_tot);
    }
    int32_t tot[2] = {0, 0};
    if ((rc = rs_stage_download(ctx, d_tot, sizeof tot, tot))) return rc;
    if ((rc = rs_stage_sync(ctx))) return rc;
    const size_t Pf = (size_t)tot[0], M = (size_t)tot[1];
    if (Pf == 0 || M == 0) { h_summary->termination = RS_BA_FAILURE; keep_poses(); return RS_OK; }
    std::vector<double> cams(6 * C);
    for (size_t c = 0; c < C; c++) rs_pack_pose(m->kfs[(size_t)h_kfs[c]].pose, &cams[6 * c]);
    double *d_cams = nullptr, *d_pts = nullptr;
    int32_t *d_ptr = nullptr, *d_cam = nullptr, *d_list = nullptr;
    float *d_uv = nullptr, *d_oxyz = nullptr;
    if ((rc = rs_stage_upload(ctx, cams.data(), sizeof(double) * 6 * C, (void**)&d_cams))) return rc;
    if ((rc = rs_stage_alloc(ctx, sizeof(double) * 3 * Pf, (void**)&d_pts))) return rc;
    if ((rc = rs_stage_alloc(ctx, sizeof(int32_t) * (Pf + 1), (void**)&d_ptr))) return rc;
    if ((rc = rs_stage_alloc(ctx, sizeof(int32_t) * M, (void**)&d_cam))) return rc;
    if ((rc = rs_stage_alloc(ctx, sizeof(float) * 2 * M, (void**)&d_uv))) return rc;
    if ((rc = rs_stage_alloc(ctx, sizeof(int32_t) * Pf, (void**)&d_list))) return rc;
    if ((rc = rs_stage_alloc(ctx, sizeof(float) * 3 * Pf, (void**)&d_oxyz))) return rc;
    {
        rs_prof_scope ps(ctx, "K9_window_build");
        hipLaunchKernelGGL(k_win_fill, dim3(pb), dim3(256), 0, s, (int)P, (int)C, d_pid, d_off, m->d_obs_ptr, m->d_obs_kf, m->d_obs_desc, d_winkf,
                           m->d_pos, (const float2*)m->d_kp_pool, d_pts, d_list, d_ptr, d_cam, (float2*)d_uv, (int)Pf, (int)M);
    }
    rc = rs_bundle_adjust(ctx, (int)C, (int)Pf, (int)M, d_cams, h_free, d_pts, d_ptr, d_cam, d_uv, h_intrinsics, options, h_summary);
    if (rc) return rc;
    if (!h_summary->usable) { keep_poses(); return RS_OK; }
    rc = rs_ba_get_cameras(ctx, cams.data(), (int)C);
    if (rc) return rc;
    hipLaunchKernelGGL(k_win_writeback, dim3((int)((Pf + 255) / 256)), dim3(256), 0, s, (int)Pf, d_list, d_pts, m->d_pos, d_oxyz);
    std::vector<int32_t> list(Pf);
    std::vector<float> xyz(3 * Pf);
    if ((rc = rs_stage_download(ctx, d_list, sizeof(int32_t) * Pf, list.data()))) return rc;
    if ((rc = rs_stage_download(ctx, d_oxyz, sizeof(float) * 3 * Pf, xyz.data()))) return rc;
    if ((rc = rs_stage_sync(ctx))) return rc;
    for (size_t c = 0; c < C; c++) {
        MapKeyFrame& k = m->kfs[(size_t)h_kfs[c]];
        if (h_free[c]) { rs_unpack_pose(&cams[6 * c], k.pose); m->dirty_centres = true; }     // :363-368
        if (h_out_poses) memcpy(h_out_poses + 16 * c, k.pose, sizeof(float) * 16);
    }
    for (size_t q = 0; q < Pf; q++) memcpy(&m->pos[3 * (size_t)list[q]], &xyz[3 * q], sizeof(float) * 3);   // the mirror follows the device image
    const size_t nout = Pf < (size_t)(capacity > 0 ? capacity : 0) ? Pf : (size_t)(capacity > 0 ? capacity : 0);
    if (h_out_points && h_out_xyz && nout) {
        memcpy(h_out_points, list.data(), sizeof(int32_t) * nout);
        memcpy(h_out_xyz, xyz.data(), sizeof(float) * 3 * nout);
    }
    *h_n_points = (int)Pf;
    return RS_OK;
}

// ------------------------------------------------------------------------------------------------ loop closure
// optimization::pose_graph (src/Optimization.cpp:540-639) on the resident map.
extern "C" int rs_map_pose_graph(rs_context* ctx, rs_map* m, const rs_pose_graph_edge* h_loops, int n_loops, int four_dof,
                                 const double h_gravity[3], const rs_ba_options* options, float* h_out_poses,
                                 float* h_velocity_rotation, rs_ba_summary* h_summary)
{
    if (!ctx || !m || m->ctx != ctx || !h_summary || n_loops < 0 || (n_loops > 0 && !h_loops)) return RS_ERR_INVALID;
    const size_t KF = m->kfs.size(), P = m->alive.size();
    std::vector<float> before(16 * (KF ? KF : 1)), after(16 * (KF ? KF : 1));
    for (size_t k = 0; k < KF; k++) memcpy(&before[16 * k], m->kfs[k].pose, sizeof(float) * 16);
    int rc = rs_pose_graph((int)KF, before.data(), h_loops, n_loops, four_dof, h_gravity, options, after.data(),
                           h_velocity_rotation, h_summary, nullptr, 0, nullptr);
    if (rc) return rs_fail(ctx, rc, "rs_pose_graph rejected its arguments");
    if (h_out_poses && KF) memcpy(h_out_poses, after.data(), sizeof(float) * 16 * KF);
    if (!h_summary->usable) return RS_OK;
    RS_HIP(ctx, hipSetDevice(ctx->device));
    if ((rc = map_sync_device(m))) return rc;                 // the device image (positions, observation CSR) is current
    for (size_t k = 0; k < KF; k++) memcpy(m->kfs[k].pose, &after[16 * k], sizeof(float) * 16);      // Frame::set_pose, :503
    m->dirty_centres = true;
    if (P == 0) return RS_OK;
    if ((rc = rs_stage_begin(ctx))) return rc;
    float *d_before = nullptr, *d_after = nullptr;
    if ((rc = rs_stage_upload(ctx, before.data(), sizeof(float) * 16 * KF, (void**)&d_before))) return rc;
    if ((rc = rs_stage_upload(ctx, after.data(), sizeof(float) * 16 * KF, (void**)&d_after))) return rc;
    if ((rc = rs_transform_points(ctx, (int)P, m->d_obs_ptr, m->d_obs_kf, d_before, d_after, (int)KF, m->d_pos))) return rc;
    if ((rc = rs_stage_download(ctx, m->d_pos, sizeof(float) * 3 * P, m->pos.data()))) return rc;     // the mirror follows
    return rs_stage_sync(ctx);
}
