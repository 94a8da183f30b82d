// slam_host.cpp — marshalling of the host mirror (slam_host.h) onto the C-ABI of librsgpu.so.
// Pointer graph -> SoA, upload, ONE C-ABI call per interface function, download.
#include "slam_host.h"

#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <cstdint>
#include <unordered_map>

namespace slam {

// ------------------------------------------------------------------ device helpers
namespace {

void hip_ok(hipError_t e, const char* what)
{
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

// Device arrays of one interface call, carved from the context's staging pool (rs_stage_*, include/rsgpu.h): no
// hipMalloc / hipFree per call, uploads go host -> pinned -> device asynchronously on the context stream, results come
// back with ONE synchronisation per call (fetch ... fetch, stage_sync).
struct StageScope {
    StageScope() { if (rs_stage_begin(Session::get().ctx()) != RS_OK) throw std::runtime_error("rs_stage_begin"); }
};

template <typename T>
class DevBuf {
  public:
    explicit DevBuf(size_t n) : m_n(n)
    {
        if (rs_stage_alloc(Session::get().ctx(), sizeof(T) * (n ? n : 1), (void**)&m_p) != RS_OK) throw std::runtime_error("rs_stage_alloc");
    }
    explicit DevBuf(const std::vector<T>& h) : m_n(h.size())
    {
        if (rs_stage_upload(Session::get().ctx(), h.data(), sizeof(T) * h.size(), (void**)&m_p) != RS_OK) throw std::runtime_error("rs_stage_upload");
    }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    // registers an asynchronous read-back of the first n entries; the vector is filled by stage_sync()
    std::vector<T> fetch(size_t n) const
    {
        std::vector<T> h(n);
        if (n && rs_stage_download(Session::get().ctx(), m_p, sizeof(T) * n, h.data()) != RS_OK) throw std::runtime_error("rs_stage_download");
        return h;           // moved out: the heap block registered above stays where it is
    }
    std::vector<T> download(size_t n) const
    {
        std::vector<T> h = fetch(n);
        stage_sync();
        return h;
    }
    static void stage_sync() { if (rs_stage_sync(Session::get().ctx()) != RS_OK) throw std::runtime_error("rs_stage_sync"); }
    T* get() const { return m_p; }
    size_t size() const { return m_n; }

  private:
    T* m_p = nullptr;
    size_t m_n = 0;
};
inline void stage_sync() { DevBuf<int>::stage_sync(); }

// The reference has no error codes: a failed C-ABI call is logged to stdout like the
// reference logs, and the interface function returns "empty / false".  Nothing throws across it.
bool rs_ok(int rc, const char* what)
{
    if (rc == RS_OK) return true;
    std::printf("%s failed (status %d): %s\n", what, rc, rs_last_error(Session::get().ctx()));
    return false;
}

rs_ba_summary g_summary{};

// Pointer -> dense index, for flattening the reference's pointer graph (MapPoint has no id: src/MapPoint.h:12-39).  Open
// addressing over a power-of-two table of {key, value, generation} entries (one cache line per probe), multiplicative hash
// of the address; the table is kept between calls and "cleared" by bumping the generation — std::unordered_map (a node per
// entry, cleared and rebuilt per call) spent 50 - 100 ns per operation and build_local_window + bundle_adjust make ~10^5 of
// them per key frame; allocating and zeroing a fresh 0.8 MB table per call cost another 0.2 ms.
class PtrIndex {
  public:
    // a table for about `expected` distinct keys (it grows by itself beyond that)
    explicit PtrIndex(std::vector<char>& storage, size_t expected) : m_store(storage)
    {
        size_t cap = 64;
        while (cap < 2 * expected + 2) cap <<= 1;
        Header* h = header();
        if (!h || h->cap < cap) {
            m_store.assign(sizeof(Header) + cap * sizeof(Entry), 0);
            h = header();
            h->cap = cap; h->gen = 0;
        }
        if (++h->gen == 0) {                                   // generation wrapped: really clear
            std::memset(entries(), 0, h->cap * sizeof(Entry));
            h->gen = 1;
        }
        m_gen = h->gen; m_mask = h->cap - 1; m_e = entries(); m_used = 0;
    }
    // index of p, inserting it with value `fresh` if absent; *inserted says which
    int find_or_insert(const void* p, int fresh, bool* inserted)
    {
        size_t h = slot(p);
        while (m_e[h].gen == m_gen && m_e[h].key != p) h = (h + 1) & m_mask;
        if (m_e[h].gen == m_gen) { *inserted = false; return m_e[h].val; }
        m_e[h].key = p; m_e[h].val = fresh; m_e[h].gen = m_gen; *inserted = true;
        if (2 * ++m_used > m_mask) grow();
        return fresh;
    }
    int find(const void* p) const
    {
        size_t h = slot(p);
        while (m_e[h].gen == m_gen && m_e[h].key != p) h = (h + 1) & m_mask;
        return m_e[h].gen == m_gen ? m_e[h].val : -1;
    }

  private:
    struct Entry { const void* key; int val; unsigned gen; };
    struct Header { size_t cap; unsigned gen; unsigned pad; };
    Header* header() { return m_store.size() >= sizeof(Header) ? reinterpret_cast<Header*>(m_store.data()) : nullptr; }
    Entry* entries() { return reinterpret_cast<Entry*>(m_store.data() + sizeof(Header)); }
    size_t slot(const void* p) const { return (size_t)(((uintptr_t)p >> 4) * 0x9E3779B97F4A7C15ull >> 20) & m_mask; }
    void grow()
    {
        std::vector<Entry> live;
        for (size_t i = 0; i <= m_mask; i++)
            if (m_e[i].gen == m_gen) live.push_back(m_e[i]);
        const size_t cap = 2 * (m_mask + 1);
        m_store.assign(sizeof(Header) + cap * sizeof(Entry), 0);
        Header* h = header();
        h->cap = cap; h->gen = 1;
        m_gen = 1; m_mask = cap - 1; m_e = entries();
        for (const Entry& e : live) {
            size_t q = slot(e.key);
            while (m_e[q].gen == m_gen) q = (q + 1) & m_mask;
            m_e[q] = Entry{e.key, e.val, m_gen};
        }
    }
    std::vector<char>& m_store;
    Entry* m_e = nullptr;
    size_t m_mask = 0, m_used = 0;
    unsigned m_gen = 0;
};
// storage of the tables, kept between calls (the callers are single-threaded: SURVEY.md 8b)
std::vector<char> g_pid_store, g_fid_store;

}  // namespace

Session::Session()
{
    const int rc = rs_context_create(0, &m_ctx);
    if (rc != RS_OK) throw std::runtime_error("rs_context_create failed (no gfx950 GPU? there is no CPU fallback)");
}
Session::~Session() { rs_context_destroy(m_ctx); }
Session& Session::get()
{
    static Session s;
    return s;
}

// ------------------------------------------------------------------------ data model
Frame::Frame(int index, ExtractedFeatures features) : m_index((size_t)index), m_features(std::move(features))
{
    const size_t n = m_features.keypoints.size();
    m_map_matches.assign(n, nullptr);
    std::vector<float> kp(2 * n);
    for (size_t i = 0; i < n; i++) { kp[2 * i] = m_features.keypoints[i].pt.x; kp[2 * i + 1] = m_features.keypoints[i].pt.y; }
    m_kd_node_kp.resize(n); m_kd_left.resize(n); m_kd_right.resize(n);
    int32_t root = -1;
    rs_kdtree_build(kp.data(), (int)n, m_kd_node_kp.data(), m_kd_left.data(), m_kd_right.data(), &root);   // src/Frame.cpp:8-15
    m_kd_root = root;
}

Vec3f Frame::camera_center() const
{
    const Mat4f& T = m_pose;
    Vec3f c;
    c.x = (-T[0] * T[3] + -T[4] * T[7]) + -T[8] * T[11];
    c.y = (-T[1] * T[3] + -T[5] * T[7]) + -T[9] * T[11];
    c.z = (-T[2] * T[3] + -T[6] * T[7]) + -T[10] * T[11];
    return c;
}

void Frame::add_map_match(const MapPointMatch& m)
{
    MapPoint* previous = m_map_matches[m.keypoint_index];
    if (previous == &m.point) return;
    if (previous == nullptr) m_num++;
    else m_matched_points.erase(previous);
    for (size_t i = 0; i < m_map_matches.size(); i++) {
        if (m_map_matches[i] != &m.point || i == m.keypoint_index) continue;
        m_map_matches[i] = nullptr;
        if (m_num > 0) m_num--;
    }
    m_map_matches[m.keypoint_index] = &m.point;
    m_matched_points.insert(&m.point);
}

bool Frame::is_matched(const MapPoint& p) const
{
    return m_matched_points.count(&p) != 0;                     // (src/Frame.cpp:148-151)
}

std::vector<MapPointMatch> Frame::map_matches() const
{
    std::vector<MapPointMatch> out;
    for (size_t i = 0; i < m_map_matches.size(); i++)
        if (m_map_matches[i]) out.push_back(MapPointMatch{*m_map_matches[i], i});
    return out;
}

bool MapPoint::is_observed_by(const KeyFrame* kf) const
{
    for (const auto& o : m_obs)
        if (o.first == kf) return true;
    return false;
}

void Map::associate(KeyFrame& kf, MapPoint& point, size_t keypoint_index)
{
    bool found = false;
    for (auto& o : point.m_obs)
        if (o.first == &kf) { o.second = keypoint_index; found = true; }
    if (!found) point.m_obs.emplace_back(&kf, keypoint_index);
    kf.add_map_match(MapPointMatch{point, keypoint_index});
}

// ------------------------------------------------------------------------ MapMatcher
MapMatcher::MapMatcher(const Camera& camera, float max_descriptor_distance, NormTypes norm_type)
    : m_camera(camera), m_max_descriptor_distance(max_descriptor_distance), m_norm_type(norm_type)
{
}

std::vector<MapPointMatch> MapMatcher::match_map(const Frame& frame, Map& map) const
{
    std::vector<MapPoint*> pts(map.size());
    for (size_t i = 0; i < map.size(); i++) pts[i] = &map[i];
    return match(frame, pts, nullptr, false);
}

std::vector<MapPointMatch> MapMatcher::match_key_frame(const Frame& frame, Map& map, KeyFrame* key_frame) const
{
    std::vector<MapPoint*> pts(map.size());
    for (size_t i = 0; i < map.size(); i++) pts[i] = &map[i];
    return match(frame, pts, key_frame, false);
}

std::vector<MapPointMatch> MapMatcher::match_for_fuse(const Frame& frame, const std::vector<MapPoint*>& points) const
{
    return match(frame, points, nullptr, true);
}

// src/MapMatcher.cpp:45-98,117-127,165-175 -> rs_reproj_match
std::vector<MapPointMatch> MapMatcher::match(const Frame& frame, const std::vector<MapPoint*>& points,
                                             const KeyFrame* required_observer, bool replace) const
{
    rs_context* ctx = Session::get().ctx();
    const size_t N = frame.features().keypoints.size(), P = points.size();
    if (N == 0) return {};
    // frame side
    std::vector<float> kp(2 * N);
    std::vector<uint8_t> matched(N);
    for (size_t i = 0; i < N; i++) {
        kp[2 * i] = frame.keypoint(i).pt.x; kp[2 * i + 1] = frame.keypoint(i).pt.y;
        matched[i] = frame.is_matched(i) ? 1 : 0;
    }
    // map side: positions, eligibility (the pointer-set tests of :53, :121-123, :169), observation CSR,
    // keyframe table and descriptor pool
    std::vector<float> pos(3 * P), centers;
    std::vector<uint8_t> eligible(P), pool;
    std::vector<int32_t> obs_ptr(P + 1, 0), obs_kf, obs_desc;
    std::unordered_map<const KeyFrame*, int> kf_id;
    std::vector<int> kf_pool_off;
    for (size_t p = 0; p < P; p++) {
        const MapPoint* mp = points[p];
        bool ok = mp != nullptr;
        if (ok && frame.is_matched(*mp)) ok = false;
        if (ok && required_observer && !mp->is_observed_by(required_observer)) ok = false;
        eligible[p] = ok ? 1 : 0;
        if (mp) { pos[3 * p] = mp->position().x; pos[3 * p + 1] = mp->position().y; pos[3 * p + 2] = mp->position().z; }
        if (ok) {
            for (const auto& o : mp->observations()) {
                auto it = kf_id.find(o.first);
                if (it == kf_id.end()) {
                    it = kf_id.emplace(o.first, (int)kf_id.size()).first;
                    const Vec3f c = o.first->camera_center();
                    centers.insert(centers.end(), {c.x, c.y, c.z});
                    kf_pool_off.push_back((int)(pool.size() / 32));
                    const auto& d = o.first->features().descriptors;
                    pool.insert(pool.end(), d.begin(), d.end());
                }
                obs_kf.push_back(it->second);
                obs_desc.push_back(kf_pool_off[it->second] + (int)o.second);
            }
        }
        obs_ptr[p + 1] = (int32_t)obs_kf.size();
    }
    StageScope stage;
    DevBuf<float> d_kp(kp), d_pos(pos), d_centers(centers);
    DevBuf<uint8_t> d_desc(frame.features().descriptors), d_matched(matched), d_elig(eligible), d_pool(pool);
    DevBuf<int32_t> d_nk(frame.kd_node_kp()), d_l(frame.kd_left()), d_r(frame.kd_right()), d_optr(obs_ptr), d_okf(obs_kf),
        d_odesc(obs_desc);
    rs_frame_view fv{};
    std::memcpy(fv.pose, frame.pose().data(), sizeof fv.pose);
    fv.fx = m_camera.fx(); fv.fy = m_camera.fy(); fv.cx = m_camera.cx(); fv.cy = m_camera.cy();
    fv.width = m_camera.get_width(); fv.height = m_camera.get_height();
    fv.n_keypoints = (int)N; fv.d_keypoints = d_kp.get(); fv.d_descriptors = d_desc.get(); fv.d_kp_matched = d_matched.get();
    fv.d_kd_node_kp = d_nk.get(); fv.d_kd_left = d_l.get(); fv.d_kd_right = d_r.get(); fv.kd_root = frame.kd_root();
    rs_map_view mv{};
    mv.n_points = (int)P; mv.d_positions = d_pos.get(); mv.d_eligible = d_elig.get(); mv.d_obs_ptr = d_optr.get();
    mv.d_obs_kf = d_okf.get(); mv.d_obs_desc = d_odesc.get(); mv.d_kf_centers = d_centers.get(); mv.d_desc_pool = d_pool.get();
    DevBuf<int32_t> pk(P), pd(P), pp(N), pdist(N), mkp(N), mpt(N), cnt(1);
    if (!rs_ok(rs_reproj_match(ctx, &fv, &mv, replace ? 1 : 0, (int)m_max_descriptor_distance, pk.get(), pd.get(), pp.get(),
                               pdist.get(), mkp.get(), mpt.get(), cnt.get()), "rs_reproj_match"))
        return {};
    const auto hn = cnt.fetch(1);
    const auto hk = mkp.fetch(N), hp = mpt.fetch(N);          // at most N matches: one read-back, one synchronisation
    stage_sync();
    const int n = hn[0];
    std::vector<MapPointMatch> out;
    for (int i = 0; i < n; i++) out.push_back(MapPointMatch{*points[(size_t)hp[i]], (size_t)hk[i]});
    return out;
}

// src/MapMatcher.cpp:129-163 -> rs_match_descriptors
std::vector<MapPointMatch> MapMatcher::match_descriptors(const Frame& frame, const KeyFrame& key_frame) const
{
    rs_context* ctx = Session::get().ctx();
    const auto km = key_frame.map_matches();                  // ascending keypoint order
    std::vector<uint8_t> train;
    for (const auto& m : km) {
        const uint8_t* row = key_frame.features().descriptors.data() + 32 * m.keypoint_index;
        train.insert(train.end(), row, row + 32);
    }
    const int nq = (int)frame.features().keypoints.size(), nt = (int)km.size();
    if (nt == 0 || nq == 0) return {};                        // :139-141
    StageScope stage;
    DevBuf<uint8_t> dq(frame.features().descriptors), dt(train);
    DevBuf<int32_t> mq((size_t)nq), mt((size_t)nq), cnt(1);
    if (!rs_ok(rs_match_descriptors(ctx, dq.get(), nq, dt.get(), nt, 1, (int)m_max_descriptor_distance, mq.get(), mt.get(),
                                    cnt.get(), nullptr, nullptr, nullptr, nullptr), "rs_match_descriptors"))
        return {};
    const auto hn = cnt.fetch(1);
    const auto hq = mq.fetch((size_t)nq), ht = mt.fetch((size_t)nq);
    stage_sync();
    const int n = hn[0];
    std::vector<MapPointMatch> out;
    for (int i = 0; i < n; i++) out.push_back(MapPointMatch{km[(size_t)ht[i]].point, (size_t)hq[i]});   // :159-160
    return out;
}

// --------------------------------------------------------------------- triangulation
namespace triangulation {

std::pair<std::vector<Vec2f>, std::vector<Vec2f>> get_matching_points(const ExtractedFeatures& f1, const ExtractedFeatures& f2,
                                                                      const std::vector<FeatureMatch>& matches)
{
    std::vector<Vec2f> p1, p2;
    for (const auto& m : matches) {                            // src/Triangulation.cpp:11-26
        p1.push_back(f1.keypoints[m.train_index].pt);
        p2.push_back(f2.keypoints[m.query_index].pt);
    }
    return {p1, p2};
}

std::vector<TriangulatedPoint> triangulate_points(const std::vector<Vec2f>& points1, const std::vector<Vec2f>& points2,
                                                  const Mat4f& pose1, const Mat4f& pose2, const Camera& camera,
                                                  float min_parallax_cosine, float max_reprojection_error)
{
    if (points1.empty() || points2.empty()) return {};         // :46-48
    rs_context* ctx = Session::get().ctx();
    const size_t n = points1.size();
    // host in, host out: Vec2f is two packed floats, so the vectors ARE the [n][2] arrays.  Up to 256 correspondences
    // (Mapper::triangulate_tracks calls this with ONE) take a single launch with the result in pinned memory.
    static_assert(sizeof(Vec2f) == 2 * sizeof(float), "Vec2f must be two packed floats");
    std::vector<int32_t> hi(n);
    std::vector<float> hx(3 * n);
    int m = 0;
    const float K[4] = {camera.fx(), camera.fy(), camera.cx(), camera.cy()};
    if (!rs_ok(rs_triangulate_host(ctx, &points1[0].x, &points2[0].x, (int)n, pose1.data(), pose2.data(), K, min_parallax_cosine,
                                   max_reprojection_error, hi.data(), hx.data(), &m), "rs_triangulate_host"))
        return {};
    std::vector<TriangulatedPoint> out((size_t)m);
    for (int i = 0; i < m; i++) out[(size_t)i] = TriangulatedPoint{Vec3f{hx[3 * i], hx[3 * i + 1], hx[3 * i + 2]}, hi[(size_t)i]};
    return out;
}

std::vector<TriangulatedPoint> triangulate_points(const Frame& frame1, const Frame& frame2,
                                                  const std::vector<FeatureMatch>& matches, const Camera& camera)
{
    auto pts = get_matching_points(frame1.features(), frame2.features(), matches);   // :28-35
    return triangulate_points(pts.first, pts.second, frame1.pose(), frame2.pose(), camera);
}

}  // namespace triangulation

// ---------------------------------------------------------------------- track triangulation
namespace tracks {

// src/Mapper.cpp:246-305 -> rs_triangulate_tracks
Selection select_track_points(const KeyFrame& key_frame, const std::vector<Track>& tracks,
                              const std::vector<Mat4f>& trajectory_poses, const Camera& camera, size_t min_new_points)
{
    Selection out;
    const size_t T = tracks.size();
    if (T == 0) return out;
    rs_context* ctx = Session::get().ctx();
    std::vector<float> track_uv(2 * T), sight_uv, poses(16 * (trajectory_poses.size() + 1));
    std::vector<uint8_t> skip(T);
    std::vector<int32_t> sight_ptr(T + 1, 0), sight_pose;
    for (size_t t = 0; t < T; t++) {
        const Track& tr = tracks[t];
        skip[t] = (key_frame.is_matched(tr.keypoint_index) || tr.sightings.empty()) ? 1 : 0;     // :248-250
        const Vec2f px = key_frame.keypoint(tr.keypoint_index).pt;
        track_uv[2 * t] = px.x; track_uv[2 * t + 1] = px.y;
        for (const auto& sg : tr.sightings) {
            sight_pose.push_back((int32_t)sg.frame_index);
            sight_uv.push_back(sg.pixel.x); sight_uv.push_back(sg.pixel.y);
        }
        sight_ptr[t + 1] = (int32_t)sight_pose.size();
    }
    for (size_t i = 0; i < trajectory_poses.size(); i++) std::memcpy(&poses[16 * i], trajectory_poses[i].data(), 64);
    const int kf_pose = (int)trajectory_poses.size();                                         // key_frame.pose(), :257
    std::memcpy(&poses[16 * (size_t)kf_pose], key_frame.pose().data(), 64);
    if (sight_pose.empty()) { sight_pose.push_back(0); sight_uv.resize(2); }
    StageScope stage;
    DevBuf<float> d_tuv(track_uv), d_suv(sight_uv), d_poses(poses), d_xyz(3 * T), d_pc(T), d_rc(T);
    DevBuf<uint8_t> d_skip(skip), d_status(T);
    DevBuf<int32_t> d_sptr(sight_ptr), d_spose(sight_pose), d_acc(T), d_inc(T), d_cnt(3);
    const float K[4] = {camera.fx(), camera.fy(), camera.cx(), camera.cy()};
    // the rotation-dependent requirement per first-sighting pose on the HOST (libm as the reference calls it, :281-288)
    std::vector<float> required((size_t)kf_pose + 1);
    if (!rs_ok(rs_parallax_requirements(poses.data(), kf_pose + 1, kf_pose, TRACK_MIN_PARALLAX_COSINE, ROTATION_PARALLAX_FACTOR,
                                        required.data()), "rs_parallax_requirements"))
        return out;
    DevBuf<float> d_req(required);
    if (!rs_ok(rs_triangulate_tracks(ctx, (int)T, d_tuv.get(), d_skip.get(), d_sptr.get(), d_spose.get(), d_suv.get(),
                                     d_poses.get(), kf_pose + 1, kf_pose, K, ANY_PARALLAX_COSINE,
                                     TRACK_MAX_REPROJECTION_ERROR, TRACK_MIN_PARALLAX_COSINE, ROTATION_PARALLAX_FACTOR,
                                     (int)min_new_points, d_status.get(), d_xyz.get(), d_pc.get(), d_rc.get(), d_acc.get(),
                                     d_inc.get(), d_cnt.get(), d_req.get()), "rs_triangulate_tracks"))
        return out;
    const auto cnt = d_cnt.fetch(3);
    const auto acc = d_acc.fetch(T);
    const auto inc = d_inc.fetch(T);
    const auto xyz = d_xyz.fetch(3 * T);
    const auto pc = d_pc.fetch(T), rc = d_rc.fetch(T);
    stage_sync();
    for (int i = 0; i < cnt[0]; i++) {
        const int32_t t = acc[(size_t)i];
        out.accepted.push_back(Candidate{(size_t)t, Vec3f{xyz[3 * (size_t)t], xyz[3 * (size_t)t + 1], xyz[3 * (size_t)t + 2]},
                                         tracks[(size_t)t].keypoint_index, pc[(size_t)t], rc[(size_t)t]});
    }
    out.topped_up = (size_t)cnt[1];
    for (int i = 0; i < cnt[2]; i++) out.inconsistent.push_back((size_t)inc[(size_t)i]);
    return out;
}

// src/Mapper.cpp:410-419 (+ src/Slam.cpp:302-317) -> rs_point_errors
CullResult point_errors(const std::vector<MapPoint*>& points, const Camera& camera, float max_mean_error)
{
    CullResult out;
    const size_t P = points.size();
    if (P == 0) return out;
    rs_context* ctx = Session::get().ctx();
    std::vector<float> pos(3 * P), uv, poses;
    std::vector<int32_t> obs_ptr(P + 1, 0), obs_pose;
    std::vector<const KeyFrame*> frames;                      // pose table: one entry per observing key frame
    for (size_t p = 0; p < P; p++) {
        const Vec3f& X = points[p]->position();
        pos[3 * p] = X.x; pos[3 * p + 1] = X.y; pos[3 * p + 2] = X.z;
        for (const auto& ob : points[p]->observations()) {
            size_t f = 0;
            while (f < frames.size() && frames[f] != ob.first) f++;
            if (f == frames.size()) { frames.push_back(ob.first); poses.insert(poses.end(), ob.first->pose().begin(), ob.first->pose().end()); }
            obs_pose.push_back((int32_t)f);
            const Vec2f px = ob.first->keypoint(ob.second).pt;
            uv.push_back(px.x); uv.push_back(px.y);
        }
        obs_ptr[p + 1] = (int32_t)obs_pose.size();
    }
    if (obs_pose.empty()) { out.mean_error.assign(P, 0.0f); return out; }
    StageScope stage;
    DevBuf<float> d_pos(pos), d_uv(uv), d_poses(poses), d_mean(P);
    DevBuf<int32_t> d_ptr(obs_ptr), d_op(obs_pose), d_idx(P), d_cnt(1);
    DevBuf<uint8_t> d_cull(P);
    DevBuf<double> d_sums(2);
    const float K[4] = {camera.fx(), camera.fy(), camera.cx(), camera.cy()};
    if (!rs_ok(rs_point_errors(ctx, (int)P, d_pos.get(), d_ptr.get(), d_op.get(), d_uv.get(), d_poses.get(), (int)frames.size(), K,
                               max_mean_error, d_mean.get(), d_cull.get(), d_idx.get(), d_cnt.get(), d_sums.get()), "rs_point_errors"))
        return out;
    out.mean_error = d_mean.fetch(P);
    const auto hn = d_cnt.fetch(1);
    const auto hidx = d_idx.fetch(P);
    const auto sums = d_sums.fetch(2);
    stage_sync();
    for (int i = 0; i < hn[0]; i++) out.to_remove.push_back((size_t)hidx[(size_t)i]);
    out.error_sum = sums[0];
    out.observations = (size_t)sums[1];
    return out;
}

}  // namespace tracks

// ---------------------------------------------------------------------- optimisation
namespace optimization {

const rs_ba_summary& last_summary() { return g_summary; }

// src/Optimization.cpp:194-267 (vision-only) -> rs_refine_pose
bool refine_pose(Frame& frame, const Camera& camera)
{
    rs_context* ctx = Session::get().ctx();
    std::vector<double> pts;
    std::vector<float> uv;
    for (const auto& m : frame.map_matches()) {
        if (m.point.observations().size() < 2) continue;       // MIN_OBSERVATIONS_TO_OPTIMIZE, :98,:206
        pts.insert(pts.end(), {m.point.position().x, m.point.position().y, m.point.position().z});
        uv.insert(uv.end(), {frame.keypoint(m.keypoint_index).pt.x, frame.keypoint(m.keypoint_index).pt.y});
    }
    if (uv.empty()) return false;                              // :227-229
    double cam[6];
    rs_pack_pose(frame.pose().data(), cam);
    StageScope stage;
    DevBuf<double> dp(pts);
    DevBuf<float> duv(uv);
    const float K[4] = {camera.fx(), camera.fy(), camera.cx(), camera.cy()};
    if (!rs_ok(rs_refine_pose(ctx, cam, dp.get(), duv.get(), (int)(uv.size() / 2), K, nullptr, &g_summary), "rs_refine_pose")) return false;
    std::printf("refine_pose: iterations %d, cost %.6e -> %.6e, termination %d\n", g_summary.iterations,
                g_summary.initial_cost, g_summary.final_cost, g_summary.termination);
    if (!g_summary.usable) { std::printf("Optimization rejected, unusable or non-improving solution\n"); return false; }
    Mat4f T;
    rs_unpack_pose(cam, T.data());
    frame.set_pose(T);
    return true;
}

// src/Optimization.cpp:269-374 (vision-only) -> rs_bundle_adjust
bool bundle_adjust(const std::vector<FrameConfig>& frames, const Camera& camera, Map&)
{
    rs_context* ctx = Session::get().ctx();
    const size_t C = frames.size();
    std::vector<double> cams(6 * C);
    std::vector<uint8_t> cam_free(C);
    for (size_t c = 0; c < C; c++) { rs_pack_pose(frames[c].frame->pose().data(), &cams[6 * c]); cam_free[c] = frames[c].optimize ? 1 : 0; }
    // free points: matched by an optimised frame, >= 2 observations (:287-302), in first-seen order
    std::vector<MapPoint*> free_pts;
    std::vector<std::vector<MapPointMatch>> matches(C);          // (map_matches() builds a list: once per frame)
    size_t n_matches = 0;
    for (size_t c = 0; c < C; c++) { matches[c] = frames[c].frame->map_matches(); n_matches += matches[c].size(); }
    PtrIndex pid(g_pid_store, n_matches / 2 + 16);
    for (size_t c = 0; c < C; c++) {
        if (!frames[c].optimize) continue;
        for (const auto& m : matches[c]) {
            if (m.point.observations().size() < 2) continue;
            bool fresh = false;
            pid.find_or_insert(&m.point, (int)free_pts.size(), &fresh);
            if (fresh) free_pts.push_back(&m.point);
        }
    }
    const size_t P = free_pts.size();
    // residual blocks: every listed frame (free or fixed) x its matched free points (:304-315), CSR by point, frames in
    // list order within a point: one pass resolves the matches to point ids, a count / prefix / fill builds the CSR
    struct Hit { int point, cam; Vec2f uv; };
    std::vector<Hit> hits;
    hits.reserve(n_matches);
    std::vector<int32_t> obs_ptr(P + 1, 0);
    for (size_t c = 0; c < C; c++)
        for (const auto& m : matches[c]) {
            const int id = pid.find(&m.point);
            if (id < 0) continue;
            hits.push_back(Hit{id, (int)c, frames[c].frame->keypoint(m.keypoint_index).pt});
            obs_ptr[(size_t)id + 1]++;
        }
    for (size_t p = 0; p < P; p++) obs_ptr[p + 1] += obs_ptr[p];
    std::vector<int32_t> obs_cam(hits.size()), fill(obs_ptr.begin(), obs_ptr.end() - 1);
    std::vector<float> obs_uv(2 * hits.size());
    for (const Hit& h : hits) {                                  // (hits are in frame order: so is every point's slice)
        const size_t o = (size_t)fill[(size_t)h.point]++;
        obs_cam[o] = h.cam; obs_uv[2 * o] = h.uv.x; obs_uv[2 * o + 1] = h.uv.y;
    }
    std::vector<double> pts(3 * P);
    for (size_t p = 0; p < P; p++) {
        pts[3 * p] = free_pts[p]->position().x; pts[3 * p + 1] = free_pts[p]->position().y; pts[3 * p + 2] = free_pts[p]->position().z;
    }
    if (P == 0 || obs_cam.empty()) return false;
    StageScope stage;
    DevBuf<double> dc(cams), dp(pts);
    DevBuf<int32_t> dptr(obs_ptr), dcam(obs_cam);
    DevBuf<float> duv(obs_uv);
    const float K[4] = {camera.fx(), camera.fy(), camera.cx(), camera.cy()};
    if (!rs_ok(rs_bundle_adjust(ctx, (int)C, (int)P, (int)obs_cam.size(), dc.get(), cam_free.data(), dp.get(), dptr.get(), dcam.get(),
                                duv.get(), K, nullptr, &g_summary), "rs_bundle_adjust"))
        return false;
    std::printf("bundle_adjust: iterations %d, cost %.6e -> %.6e, termination %d\n", g_summary.iterations,
                g_summary.initial_cost, g_summary.final_cost, g_summary.termination);
    if (!g_summary.usable) { std::printf("Optimization rejected, unusable or non-improving solution\n"); return false; }
    std::vector<double> hc(6 * C);
    rs_ba_get_cameras(ctx, hc.data(), (int)C);                  // pinned mirror written by the solve's last kernel
    const auto hp = dp.download(3 * P);
    for (size_t c = 0; c < C; c++)
        if (frames[c].optimize) { Mat4f T; rs_unpack_pose(&hc[6 * c], T.data()); frames[c].frame->set_pose(T); }   // :363-368
    for (size_t p = 0; p < P; p++) free_pts[p]->set_position(Vec3f{(float)hp[3 * p], (float)hp[3 * p + 1], (float)hp[3 * p + 2]});
    return true;
}

// src/LocalWindow.cpp:10-52 -> rs_build_local_window
std::vector<FrameConfig> build_local_window(const std::vector<std::shared_ptr<KeyFrame>>& key_frames, Frame& new_frame,
                                            size_t window_size, bool fix_oldest)
{
    const int n = (int)key_frames.size();
    int new_index = -1;
    PtrIndex fid(g_fid_store, (size_t)n);
    size_t n_matches = 0;
    for (int i = 0; i < n; i++) {
        bool fresh;
        fid.find_or_insert(static_cast<const Frame*>(key_frames[(size_t)i].get()), i, &fresh);
        if (key_frames[(size_t)i].get() == &new_frame) new_index = i;
        n_matches += key_frames[(size_t)i]->num_map_matches();
    }
    PtrIndex pid(g_pid_store, (n_matches + new_frame.num_map_matches()) / 2 + 16);
    std::vector<const MapPoint*> pts;
    std::vector<int32_t> frame_ptr((size_t)n + 2, 0), frame_pt;
    frame_pt.reserve(n_matches + new_frame.num_map_matches());
    auto add_frame = [&](const Frame& f, int slot) {
        const std::vector<MapPoint*>& tab = f.match_table();   // ascending keypoint index, as map_matches()
        for (size_t k = 0; k < tab.size(); k++) {
            const MapPoint* mp = tab[k];
            if (!mp) continue;
            bool fresh = false;
            const int id = pid.find_or_insert(mp, (int)pts.size(), &fresh);
            if (fresh) pts.push_back(mp);
            frame_pt.push_back(id);
        }
        frame_ptr[(size_t)slot + 1] = (int32_t)frame_pt.size();
    };
    for (int i = 0; i < n; i++) add_frame(*key_frames[(size_t)i], i);
    if (new_index < 0) add_frame(new_frame, n); else frame_ptr[(size_t)n + 1] = frame_ptr[(size_t)n];
    std::vector<int32_t> pt_ptr(pts.size() + 1, 0), pt_obs;
    pt_obs.reserve(frame_pt.size());
    for (size_t p = 0; p < pts.size(); p++) {
        if (p + 8 < pts.size()) __builtin_prefetch(pts[p + 8]);                   // (the points are scattered heap objects)
        if (p + 4 < pts.size()) __builtin_prefetch(pts[p + 4]->observations().data());
        for (const auto& o : pts[p]->observations()) {
            const int f = fid.find(static_cast<const Frame*>(o.first));
            if (f >= 0) pt_obs.push_back(f);
        }
        pt_ptr[p + 1] = (int32_t)pt_obs.size();
    }
    std::vector<int32_t> of((size_t)n + 1);
    std::vector<uint8_t> oo((size_t)n + 1);
    int32_t cnt = 0;
    if (rs_build_local_window(n, new_index, (int)window_size, fix_oldest ? 1 : 0, frame_ptr.data(), frame_pt.data(), pt_ptr.data(),
                              pt_obs.data(), of.data(), oo.data(), &cnt) != RS_OK)
        return {};
    std::vector<FrameConfig> out;
    for (int i = 0; i < cnt; i++) {
        Frame* f = of[(size_t)i] == n ? &new_frame : key_frames[(size_t)of[(size_t)i]].get();
        out.push_back(FrameConfig{oo[(size_t)i] != 0, f});
    }
    return out;
}

}  // namespace optimization
}  // namespace slam
