// Drop-in replacement of the reference's src/MapMatcher.cpp (keeps src/MapMatcher.h byte for byte).
// NOT COMPILED IN THIS REPO (needs the reference tree + OpenCV + Eigen); see rs_shim_common.h.
#include "MapMatcher.h"

#include <unordered_map>

#include "Camera.h"
#include "Frame.h"
#include "Map.h"
#include "MapPoint.h"
#include "rs_shim_common.h"

namespace slam {
namespace {

// Frame keeps its KD-tree private; the shim rebuilds the flattened tree from the keypoints with the
// library's host builder (same median-split rule, ties by keypoint index).  One build per call;
// a maintainer would cache it on the Frame.
struct FrameArrays {
    std::vector<float> kp;
    std::vector<uint8_t> matched;
    std::vector<int32_t> node_kp, left, right;
    int32_t root = -1;
    explicit FrameArrays(const Frame& f)
    {
        const size_t n = f.features().keypoints.size();
        kp.resize(2 * n); matched.resize(n); node_kp.resize(n); left.resize(n); right.resize(n);
        for (size_t i = 0; i < n; i++) { kp[2 * i] = f.keypoint(i).pt.x; kp[2 * i + 1] = f.keypoint(i).pt.y; matched[i] = f.is_matched(i); }
        rs_kdtree_build(kp.data(), (int)n, node_kp.data(), left.data(), right.data(), &root);
    }
};

std::vector<MapPointMatch> reproj(const Camera& camera, float max_distance, const Frame& frame,
                                  const std::vector<MapPoint*>& points, KeyFrame* required_observer, bool replace)
{
    using namespace rs_shim;
    const size_t N = frame.features().keypoints.size(), P = points.size();
    if (N == 0) return {};
    FrameArrays fa(frame);
    std::vector<float> pos(3 * P), centers;
    std::vector<uint8_t> eligible(P), pool;
    std::vector<int32_t> obs_ptr(P + 1, 0), obs_kf, obs_desc, pool_off;
    std::unordered_map<const KeyFrame*, int> kf_id;
    for (size_t p = 0; p < P; p++) {
        MapPoint* mp = points[p];
        bool run = mp != nullptr && !frame.is_matched(*mp);                               // :53, :121-123
        if (run && required_observer != nullptr && !mp->is_observed_by(required_observer)) run = false;   // :169
        eligible[p] = run;
        if (mp) for (int k = 0; k < 3; k++) pos[3 * p + k] = mp->position()[k];
        if (run)
            for (const auto& [kf, idx] : mp->observations()) {
                auto it = kf_id.find(kf);
                if (it == kf_id.end()) {
                    it = kf_id.emplace(kf, (int)kf_id.size()).first;
                    const Eigen::Vector3f c = kf->camera_center();
                    centers.insert(centers.end(), {c.x(), c.y(), c.z()});
                    pool_off.push_back((int32_t)(pool.size() / 32));
                    const cv::Mat& d = kf->features().descriptors;                        // N x 32 CV_8U, row-contiguous
                    pool.insert(pool.end(), d.datastart, d.dataend);
                }
                obs_kf.push_back(it->second);
                obs_desc.push_back(pool_off[it->second] + (int32_t)idx);
            }
        obs_ptr[p + 1] = (int32_t)obs_kf.size();
    }
    const cv::Mat& fd = frame.features().descriptors;
    DevBuf<float> d_kp(fa.kp), d_pos(pos), d_centers(centers);
    DevBuf<uint8_t> d_desc(std::vector<uint8_t>(fd.datastart, fd.dataend)), d_matched(fa.matched), d_elig(eligible), d_pool(pool);
    DevBuf<int32_t> d_nk(fa.node_kp), d_l(fa.left), d_r(fa.right), d_optr(obs_ptr), d_okf(obs_kf), d_odesc(obs_desc);
    rs_frame_view fv{};
    pose_to_row_major(frame.pose(), fv.pose);
    const Eigen::Matrix3f& K = camera.get_intrinsic_matrix();
    fv.fx = K(0, 0); fv.fy = K(1, 1); fv.cx = K(0, 2); fv.cy = K(1, 2);
    fv.width = camera.get_width(); fv.height = camera.get_height(); fv.n_keypoints = (int)N;
    fv.d_keypoints = d_kp.p; fv.d_descriptors = d_desc.p; fv.d_kp_matched = d_matched.p;
    fv.d_kd_node_kp = d_nk.p; fv.d_kd_left = d_l.p; fv.d_kd_right = d_r.p; fv.kd_root = fa.root;
    rs_map_view mv{(int)P, d_pos.p, d_elig.p, d_optr.p, d_okf.p, d_odesc.p, d_centers.p, d_pool.p};
    DevBuf<int32_t> pk(P), pd(P), pp(N), pdist(N), mkp(N), mpt(N), cnt(1);
    if (!ok(rs_reproj_match(context(), &fv, &mv, replace, (int)max_distance, pk.p, pd.p, pp.p, pdist.p, mkp.p, mpt.p, cnt.p), "rs_reproj_match"))
        return {};
    rs_context_synchronize(context());
    const int n = cnt.download(1)[0];
    const auto hk = mkp.download(n), hp = mpt.download(n);
    std::vector<MapPointMatch> out;
    for (int i = 0; i < n; i++) out.push_back(MapPointMatch{*points[hp[i]], (size_t)hk[i]});
    return out;
}

std::vector<MapPoint*> all_points(Map& map)
{
    std::vector<MapPoint*> v;
    for (auto& p : map) v.push_back(&p);      // map order, src/Map.h:66
    return v;
}

}  // namespace

MapMatcher::MapMatcher(const Camera& camera, float max_descriptor_distance, cv::NormTypes norm_type)
    : m_camera(camera), m_max_descriptor_distance(max_descriptor_distance), m_norm_type(norm_type) {}

std::vector<MapPointMatch> MapMatcher::match_map(const Frame& frame, Map& map) const { return match(frame, map, nullptr); }

std::vector<MapPointMatch> MapMatcher::match_key_frame(const Frame& frame, Map& map, KeyFrame* key_frame) const { return match(frame, map, key_frame); }

std::vector<MapPointMatch> MapMatcher::match_for_fuse(const Frame& frame, const std::vector<MapPoint*>& points) const
{
    return reproj(m_camera, m_max_descriptor_distance, frame, points, nullptr, true);
}

std::vector<MapPointMatch> MapMatcher::match(const Frame& frame, Map& map, KeyFrame* required_observer) const
{
    return reproj(m_camera, m_max_descriptor_distance, frame, all_points(map), required_observer, false);
}

std::vector<MapPointMatch> MapMatcher::match_descriptors(const Frame& frame, const KeyFrame& key_frame) const
{
    using namespace rs_shim;
    std::vector<MapPoint*> points;
    std::vector<uint8_t> train;
    for (const auto& m : key_frame.map_matches()) {                     // ascending keypoint order
        points.push_back(&m.point);
        const cv::Mat row = key_frame.descriptor(m.keypoint_index);
        train.insert(train.end(), row.datastart, row.dataend);
    }
    const cv::Mat& q = frame.features().descriptors;
    if (points.empty() || q.empty()) return {};
    const int nq = q.rows, nt = (int)points.size();
    DevBuf<uint8_t> dq(std::vector<uint8_t>(q.datastart, q.dataend)), dt(train);
    DevBuf<int32_t> mq(nq), mt(nq), cnt(1);
    if (!ok(rs_match_descriptors(context(), dq.p, nq, dt.p, nt, 1, (int)m_max_descriptor_distance, mq.p, mt.p, cnt.p,
                                 nullptr, nullptr, nullptr, nullptr), "rs_match_descriptors"))
        return {};
    rs_context_synchronize(context());
    const int n = cnt.download(1)[0];
    const auto hq = mq.download(n), ht = mt.download(n);
    std::vector<MapPointMatch> out;
    for (int i = 0; i < n; i++) out.push_back({*points[(size_t)ht[i]], (size_t)hq[i]});
    return out;
}

}  // namespace slam
