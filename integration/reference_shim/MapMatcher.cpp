// Drop-in replacement of the reference's src/MapMatcher.cpp (keeps src/MapMatcher.h byte for byte).
// Compiled only in the reference's tree (OpenCV + Eigen); syntax-checked here, see rs_shim_common.h.
//
// ORB ONLY.  The GPU matchers are 256-bit Hamming kernels on 32-byte CV_8U rows, which is what the reference runs
// (OrbFeatureExtractor: NORM_HAMMING, max distance 64, src/features/OrbFeatureExtractor.h:14-22; the lightglue
// submodule behind DeepFeatureExtractor is empty).  Any other norm type or descriptor layout is refused loudly —
// log + empty result — instead of feeding float descriptors to a Hamming kernel.
#include "MapMatcher.h"

#include <unordered_map>

#include "Camera.h"
#include "Frame.h"
#include "Map.h"
#include "MapPoint.h"
#include "rs_shim_common.h"

namespace slam {
namespace {

bool orb_rows(const cv::Mat& d, cv::NormTypes norm, const char* who)
{
    if (norm == cv::NORM_HAMMING && (d.empty() || (d.type() == CV_8U && d.cols == RS_DESC_BYTES && d.isContinuous()))) return true;
    std::printf("%s: the GPU path matches 32-byte ORB descriptors under NORM_HAMMING only (norm %d, type %d, cols %d)\n",
                who, (int)norm, d.empty() ? -1 : d.type(), d.empty() ? 0 : d.cols);
    return false;
}

// Frame keeps its KD-tree private; the shim rebuilds the flattened tree from the keypoints with the library's host
// builder (same median-split rule, ties by keypoint index).  One build per call; a maintainer would cache it on Frame.
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

std::vector<MapPointMatch> reproj(const Camera& camera, float max_distance, cv::NormTypes norm, const Frame& frame,
                                  const std::vector<MapPoint*>& points, KeyFrame* required_observer, bool replace)
{
    using namespace rs_shim;
    const size_t N = frame.features().keypoints.size(), P = points.size();
    const cv::Mat& fd = frame.features().descriptors;
    if (N == 0 || !orb_rows(fd, norm, "MapMatcher::match")) return {};
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
                    const cv::Mat& d = kf->features().descriptors;                        // N x 32 CV_8U, row-contiguous
                    if (!orb_rows(d, norm, "MapMatcher::match (key frame)")) return {};
                    it = kf_id.emplace(kf, (int)kf_id.size()).first;
                    const Eigen::Vector3f c = kf->camera_center();
                    centers.insert(centers.end(), {c.x(), c.y(), c.z()});
                    pool_off.push_back((int32_t)(pool.size() / RS_DESC_BYTES));
                    pool.insert(pool.end(), d.ptr<uint8_t>(0), d.ptr<uint8_t>(0) + (size_t)d.rows * RS_DESC_BYTES);
                }
                obs_kf.push_back(it->second);
                obs_desc.push_back(pool_off[it->second] + (int32_t)idx);
            }
        obs_ptr[p + 1] = (int32_t)obs_kf.size();
    }
    Stage stage;
    DevBuf<float> d_kp(fa.kp), d_pos(pos), d_centers(centers);
    DevBuf<uint8_t> d_desc(fd.ptr<uint8_t>(0), N * RS_DESC_BYTES), d_matched(fa.matched), d_elig(eligible), d_pool(pool);
    DevBuf<int32_t> d_nk(fa.node_kp), d_l(fa.left), d_r(fa.right), d_optr(obs_ptr), d_okf(obs_kf), d_odesc(obs_desc);
    rs_frame_view fv{};
    pose_to_row_major(frame.pose(), fv.pose);
    float K[4];
    intrinsics(camera.get_intrinsic_matrix(), K);
    fv.fx = K[0]; fv.fy = K[1]; fv.cx = K[2]; fv.cy = K[3];
    fv.width = camera.get_width(); fv.height = camera.get_height(); fv.n_keypoints = (int)N;
    fv.d_keypoints = d_kp.p; fv.d_descriptors = d_desc.p; fv.d_kp_matched = d_matched.p;
    fv.d_kd_node_kp = d_nk.p; fv.d_kd_left = d_l.p; fv.d_kd_right = d_r.p; fv.kd_root = fa.root;
    rs_map_view mv{(int)P, d_pos.p, d_elig.p, d_optr.p, d_okf.p, d_odesc.p, d_centers.p, d_pool.p};
    DevBuf<int32_t> pk(P), pd(P), pp(N), pdist(N), mkp(N), mpt(N), cnt(1);
    // the reference keeps a candidate iff distance < max_descriptor_distance (:78, :88) with an integer distance and
    // a FLOAT threshold: d < 50.5 <=> d < 51, d < 64.0 <=> d < 64, i.e. ceil
    const int strict_max = (int)std::ceil(max_distance);
    if (!ok(rs_reproj_match(context(), &fv, &mv, replace, strict_max, pk.p, pd.p, pp.p, pdist.p, mkp.p, mpt.p, cnt.p), "rs_reproj_match"))
        return {};
    const auto hn = cnt.fetch(1);
    const auto hk = mkp.fetch(N), hp = mpt.fetch(N);
    stage.sync();
    std::vector<MapPointMatch> out;
    for (int i = 0; i < hn[0]; i++) out.push_back(MapPointMatch{*points[hp[i]], (size_t)hk[i]});
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
    return reproj(m_camera, m_max_descriptor_distance, m_norm_type, frame, points, nullptr, true);
}

std::vector<MapPointMatch> MapMatcher::match(const Frame& frame, Map& map, KeyFrame* required_observer) const
{
    return reproj(m_camera, m_max_descriptor_distance, m_norm_type, frame, all_points(map), required_observer, false);
}

std::vector<MapPointMatch> MapMatcher::match_descriptors(const Frame& frame, const KeyFrame& key_frame) const
{
    using namespace rs_shim;
    const cv::Mat& q = frame.features().descriptors;
    const cv::Mat& kd = key_frame.features().descriptors;
    if (!orb_rows(q, m_norm_type, "MapMatcher::match_descriptors") || !orb_rows(kd, m_norm_type, "MapMatcher::match_descriptors")) return {};
    std::vector<MapPoint*> points;
    std::vector<uint8_t> train;
    for (const auto& m : key_frame.map_matches()) {                     // ascending keypoint order, :143-144
        points.push_back(&m.point);
        // Frame::descriptor(i) is descriptors.row(i) (src/Frame.cpp:128-131): a row HEADER whose datastart / dataend
        // still span the whole parent matrix — the row's bytes are ptr<uint8_t>(i) .. + 32
        const uint8_t* row = kd.ptr<uint8_t>((int)m.keypoint_index);
        train.insert(train.end(), row, row + RS_DESC_BYTES);
    }
    if (points.empty() || q.empty()) return {};                         // :139-141
    const int nq = q.rows, nt = (int)points.size();
    Stage stage;
    DevBuf<uint8_t> dq(q.ptr<uint8_t>(0), (size_t)nq * RS_DESC_BYTES), dt(train);
    DevBuf<int32_t> mq(nq), mt(nq), cnt(1);
    // :152 rejects d0 > max (float) with an integer d0: keep d0 <= floor(max)
    const int keep_max = (int)std::floor(m_max_descriptor_distance);
    if (!ok(rs_match_descriptors(context(), dq.p, nq, dt.p, nt, 1, keep_max, mq.p, mt.p, cnt.p, nullptr, nullptr, nullptr, nullptr),
            "rs_match_descriptors"))
        return {};
    const auto hn = cnt.fetch(1);
    const auto hq = mq.fetch(nq), ht = mt.fetch(nq);
    stage.sync();
    std::vector<MapPointMatch> out;
    for (int i = 0; i < hn[0]; i++) out.push_back({*points[(size_t)ht[i]], (size_t)hq[i]});
    return out;
}

}  // namespace slam
