// Drop-in replacement of the reference's src/Triangulation.cpp (keeps src/Triangulation.h).
// Compiled only in the reference's tree; syntax-checked here, see rs_shim_common.h.
#include "Triangulation.h"

#include "Frame.h"
#include "rs_shim_common.h"

namespace slam::triangulation {

std::pair<std::vector<Eigen::Vector2f>, std::vector<Eigen::Vector2f>>
get_matching_points(const ExtractedFeatures& features1, const ExtractedFeatures& features2, const std::vector<FeatureMatch>& matches)
{
    std::vector<Eigen::Vector2f> p1, p2;
    for (const auto& m : matches) {
        p1.emplace_back(features1.keypoints[m.train_index].pt.x, features1.keypoints[m.train_index].pt.y);
        p2.emplace_back(features2.keypoints[m.query_index].pt.x, features2.keypoints[m.query_index].pt.y);
    }
    return {p1, p2};
}

std::vector<TriangulatedPoint> triangulate_points(const Frame& frame1, const Frame& frame2,
                                                  const std::vector<FeatureMatch>& matches, const Camera& camera)
{
    auto [p1, p2] = get_matching_points(frame1.features(), frame2.features(), matches);
    return triangulate_points(p1, p2, frame1.pose(), frame2.pose(), camera);
}

std::vector<TriangulatedPoint> triangulate_points(const std::vector<Eigen::Vector2f>& points1,
                                                  const std::vector<Eigen::Vector2f>& points2,
                                                  const Eigen::Matrix4f& pose1, const Eigen::Matrix4f& pose2,
                                                  const Camera& camera, float min_parallax_cosine,
                                                  float max_reprojection_error)
{
    using namespace rs_shim;
    if (points1.empty() || points2.empty()) return {};
    const size_t n = points1.size();
    std::vector<float> uv1(2 * n), uv2(2 * n), poses(32);
    for (size_t i = 0; i < n; i++) { uv1[2 * i] = points1[i].x(); uv1[2 * i + 1] = points1[i].y(); uv2[2 * i] = points2[i].x(); uv2[2 * i + 1] = points2[i].y(); }
    pose_to_row_major(pose1, poses.data());
    pose_to_row_major(pose2, poses.data() + 16);
    float K[4];
    intrinsics(camera.get_intrinsic_matrix(), K);
    // host in, host out (rs_triangulate_host): up to 256 correspondences — Mapper::triangulate_tracks calls this with one —
    // are a single launch whose result lands in pinned memory behind a completion flag
    std::vector<int32_t> hi(n);
    std::vector<float> hx(3 * n);
    int m = 0;
    if (!ok(rs_triangulate_host(context(), uv1.data(), uv2.data(), (int)n, poses.data(), poses.data() + 16, K, min_parallax_cosine,
                                max_reprojection_error, hi.data(), hx.data(), &m), "rs_triangulate_host"))
        return {};
    std::vector<TriangulatedPoint> out;
    for (int i = 0; i < m; i++) out.push_back(TriangulatedPoint{Eigen::Vector3f(hx[3 * i], hx[3 * i + 1], hx[3 * i + 2]), hi[i]});
    return out;
}

}  // namespace slam::triangulation
// Mapper::triangulate_tracks (src/Mapper.cpp:246-259) calls the function above once per track with N = 1 (and
// pose::recover_pose four times per frame).  That keeps working unchanged but costs one staging group + two launches +
// one synchronisation per call (bench.py --config pass -> boundary.triangulate_points_n1); the one-launch-per-key-frame
// form is Mapper_triangulate_tracks.inc in this directory.
