// slam_host.h — host-side C++ mirror of the reference's hot-path interfaces over the C-ABI.
//
// Same names, argument meaning and error behaviour as the reference headers
//   src/MapMatcher.h:16-38, src/Triangulation.h:9-37, src/Optimization.h:21-97, src/LocalWindow.h:13-19
// but on plain array types: OpenCV / Eigen / Ceres are not available in this image, so cv::Mat
// descriptors become a byte vector, Eigen::Matrix4f a row-major std::array<float,16>, etc.
// The data model (Frame / KeyFrame / MapPoint / Map, reference src/Frame.h, src/MapPoint.h,
// src/Map.h) is reduced to what the hot path reads and writes.  Every method marshals the
// pointer graph to the SoA layout of include/rsgpu.h, uploads, calls ONE C-ABI entry point and
// downloads — exactly what the drop-in shims of INTEGRATION.md do with the reference's own types.
// There is no CPU fallback: without a GPU the Session constructor throws.
#pragma once
#include <array>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <unordered_set>
#include <utility>
#include <vector>

#include "../../include/rsgpu.h"

namespace slam {

struct Vec2f { float x = 0, y = 0; };
struct Vec3f { float x = 0, y = 0, z = 0; };
using Mat4f = std::array<float, 16>;   // row-major world->camera, Frame::pose()

inline Mat4f identity4() { return {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}; }

// src/Camera.h:8-25
class Camera {
  public:
    Camera(float fx, float fy, float cx, float cy, int width, int height)
        : m_fx(fx), m_fy(fy), m_cx(cx), m_cy(cy), m_width(width), m_height(height) {}
    float fx() const { return m_fx; }
    float fy() const { return m_fy; }
    float cx() const { return m_cx; }
    float cy() const { return m_cy; }
    int get_width() const { return m_width; }
    int get_height() const { return m_height; }

  private:
    float m_fx, m_fy, m_cx, m_cy;
    int m_width, m_height;
};

// src/features/FeatureExtractor.h:15-38
struct KeyPoint { Vec2f pt; };
struct ExtractedFeatures {
    std::vector<KeyPoint> keypoints;
    std::vector<uint8_t> descriptors;   // N x 32, row-major (cv::Mat N x 32 CV_8U)
};
struct FeatureMatch {
    FeatureMatch(int train, int query) : train_index((size_t)train), query_index((size_t)query) {}
    size_t train_index, query_index;
};

class MapPoint;
class KeyFrame;
struct MapPointMatch { MapPoint& point; size_t keypoint_index; };

// src/Frame.h:18-74 (hot-path subset).  The KD-tree is built at construction like the reference's.
class Frame {
  public:
    Frame(int index, ExtractedFeatures features);
    virtual ~Frame() = default;
    size_t index() const { return m_index; }
    const ExtractedFeatures& features() const { return m_features; }
    const KeyPoint& keypoint(size_t i) const { return m_features.keypoints[i]; }
    const Mat4f& pose() const { return m_pose; }
    void set_pose(const Mat4f& p) { m_pose = p; }
    Vec3f camera_center() const;                               // -R^T t, src/Frame.cpp:39-42
    void add_map_match(const MapPointMatch& m);                // src/Frame.cpp:80-102
    bool is_matched(size_t keypoint_index) const { return m_map_matches[keypoint_index] != nullptr; }
    bool is_matched(const MapPoint& p) const;
    size_t num_map_matches() const { return m_num; }
    std::vector<MapPointMatch> map_matches() const;            // ascending keypoint index, src/Frame.cpp:154-174
    const std::vector<MapPoint*>& match_table() const { return m_map_matches; }   // keypoint -> point or null (what the reference's
                                                               // MapPointIterator walks, src/Frame.h:20-33: no list is built)
    const std::vector<int32_t>& kd_node_kp() const { return m_kd_node_kp; }
    const std::vector<int32_t>& kd_left() const { return m_kd_left; }
    const std::vector<int32_t>& kd_right() const { return m_kd_right; }
    int kd_root() const { return m_kd_root; }

  private:
    size_t m_index;
    ExtractedFeatures m_features;
    Mat4f m_pose = identity4();
    std::vector<MapPoint*> m_map_matches;
    std::unordered_set<const MapPoint*> m_matched_points;      // the points in m_map_matches (src/Frame.h:72): is_matched(point) is a lookup
    size_t m_num = 0;
    std::vector<int32_t> m_kd_node_kp, m_kd_left, m_kd_right;
    int m_kd_root = -1;
};

class KeyFrame : public Frame {
  public:
    explicit KeyFrame(Frame&& f) : Frame(std::move(f)) {}
};

// src/MapPoint.h:12-39.  Observations keep INSERTION order (the reference's unordered_map order is
// unspecified).
class MapPoint {
  public:
    explicit MapPoint(const Vec3f& p) : m_position(p) {}
    const Vec3f& position() const { return m_position; }
    void set_position(const Vec3f& p) { m_position = p; }
    const std::vector<std::pair<KeyFrame*, size_t>>& observations() const { return m_obs; }
    bool is_observed_by(const KeyFrame* kf) const;

  private:
    friend class Map;
    Vec3f m_position;
    std::vector<std::pair<KeyFrame*, size_t>> m_obs;
};

// src/Map.h:12-66 (hot-path subset): owns the points, iteration order = insertion order
class Map {
  public:
    MapPoint& create_point(const Vec3f& position) { m_points.push_back(std::make_unique<MapPoint>(position)); return *m_points.back(); }
    void associate(KeyFrame& kf, MapPoint& point, size_t keypoint_index);   // src/Map.cpp:association
    size_t size() const { return m_points.size(); }
    MapPoint& operator[](size_t i) { return *m_points[i]; }
    const MapPoint& operator[](size_t i) const { return *m_points[i]; }

  private:
    std::vector<std::unique_ptr<MapPoint>> m_points;
};

enum NormTypes { NORM_HAMMING = 6 };   // cv::NORM_HAMMING

// src/MapMatcher.h:16-38
class MapMatcher {
  public:
    MapMatcher(const Camera& camera, float max_descriptor_distance, NormTypes norm_type);
    std::vector<MapPointMatch> match_map(const Frame& frame, Map& map) const;
    std::vector<MapPointMatch> match_key_frame(const Frame& frame, Map& map, KeyFrame* key_frame) const;
    std::vector<MapPointMatch> match_for_fuse(const Frame& frame, const std::vector<MapPoint*>& points) const;
    std::vector<MapPointMatch> match_descriptors(const Frame& frame, const KeyFrame& key_frame) const;

  private:
    std::vector<MapPointMatch> match(const Frame& frame, const std::vector<MapPoint*>& points,
                                     const KeyFrame* required_observer, bool replace) const;
    const Camera& m_camera;
    float m_max_descriptor_distance;
    NormTypes m_norm_type;
};

namespace triangulation {
// src/Triangulation.h:9-37
static const float MIN_PARALLAX_COSINE = 0.9999f;
struct TriangulatedPoint { Vec3f position; int match_index; };
std::pair<std::vector<Vec2f>, std::vector<Vec2f>> get_matching_points(const ExtractedFeatures& f1,
                                                                      const ExtractedFeatures& f2,
                                                                      const std::vector<FeatureMatch>& matches);
std::vector<TriangulatedPoint> triangulate_points(const std::vector<Vec2f>& points1, const std::vector<Vec2f>& points2,
                                                  const Mat4f& pose1, const Mat4f& pose2, const Camera& camera,
                                                  float min_parallax_cosine = MIN_PARALLAX_COSINE,
                                                  float max_reprojection_error = 2.0f);
std::vector<TriangulatedPoint> triangulate_points(const Frame& frame1, const Frame& frame2,
                                                  const std::vector<FeatureMatch>& matches, const Camera& camera);
}  // namespace triangulation

namespace tracks {
// The arithmetic body of Mapper::triangulate_tracks (src/Mapper.cpp:246-305) on the types of
// src/TrackStore.h:17-29.  The Mapper keeps the pointer work (window membership, create_point /
// associate, :306-330): it passes the tracks in std::map<TrackId, Track> order together with the
// trajectory poses its sightings refer to and gets back what to create and what to erase.
struct TrackSighting { size_t frame_index = 0; Vec2f pixel; };
struct Track { std::vector<TrackSighting> sightings; size_t keypoint_index = 0; };
struct Candidate {
    size_t track;            // index into the input vector
    Vec3f position;
    size_t keypoint_index;
    float parallax_cosine, required_cosine;
};
struct Selection {
    std::vector<Candidate> accepted;      // creation order: above-threshold in track order, then the quota top-up
    size_t topped_up = 0;
    std::vector<size_t> inconsistent;     // tracks to erase (:332-334)
};
static const float TRACK_MIN_PARALLAX_COSINE = 0.999848f, ROTATION_PARALLAX_FACTOR = 0.20f;
static const float ANY_PARALLAX_COSINE = 1.0f, TRACK_MAX_REPROJECTION_ERROR = 4.0f;
static const size_t MIN_NEW_POINTS_PER_KEY_FRAME = 100;
// `trajectory_poses[i]` = Trajectory::pose_at(i); the key frame contributes its pose and keypoints.
Selection select_track_points(const KeyFrame& key_frame, const std::vector<Track>& tracks,
                              const std::vector<Mat4f>& trajectory_poses, const Camera& camera,
                              size_t min_new_points = MIN_NEW_POINTS_PER_KEY_FRAME);

// The arithmetic of Mapper::cull_points (src/Mapper.cpp:410-419): mean reprojection error of every given point
// over its observations, and which of them exceed the limit.  The Mapper selects the local points (:398-408)
// and removes the returned ones from the map (:426-429).
static const float MAX_POINT_REPROJECTION_ERROR = 3.0f;
struct CullResult {
    std::vector<float> mean_error;            // per input point
    std::vector<size_t> to_remove;            // indices into the input vector, ascending
    double error_sum = 0.0;                   // Slam::reprojection_error() = error_sum / observations (src/Slam.cpp:302-317)
    size_t observations = 0;
};
CullResult point_errors(const std::vector<MapPoint*>& points, const Camera& camera,
                        float max_mean_error = MAX_POINT_REPROJECTION_ERROR);
}  // namespace tracks

namespace optimization {
// src/Optimization.h:23-26, 76-81; src/LocalWindow.h:13-19 (vision-only: InertialInput{} default)
struct FrameConfig { bool optimize; Frame* frame; };
bool refine_pose(Frame& frame, const Camera& camera);
bool bundle_adjust(const std::vector<FrameConfig>& frames, const Camera& camera, Map& map);
std::vector<FrameConfig> build_local_window(const std::vector<std::shared_ptr<KeyFrame>>& key_frames,
                                            Frame& new_frame, size_t window_size, bool fix_oldest = false);
// last solver summary (the reference prints ceres' BriefReport, src/Optimization.cpp:135)
const rs_ba_summary& last_summary();
}  // namespace optimization

// Process-wide device session (one context per process, SURVEY.md §8b "Threading").
class Session {
  public:
    static Session& get();
    rs_context* ctx() const { return m_ctx; }
    ~Session();

  private:
    Session();
    rs_context* m_ctx = nullptr;
};

}  // namespace slam
