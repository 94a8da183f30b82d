// Drop-in replacement of the reference's src/Optimization.cpp (keeps src/Optimization.h).
// Compiled only in the reference's tree; syntax-checked here, see rs_shim_common.h.
//  * Inertial residual blocks (a15): InertialInput::usable() -> rs_bundle_adjust_inertial (one IMU factor pair per
//    consecutive pair of optimised frames with >= 2 samples, src/Optimization.cpp:317-346), RotationPrior /
//    InertialDelta -> rs_refine_pose_inertial (:231-267).  imu::preintegrate and imu::Stream::between
//    (src/Imu.cpp, src/ImuStream.cpp: small sequential host code) stay the reference's and are called from here.
//  * pose_graph (a14, src/Optimization.cpp:376-639): the solve is rs_pose_graph (a host function of the library —
//    sparse envelope Cholesky, same Ceres trust-region schedule), the point transform (transform_points, :512-536) runs
//    on the device through rs_reanchor_points with the owner chosen here exactly as the reference chooses it
//    (smallest KeyFrame::index among the observers).
#include "Optimization.h"

#include <algorithm>
#include <unordered_map>

#include "Camera.h"
#include "Frame.h"
#include "Map.h"
#include "MapPoint.h"
#include "rs_shim_common.h"

namespace slam::optimization {

namespace {
constexpr size_t MIN_OBSERVATIONS_TO_OPTIMIZE = 2;

void intrinsics(const Camera& camera, float K[4]) { rs_shim::intrinsics(camera.get_intrinsic_matrix(), K); }

// imu::Preintegrated (Eigen, column-major) -> rs_imu_factor (row-major)
rs_imu_factor to_factor(const imu::Preintegrated& d, const imu::NoiseDensity& noise, int cam_i, int cam_j)
{
    rs_imu_factor f{};
    f.cam_i = cam_i; f.cam_j = cam_j; f.duration = d.duration;
    for (int r = 0; r < 3; r++) {
        f.velocity[r] = d.velocity[r]; f.position[r] = d.position[r];
        f.bias_gyro[r] = d.bias.gyro[r]; f.bias_accel[r] = d.bias.accel[r];
        for (int c = 0; c < 3; c++) f.rotation[3 * r + c] = d.rotation(r, c);
    }
    for (int r = 0; r < 9; r++) {
        for (int c = 0; c < 9; c++) f.covariance[9 * r + c] = d.covariance(r, c);
        for (int c = 0; c < 6; c++) f.bias_jacobian[6 * r + c] = d.bias_jacobian(r, c);
    }
    f.gyro_bias_sigma = noise.gyro_bias; f.accel_bias_sigma = noise.accel_bias;
    return f;
}

void pack_inertial(const Frame& frame, double velocity[3], double bias[6])
{
    const auto& st = frame.inertial();
    for (int k = 0; k < 3; k++) { velocity[k] = st.velocity[k]; bias[k] = st.bias.gyro[k]; bias[3 + k] = st.bias.accel[k]; }
}

void unpack_inertial(const double velocity[3], const double bias[6], Frame& frame)     // src/Optimization.cpp:182-190
{
    InertialState st;
    st.velocity = Eigen::Vector3d(velocity[0], velocity[1], velocity[2]);
    st.bias.gyro = Eigen::Vector3d(bias[0], bias[1], bias[2]);
    st.bias.accel = Eigen::Vector3d(bias[3], bias[4], bias[5]);
    frame.set_inertial(st);
}

bool report(const char* what, const rs_ba_summary& s)
{
    std::printf("%s: iterations %d, cost %.6e -> %.6e, termination %d\n", what, s.iterations, s.initial_cost, s.final_cost, s.termination);
    if (!s.usable) std::printf("Optimization rejected, unusable or non-improving solution\n");
    return s.usable != 0;
}
}  // namespace

bool refine_pose(Frame& frame, const Camera& camera, const InertialConstraint& inertial)
{
    using namespace rs_shim;
    std::vector<double> pts;
    std::vector<float> uv;
    for (const auto& m : frame.map_matches()) {
        if (m.point.observations().size() < MIN_OBSERVATIONS_TO_OPTIMIZE) continue;
        for (int k = 0; k < 3; k++) pts.push_back((double)m.point.position()[k]);
        uv.push_back(frame.keypoint(m.keypoint_index).pt.x);
        uv.push_back(frame.keypoint(m.keypoint_index).pt.y);
    }
    if (uv.empty()) return false;
    float T[16], K[4];
    double cam[6];
    pose_to_row_major(frame.pose(), T);
    rs_pack_pose(T, cam);
    intrinsics(camera, K);
    Stage stage;
    DevBuf<double> dp(pts);
    DevBuf<float> duv(uv);
    rs_ba_summary s{};
    // the InertialConstraint (:231-258): an enabled InertialDelta wins over a RotationPrior, a disabled one is nothing
    const auto* prior = std::get_if<RotationPrior>(&inertial);
    const auto* delta = std::get_if<InertialDelta>(&inertial);
    int kind = 0;
    double predicted[9] = {}, prev_pose[6] = {}, prev_velocity[3] = {}, prev_bias[6] = {}, velocity[3] = {}, bias_unused[6] = {}, gravity[3] = {};
    rs_imu_factor factor{};
    if (delta != nullptr && delta->enabled()) {
        kind = 2;
        float Tp[16];
        pose_to_row_major(delta->previous->pose(), Tp);
        rs_pack_pose(Tp, prev_pose);
        pack_inertial(*delta->previous, prev_velocity, prev_bias);
        pack_inertial(frame, velocity, bias_unused);
        factor = to_factor(delta->summary, delta->noise, 0, 0);
        for (int k = 0; k < 3; k++) gravity[k] = delta->gravity[k];
    } else if (prior != nullptr && prior->enabled()) {
        kind = 1;
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) predicted[3 * r + c] = prior->predicted(r, c);
    }
    if (!ok(rs_refine_pose_inertial(context(), cam, dp.p, duv.p, (int)(uv.size() / 2), K, kind, predicted,
                                    prior != nullptr ? prior->sigma_radians : 0.0, prev_pose, prev_velocity, prev_bias, &factor,
                                    gravity, velocity, nullptr, &s), "rs_refine_pose_inertial"))
        return false;
    if (!report("refine_pose", s)) return false;
    rs_unpack_pose(cam, T);
    frame.set_pose(pose_from_row_major(T));
    if (kind == 2) unpack_inertial(velocity, prev_bias, frame);           // :263-265: this frame takes the previous bias
    return true;
}

bool bundle_adjust(const std::vector<FrameConfig>& frames, const Camera& camera, Map&, const InertialInput& inertial)
{
    using namespace rs_shim;
    const size_t C = frames.size();
    std::vector<double> cams(6 * C);
    std::vector<uint8_t> cam_free(C);
    for (size_t c = 0; c < C; c++) {
        float T[16];
        pose_to_row_major(frames[c].frame->pose(), T);
        rs_pack_pose(T, &cams[6 * c]);
        cam_free[c] = frames[c].optimize;
    }
    std::vector<MapPoint*> free_pts;
    std::unordered_map<const MapPoint*, int> pid;
    for (const auto& fc : frames) {
        if (!fc.optimize) continue;
        for (auto m : fc.frame->map_matches()) {
            if (m.point.observations().size() < MIN_OBSERVATIONS_TO_OPTIMIZE) continue;
            if (pid.emplace(&m.point, (int)free_pts.size()).second) free_pts.push_back(&m.point);
        }
    }
    const size_t P = free_pts.size();
    std::vector<std::vector<std::pair<int, cv::Point2f>>> per_point(P);
    for (size_t c = 0; c < C; c++)
        for (const auto& m : frames[c].frame->map_matches()) {
            auto it = pid.find(&m.point);
            if (it != pid.end()) per_point[it->second].emplace_back((int)c, frames[c].frame->keypoint(m.keypoint_index).pt);
        }
    std::vector<int32_t> obs_ptr(P + 1, 0), obs_cam;
    std::vector<float> obs_uv;
    std::vector<double> pts(3 * P);
    for (size_t p = 0; p < P; p++) {
        for (int k = 0; k < 3; k++) pts[3 * p + k] = (double)free_pts[p]->position()[k];
        for (const auto& [c, px] : per_point[p]) { obs_cam.push_back(c); obs_uv.push_back(px.x); obs_uv.push_back(px.y); }
        obs_ptr[p + 1] = (int32_t)obs_cam.size();
    }
    if (P == 0 || obs_cam.empty()) return false;
    float K[4];
    intrinsics(camera, K);
    Stage stage;
    DevBuf<double> dc(cams), dp(pts);
    DevBuf<int32_t> dptr(obs_ptr), dcam(obs_cam);
    DevBuf<float> duv(obs_uv);
    rs_ba_summary s{};
    // IMU factor pairs between consecutive optimised frames (:317-346); none when the input is not usable
    std::vector<rs_imu_factor> factors;
    std::vector<double> velocity(3 * C), bias(6 * C);
    for (size_t c = 0; c < C; c++) pack_inertial(*frames[c].frame, &velocity[3 * c], &bias[6 * c]);
    if (inertial.usable())
        for (size_t i = 0; i + 1 < C; i++) {
            if (!frames[i].optimize || !frames[i + 1].optimize) continue;
            const auto samples = inertial.stream->between(inertial.time_of(frames[i].frame->index()), inertial.time_of(frames[i + 1].frame->index()));
            if (samples.size() < 2) continue;
            factors.push_back(to_factor(imu::preintegrate(samples, inertial.noise, frames[i].frame->inertial().bias), inertial.noise, (int)i, (int)(i + 1)));
        }
    const double gravity[3] = {inertial.gravity[0], inertial.gravity[1], inertial.gravity[2]};
    if (!ok(rs_bundle_adjust_inertial(context(), (int)C, (int)P, (int)obs_cam.size(), dc.p, cam_free.data(), dp.p, dptr.p, dcam.p, duv.p, K,
                                      velocity.data(), bias.data(), factors.data(), (int)factors.size(), gravity, nullptr, &s),
            "rs_bundle_adjust_inertial"))
        return false;
    if (!report("bundle_adjust", s)) return false;
    std::vector<double> hc(6 * C);
    if (!ok(rs_ba_get_cameras(context(), hc.data(), (int)C), "rs_ba_get_cameras")) return false;   // pinned mirror, no read-back
    const auto hp = dp.fetch(3 * P);
    stage.sync();
    for (size_t c = 0; c < C; c++)
        if (frames[c].optimize) {
            float T[16];
            rs_unpack_pose(&hc[6 * c], T);
            frames[c].frame->set_pose(pose_from_row_major(T));
            unpack_inertial(&velocity[3 * c], &bias[6 * c], *frames[c].frame);      // :366 (unchanged values without factors)
        }
    for (size_t p = 0; p < P; p++) free_pts[p]->set_position(Eigen::Vector3f((float)hp[3 * p], (float)hp[3 * p + 1], (float)hp[3 * p + 2]));
    return true;
}

// a14: optimization::pose_graph, src/Optimization.cpp:540-639.
bool pose_graph(const std::vector<std::shared_ptr<KeyFrame>>& key_frames, const std::vector<PoseGraphConstraint>& loops, Map& map,
                bool four_dof, const Eigen::Vector3d& gravity)
{
    if (key_frames.size() < 3 || loops.empty()) return false;                                           // :546-548
    const size_t n = key_frames.size();
    std::vector<float> before(16 * n), after(16 * n), r_delta(9 * n);
    std::unordered_map<const Frame*, int32_t> slot;
    slot.reserve(n);
    for (size_t i = 0; i < n; i++) {
        rs_shim::pose_to_row_major(key_frames[i]->pose(), &before[16 * i]);
        slot[key_frames[i].get()] = (int32_t)i;
    }
    std::vector<rs_pose_graph_edge> edges(loops.size());
    for (size_t l = 0; l < loops.size(); l++) {
        edges[l].from = loops[l].from < n ? (int32_t)loops[l].from : -1;                                // out of range: skipped (:590-592)
        edges[l].to = loops[l].to < n ? (int32_t)loops[l].to : -1;
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) edges[l].relative[4 * r + c] = loops[l].relative(r, c);
    }
    const double g[3] = {gravity[0], gravity[1], gravity[2]};
    rs_ba_summary summary{};
    if (!rs_shim::ok(rs_pose_graph((int)n, before.data(), edges.data(), (int)edges.size(), four_dof ? 1 : 0, g, nullptr, after.data(),
                                   r_delta.data(), &summary, nullptr, 0, nullptr), "rs_pose_graph"))
        return false;
    const bool ran_four_dof = four_dof && (g[0] * g[0] + g[1] * g[1] + g[2] * g[2]) >= 1e-6;              // :552-553
    std::printf("Pose graph %s iterations %d cost %g -> %g\n", ran_four_dof ? "4-DOF" : "SE3", summary.iterations,
                summary.initial_cost, summary.final_cost);
    if (!summary.usable) {                                                                              // :610-616
        std::printf("Pose graph rejected, unusable or non-improving solution\n");
        return false;
    }

    // transform_points, :512-536: owner = observer of smallest index; listed only when it is one of the key frames
    std::vector<MapPoint*> moved;
    std::vector<int32_t> owner_slot;
    std::vector<float> positions;
    for (auto& point : map) {
        if (point.observations().empty()) continue;
        KeyFrame* owner = nullptr;
        for (const auto& [observer, _] : point.observations())
            if (owner == nullptr || observer->index() < owner->index()) owner = observer;
        const auto it = slot.find(owner);
        if (it == slot.end()) continue;
        const Eigen::Vector3f p = point.position();
        moved.push_back(&point);
        owner_slot.push_back(it->second);
        positions.insert(positions.end(), {p[0], p[1], p[2]});
    }
    {
        rs_shim::Stage stage;
        rs_shim::DevBuf<float> d_pos(positions);
        rs_shim::DevBuf<int32_t> d_owner(owner_slot);
        // the poses stay where the reference keeps them: host arrays (kernel arguments for up to 32 frames, no upload)
        if (!rs_shim::ok(rs_reanchor_points_host_poses(rs_shim::context(), (int)moved.size(), nullptr, d_owner.p, before.data(), after.data(), (int)n, d_pos.p),
                         "rs_reanchor_points_host_poses"))
            return false;
        std::vector<float> out = d_pos.fetch(positions.size());
        stage.sync();
        for (size_t i = 0; i < moved.size(); i++) moved[i]->set_position(Eigen::Vector3f(out[3 * i], out[3 * i + 1], out[3 * i + 2]));
    }

    float max_correction = 0.0F;
    for (size_t i = 0; i < n; i++) {                                                                    // apply_corrected_pose, :499-510
        Frame& frame = *key_frames[i];
        const Eigen::Vector3f c0 = frame.camera_center();
        frame.set_pose(rs_shim::pose_from_row_major(&after[16 * i]));
        InertialState inertial = frame.inertial();
        const double v[3] = {inertial.velocity[0], inertial.velocity[1], inertial.velocity[2]};
        for (int r = 0; r < 3; r++)
            inertial.velocity[r] = (double)r_delta[9 * i + 3 * r] * v[0] + (double)r_delta[9 * i + 3 * r + 1] * v[1] + (double)r_delta[9 * i + 3 * r + 2] * v[2];
        frame.set_inertial(inertial);
        const Eigen::Vector3f c1 = frame.camera_center();
        const float d[3] = {c1[0] - c0[0], c1[1] - c0[1], c1[2] - c0[2]};
        max_correction = std::max(max_correction, std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]));
    }
    std::printf("Pose graph applied, max snap %g loops %zu\n", max_correction, loops.size());
    return true;
}

}  // namespace slam::optimization
