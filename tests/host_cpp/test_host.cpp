// test_host.cpp — exercises the host mirror (racing-slam_amd/host/slam_host.h) on a small synthetic
// scene and checks every interface function against the CPU oracle on identical inputs
// (test infrastructure: links oracle/liboracle.so).  Written the way a test of the reference's own
// classes would read: build frames / a map, call MapMatcher / triangulate_points / bundle_adjust.
// Exit code 0 = all checks passed.  Needs a GPU (the host mirror has no CPU fallback).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <random>
#include <vector>

#include "../../oracle/rs_oracle.h"
#include "../../racing-slam_amd/host/slam_host.h"

using namespace slam;

static int g_fail = 0;
#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) { std::printf("CHECK FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); g_fail++; } \
    } while (0)

static Mat4f make_pose(double yaw, double cx, double cy, double cz)
{
    // camera-to-world rotation = yaw about y; world->camera pose
    const double c = std::cos(yaw), s = std::sin(yaw);
    const double R[3][3] = {{c, 0, s}, {0, 1, 0}, {-s, 0, c}};   // R_wc
    Mat4f T = identity4();
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) T[4 * i + j] = (float)R[j][i];   // R_cw = R_wc^T
    const double ctr[3] = {cx, cy, cz};
    for (int i = 0; i < 3; i++) {
        double t = 0;
        for (int j = 0; j < 3; j++) t -= R[j][i] * ctr[j];
        T[4 * i + 3] = (float)t;
    }
    return T;
}

static Vec2f project(const Mat4f& T, const Camera& cam, const double X[3], double* depth)
{
    double p[3];
    for (int i = 0; i < 3; i++) p[i] = T[4 * i] * X[0] + T[4 * i + 1] * X[1] + T[4 * i + 2] * X[2] + T[4 * i + 3];
    *depth = p[2];
    return Vec2f{(float)(cam.fx() * p[0] / p[2] + cam.cx()), (float)(cam.fy() * p[1] / p[2] + cam.cy())};
}

int main()
{
    std::mt19937_64 rng(20261004);
    std::normal_distribution<double> gauss(0.0, 1.0);
    std::uniform_real_distribution<double> uni(0.0, 1.0);
    const Camera camera(500.f, 500.f, 320.f, 240.f, 640, 480);
    const int NKF = 6, NPT = 400;

    // ---- scene: 6 keyframes on a forward track, 400 landmarks
    std::vector<Mat4f> poses;
    for (int k = 0; k < NKF + 1; k++) poses.push_back(make_pose(0.02 * k, 0.05 * k, 0.0, 0.4 * k));
    std::vector<std::array<double, 3>> X(NPT);
    std::vector<std::array<uint8_t, 32>> base(NPT);
    for (int p = 0; p < NPT; p++) {
        const double z = 4 + 10 * uni(rng), u = 640 * uni(rng), v = 480 * uni(rng);
        X[p] = {(u - 320) / 500 * z, (v - 240) / 500 * z, z + 1.0};
        for (auto& b : base[p]) b = (uint8_t)(rng() & 0xFF);
    }
    auto noisy_desc = [&](int p, uint8_t* out) {
        for (int i = 0; i < 32; i++) {
            uint8_t b = base[p][i];
            for (int bit = 0; bit < 8; bit++)
                if (uni(rng) < 0.04) b ^= (uint8_t)(1u << bit);
            out[i] = b;
        }
    };
    // frames: every keyframe sees the landmarks that project inside the image
    std::vector<std::shared_ptr<KeyFrame>> kfs;
    std::vector<std::vector<int>> kp_landmark(NKF + 1);
    auto make_frame = [&](int k) {
        ExtractedFeatures f;
        for (int p = 0; p < NPT; p++) {
            double depth;
            Vec2f uv = project(poses[k], camera, X[p].data(), &depth);
            if (depth < 0.5 || uv.x < 0 || uv.x >= 640 || uv.y < 0 || uv.y >= 480 || uni(rng) < 0.15) continue;
            uv.x += (float)(0.5 * gauss(rng)); uv.y += (float)(0.5 * gauss(rng));
            f.keypoints.push_back(KeyPoint{uv});
            f.descriptors.resize(f.descriptors.size() + 32);
            noisy_desc(p, f.descriptors.data() + f.descriptors.size() - 32);
            kp_landmark[k].push_back(p);
        }
        return Frame(k, std::move(f));
    };
    for (int k = 0; k < NKF; k++) {
        kfs.push_back(std::make_shared<KeyFrame>(make_frame(k)));
        kfs.back()->set_pose(poses[k]);
    }
    Frame new_frame = make_frame(NKF);
    new_frame.set_pose(poses[NKF]);

    // map: a landmark becomes a map point when >= 2 keyframes see it; positions perturbed
    Map map;
    std::vector<int> point_of(NPT, -1);
    for (int p = 0; p < NPT; p++) {
        int seen = 0;
        for (int k = 0; k < NKF; k++)
            for (int lm : kp_landmark[k]) seen += lm == p;
        if (seen < 2) continue;
        point_of[p] = (int)map.size();
        map.create_point(Vec3f{(float)(X[p][0] + 0.02 * gauss(rng)), (float)(X[p][1] + 0.02 * gauss(rng)), (float)(X[p][2] + 0.05 * gauss(rng))});
    }
    for (int k = 0; k < NKF; k++)
        for (size_t i = 0; i < kp_landmark[k].size(); i++)
            if (point_of[kp_landmark[k][i]] >= 0) map.associate(*kfs[k], map[(size_t)point_of[kp_landmark[k][i]]], i);
    std::printf("scene: %zu map points, new frame %zu keypoints\n", map.size(), new_frame.features().keypoints.size());

    // ---- a4: match_descriptors(frame, key_frame) vs oracle
    {
        MapMatcher matcher(camera, 64.f, NORM_HAMMING);
        const KeyFrame& kf = *kfs[NKF - 1];
        auto got = matcher.match_descriptors(new_frame, kf);
        auto km = kf.map_matches();
        std::vector<uint8_t> train;
        for (auto& m : km) train.insert(train.end(), kf.features().descriptors.begin() + 32 * m.keypoint_index, kf.features().descriptors.begin() + 32 * (m.keypoint_index + 1));
        const int nq = (int)new_frame.features().keypoints.size(), nt = (int)km.size();
        std::vector<int32_t> mq(nq), mt(nq);
        int32_t cnt = 0;
        orc_match_descriptors(new_frame.features().descriptors.data(), nq, train.data(), nt, 64, mq.data(), mt.data(), &cnt);
        CHECK((int)got.size() == cnt);
        CHECK(cnt > 50);
        for (int i = 0; i < cnt && i < (int)got.size(); i++) {
            CHECK(got[i].keypoint_index == (size_t)mq[i]);
            CHECK(&got[i].point == &km[(size_t)mt[i]].point);
        }
        std::printf("match_descriptors: %d matches\n", cnt);
    }

    // ---- a6: triangulate_points(frame1, frame2, matches) vs oracle
    {
        const KeyFrame& f1 = *kfs[0];
        const KeyFrame& f2 = *kfs[NKF - 1];
        std::vector<FeatureMatch> matches;
        for (size_t i = 0; i < kp_landmark[0].size(); i++)
            for (size_t j = 0; j < kp_landmark[NKF - 1].size(); j++)
                if (kp_landmark[0][i] == kp_landmark[NKF - 1][j]) matches.emplace_back((int)i, (int)j);
        auto got = triangulation::triangulate_points(f1, f2, matches, camera);
        auto pts = triangulation::get_matching_points(f1.features(), f2.features(), matches);
        const int n = (int)matches.size();
        std::vector<float> uv1(2 * n), uv2(2 * n), ps(32), xyz(3 * n), oxyz(3 * n);
        for (int i = 0; i < n; i++) { uv1[2 * i] = pts.first[i].x; uv1[2 * i + 1] = pts.first[i].y; uv2[2 * i] = pts.second[i].x; uv2[2 * i + 1] = pts.second[i].y; }
        for (int i = 0; i < 16; i++) { ps[i] = f1.pose()[i]; ps[16 + i] = f2.pose()[i]; }
        std::vector<uint8_t> keep(n);
        std::vector<int32_t> oi(n);
        int32_t cnt = 0;
        const float K[4] = {500.f, 500.f, 320.f, 240.f};
        orc_triangulate(uv1.data(), uv2.data(), n, ps.data(), 2, nullptr, nullptr, K, 0.9999f, 2.0f, xyz.data(), keep.data(), oi.data(), oxyz.data(), &cnt);
        CHECK((int)got.size() == cnt);
        CHECK(cnt > 20);
        for (int i = 0; i < cnt && i < (int)got.size(); i++) {
            CHECK(got[i].match_index == oi[i]);
            CHECK(got[i].position.x == oxyz[3 * i] && got[i].position.y == oxyz[3 * i + 1] && got[i].position.z == oxyz[3 * i + 2]);
        }
        CHECK(triangulation::triangulate_points({}, {}, f1.pose(), f2.pose(), camera).empty());   // empty guard
        std::printf("triangulate_points: %d of %d kept\n", cnt, n);
    }

    // ---- a2: match_map / match_key_frame / match_for_fuse: well-formed and consistent
    {
        MapMatcher matcher(camera, 64.f, NORM_HAMMING);
        auto mm = matcher.match_map(new_frame, map);
        CHECK(mm.size() > 30);
        size_t correct = 0;
        for (size_t i = 0; i < mm.size(); i++) {
            if (i) CHECK(mm[i].keypoint_index > mm[i - 1].keypoint_index);   // ascending keypoint order
            const int lm = kp_landmark[NKF][mm[i].keypoint_index];
            if (point_of[lm] >= 0 && &map[(size_t)point_of[lm]] == &mm[i].point) correct++;
        }
        CHECK(correct * 10 > mm.size() * 9);
        auto mk = matcher.match_key_frame(new_frame, map, kfs[NKF - 1].get());
        CHECK(mk.size() <= mm.size() && !mk.empty());
        for (auto& m : mk) CHECK(m.point.is_observed_by(kfs[NKF - 1].get()));
        std::vector<MapPoint*> some;
        for (size_t i = 0; i < map.size(); i += 2) some.push_back(&map[i]);
        some.push_back(nullptr);                                            // nulls are skipped (:121-123)
        auto mf = matcher.match_for_fuse(new_frame, some);
        CHECK(!mf.empty());
        std::printf("match_map: %zu (%zu correct), match_key_frame: %zu, match_for_fuse: %zu\n", mm.size(), correct, mk.size(), mf.size());
        for (auto& m : mm) new_frame.add_map_match(m);
    }

    // ---- a13: refine_pose on a perturbed pose moves it back
    {
        Mat4f good = new_frame.pose();
        Mat4f bad = make_pose(0.02 * NKF + 0.01, 0.05 * NKF + 0.03, 0.02, 0.4 * NKF - 0.05);
        new_frame.set_pose(bad);
        const bool ok = optimization::refine_pose(new_frame, camera);
        CHECK(ok);
        double err = 0;
        for (int i = 0; i < 12; i++) err = std::fmax(err, std::fabs(new_frame.pose()[i] - good[i]));
        CHECK(err < 0.02);
        CHECK(optimization::last_summary().final_cost < optimization::last_summary().initial_cost);
        std::printf("refine_pose: max pose error %.4f\n", err);
    }

    // ---- §8(f)1: the body of Mapper::triangulate_tracks vs oracle
    {
        // tracks = the landmarks the new frame sees that are NOT map points yet would be the real case; here every
        // landmark seen by the last key frame and by >= 2 earlier frames forms a track ending at that key frame
        // a copy of the last key frame whose key points are still unmatched (except a few, to exercise the skip rule)
        static std::shared_ptr<KeyFrame> kf_keep;        // the map keeps observer pointers: outlive this block
        kf_keep = std::make_shared<KeyFrame>(Frame(99, kfs[NKF - 1]->features()));
        KeyFrame& kf = *kf_keep;
        kf.set_pose(kfs[NKF - 1]->pose());
        for (size_t j = 0; j < 8 && j < kp_landmark[NKF - 1].size(); j++)
            if (point_of[kp_landmark[NKF - 1][j]] >= 0) map.associate(kf, map[(size_t)point_of[kp_landmark[NKF - 1][j]]], j);
        std::vector<tracks::Track> tr;
        for (size_t j = 0; j < kp_landmark[NKF - 1].size(); j++) {
            tracks::Track t;
            t.keypoint_index = j;
            for (int k = 0; k < NKF - 1; k++)
                for (size_t i = 0; i < kp_landmark[k].size(); i++)
                    if (kp_landmark[k][i] == kp_landmark[NKF - 1][j])
                        t.sightings.push_back(tracks::TrackSighting{(size_t)k, kfs[k]->keypoint(i).pt});
            tr.push_back(t);
        }
        std::vector<Mat4f> traj(poses.begin(), poses.begin() + NKF - 1);
        auto sel = tracks::select_track_points(kf, tr, traj, camera, 40);
        // oracle on the same marshalled arrays
        const int T = (int)tr.size();
        std::vector<float> tuv(2 * T), suv, ps(16 * NKF);
        std::vector<uint8_t> skip(T), status(T);
        std::vector<int32_t> sptr(T + 1, 0), spose, acc(T), inc(T);
        for (int t = 0; t < T; t++) {
            skip[t] = (kf.is_matched(tr[t].keypoint_index) || tr[t].sightings.empty()) ? 1 : 0;
            tuv[2 * t] = kf.keypoint(tr[t].keypoint_index).pt.x; tuv[2 * t + 1] = kf.keypoint(tr[t].keypoint_index).pt.y;
            for (auto& sg : tr[t].sightings) { spose.push_back((int32_t)sg.frame_index); suv.push_back(sg.pixel.x); suv.push_back(sg.pixel.y); }
            sptr[t + 1] = (int32_t)spose.size();
        }
        for (int k = 0; k < NKF - 1; k++) for (int i = 0; i < 16; i++) ps[16 * k + i] = traj[k][i];
        for (int i = 0; i < 16; i++) ps[16 * (NKF - 1) + i] = kf.pose()[i];
        std::vector<float> xyz(3 * T), pc(T), rc(T);
        int32_t na = 0, ntop = 0, ninc = 0;
        const float K[4] = {500.f, 500.f, 320.f, 240.f};
        orc_triangulate_tracks(T, tuv.data(), skip.data(), sptr.data(), spose.data(), suv.data(), ps.data(), NKF, NKF - 1, K,
                               1.0f, 4.0f, 0.999848f, 0.20f, 40, status.data(), xyz.data(), pc.data(), rc.data(), acc.data(),
                               &na, &ntop, inc.data(), &ninc);
        CHECK((int)sel.accepted.size() == na);
        CHECK((int)sel.inconsistent.size() == ninc);
        CHECK(na >= 20);
        for (int i = 0; i < na && i < (int)sel.accepted.size(); i++) {
            CHECK((int)sel.accepted[i].track == acc[i]);
            CHECK(sel.accepted[i].position.x == xyz[3 * acc[i]] && sel.accepted[i].position.z == xyz[3 * acc[i] + 2]);
            CHECK(sel.accepted[i].keypoint_index == tr[(size_t)acc[i]].keypoint_index);
        }
        // matched key points were skipped: every accepted track's key point is unmatched in the key frame
        for (auto& c : sel.accepted) CHECK(!kf.is_matched(c.keypoint_index));
        CHECK(tracks::select_track_points(kf, {}, traj, camera).accepted.empty());
        std::printf("select_track_points: %d tracks, %d accepted (%zu topped up), %d inconsistent\n", T, na, sel.topped_up, ninc);
    }

    // ---- §8(f)3: Mapper::cull_points arithmetic vs oracle (before BA, on the perturbed map)
    {
        std::vector<MapPoint*> pts;
        for (size_t i = 0; i < map.size(); i++) pts.push_back(&map[i]);
        auto res = tracks::point_errors(pts, camera, 1.0f);
        // oracle on the same marshalling: pose table in first-seen order
        std::vector<float> pos, uv, ps;
        std::vector<int32_t> optr{0}, opose;
        std::vector<const KeyFrame*> frames;
        for (auto* pt : pts) {
            pos.insert(pos.end(), {pt->position().x, pt->position().y, pt->position().z});
            for (auto& ob : pt->observations()) {
                size_t f = 0;
                while (f < frames.size() && frames[f] != ob.first) f++;
                if (f == frames.size()) { frames.push_back(ob.first); ps.insert(ps.end(), ob.first->pose().begin(), ob.first->pose().end()); }
                opose.push_back((int32_t)f);
                uv.push_back(ob.first->keypoint(ob.second).pt.x); uv.push_back(ob.first->keypoint(ob.second).pt.y);
            }
            optr.push_back((int32_t)opose.size());
        }
        const int P = (int)pts.size();
        std::vector<float> mean(P);
        std::vector<uint8_t> cull(P);
        std::vector<int32_t> idx(P);
        int32_t cnt = 0;
        double sums[2];
        const float K[4] = {500.f, 500.f, 320.f, 240.f};
        orc_point_errors(P, pos.data(), optr.data(), opose.data(), uv.data(), ps.data(), (int)frames.size(), K, 1.0f, mean.data(),
                         cull.data(), idx.data(), &cnt, sums);
        CHECK((int)res.to_remove.size() == cnt);
        CHECK(cnt > 0 && cnt < P);
        for (int i = 0; i < P; i++) CHECK(res.mean_error[(size_t)i] == mean[(size_t)i]);
        for (int i = 0; i < cnt && i < (int)res.to_remove.size(); i++) CHECK((int)res.to_remove[(size_t)i] == idx[(size_t)i]);
        CHECK(res.observations == (size_t)sums[1]);
        CHECK(std::fabs(res.error_sum - sums[0]) < 1e-9 * sums[0]);
        std::printf("point_errors: %d of %d points above 1 px, mean reprojection %.3f px\n", cnt, P, res.error_sum / (double)res.observations);
    }

    // ---- a8 + a12: build_local_window + bundle_adjust
    {
        auto window = optimization::build_local_window(kfs, new_frame, 20);
        CHECK(window.size() == (size_t)NKF + 1);
        CHECK(!window[0].optimize && !window[1].optimize && window[2].optimize && window.back().optimize);
        CHECK(window.back().frame == &new_frame);
        // perturb the free frames, then adjust
        for (size_t i = 2; i < window.size(); i++) {
            Mat4f T = window[i].frame->pose();
            T[3] += 0.02f; T[11] -= 0.03f;
            window[i].frame->set_pose(T);
        }
        std::vector<Mat4f> before;
        for (auto& fc : window) before.push_back(fc.frame->pose());
        const bool ok = optimization::bundle_adjust(window, camera, map);
        CHECK(ok);
        CHECK(optimization::last_summary().usable == 1);
        CHECK(optimization::last_summary().final_cost < 0.5 * optimization::last_summary().initial_cost);
        for (size_t i = 0; i < 2; i++)
            for (int k = 0; k < 16; k++) CHECK(window[i].frame->pose()[k] == before[i][k]);        // fixed frames untouched
        double moved = 0;
        for (int k = 0; k < 16; k++) moved = std::fmax(moved, std::fabs(window[3].frame->pose()[k] - before[3][k]));
        CHECK(moved > 1e-4);
    }

    std::printf(g_fail ? "FAILED: %d checks\n" : "host mirror: all checks passed\n", g_fail);
    return g_fail ? 1 : 0;
}
