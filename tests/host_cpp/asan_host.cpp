// AddressSanitizer run of the HOST-ONLY entry points of librsgpu (csrc/host.cpp, csrc/pose_graph.cpp compiled with g++
// -fsanitize=address; no GPU, no HIP): every input and output lives in an exactly-sized heap block, so a read or write one
// element past an array — the kind that only crashes under a different heap layout — aborts the test.
// (rs_build_local_window read frame_ptr[n + 1] for a key-frame call until rocprofv3's heap made it fault.)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../include/rsgpu.h"

extern "C" void rs_ba_default_options(rs_ba_options* o)      // defined in ba.hip (a GPU translation unit); same values
{
    o->max_num_iterations = 10; o->huber_delta = std::sqrt(5.991); o->initial_trust_region_radius = 1e4;
    o->max_trust_region_radius = 1e16; o->min_trust_region_radius = 1e-32; o->min_relative_decrease = 1e-3;
    o->min_lm_diagonal = 1e-6; o->max_lm_diagonal = 1e32; o->function_tolerance = 1e-6; o->gradient_tolerance = 1e-10;
    o->parameter_tolerance = 1e-8; o->max_num_consecutive_invalid_steps = 5; o->jacobi_scaling = 1;
}

template <typename T> struct Exact {        // heap block of exactly n elements
    T* p; size_t n;
    explicit Exact(size_t count) : p((T*)malloc(sizeof(T) * (count ? count : 1))), n(count) { memset(p, 0, sizeof(T) * (count ? count : 1)); }
    Exact(const std::vector<T>& v) : Exact(v.size()) { if (n) memcpy(p, v.data(), sizeof(T) * n); }
    ~Exact() { free(p); }
};

static void make_pose(std::mt19937& g, float T[16], double x, double z)
{
    std::normal_distribution<double> nd(0.0, 0.05);
    const double a = nd(g) + 0.01 * x;
    const double c = cos(a), s = sin(a);
    const float R[9] = {(float)c, 0.f, (float)s, 0.f, 1.f, 0.f, (float)-s, 0.f, (float)c};
    const float ctr[3] = {(float)x, 0.f, (float)z};
    for (int r = 0; r < 3; r++) {
        for (int k = 0; k < 3; k++) T[4 * r + k] = R[3 * r + k];
        T[4 * r + 3] = -(R[3 * r] * ctr[0] + R[3 * r + 1] * ctr[1] + R[3 * r + 2] * ctr[2]);
    }
    T[12] = T[13] = T[14] = 0.f; T[15] = 1.f;
}

int main()
{
    std::mt19937 g(7);
    // ---- local window: key-frame call (n + 1 row pointers) and non-key-frame call (n + 2)
    for (int mode = 0; mode < 2; mode++) {
        const int n = 30, P = 400, rows = mode == 0 ? n : n + 1;
        std::vector<std::vector<int>> fp(rows);
        std::vector<std::vector<int>> po(P);
        std::uniform_int_distribution<int> start(0, n - 4), len(2, 4);
        for (int p = 0; p < P; p++) {
            const int s = start(g), l = len(g);
            for (int k = s; k < s + l && k < n; k++) { fp[k].push_back(p); po[p].push_back(k); }
            if (mode == 1 && p % 7 == 0) fp[n].push_back(p);
        }
        std::vector<int32_t> frame_ptr(1, 0), frame_pt, pt_ptr(1, 0), pt_obs;
        for (auto& v : fp) { frame_pt.insert(frame_pt.end(), v.begin(), v.end()); frame_ptr.push_back((int32_t)frame_pt.size()); }
        for (auto& v : po) { pt_obs.insert(pt_obs.end(), v.begin(), v.end()); pt_ptr.push_back((int32_t)pt_obs.size()); }
        Exact<int32_t> a(frame_ptr), b(frame_pt), c(pt_ptr), d(pt_obs), of(n + 1), cnt(1);
        Exact<uint8_t> oo(n + 1);
        for (int win : {5, 20, 40})
            for (int fix = 0; fix < 2; fix++)
                if (rs_build_local_window(n, mode == 0 ? n - 1 : -1, win, fix, a.p, b.p, c.p, d.p, of.p, oo.p, cnt.p) != RS_OK || cnt.p[0] <= 0) { printf("local window failed\n"); return 1; }
    }
    // ---- pose packing, KD-tree
    {
        const int n = 257;
        Exact<float> poses(16 * n), back(16 * n), kp(2 * n);
        Exact<double> cams(6 * n);
        Exact<uint8_t> mask(n);
        for (int i = 0; i < n; i++) { make_pose(g, poses.p + 16 * i, 0.3 * i, 0.1 * i); mask.p[i] = i & 1; }
        rs_pack_poses(poses.p, n, cams.p);
        rs_unpack_poses(cams.p, n, mask.p, back.p);
        rs_unpack_poses(cams.p, n, nullptr, back.p);
        std::uniform_int_distribution<int> px(0, 600);
        for (int i = 0; i < 2 * n; i++) kp.p[i] = (float)px(g);          // integer pixels: ties
        Exact<int32_t> node(n), left(n), right(n);
        int32_t root = -1;
        if (rs_kdtree_build(kp.p, n, node.p, left.p, right.p, &root) != RS_OK || root < 0) { printf("kdtree failed\n"); return 1; }
    }
    // ---- pose graph: SE(3) and 4-DoF, loops incl. entries to be skipped, partial last block sizes
    for (int n : {3, 17, 64}) {
        Exact<float> poses(16 * n), out(16 * n), rot(9 * n);
        for (int i = 0; i < n; i++) make_pose(g, poses.p + 16 * i, 2.0 * cos(6.28 * i / n * 1.3), 2.0 * sin(6.28 * i / n * 1.3));
        std::vector<rs_pose_graph_edge> loops;
        for (int i = n / 2 + 1; i < n; i += 3) {
            rs_pose_graph_edge e; e.from = i; e.to = i - n / 2; rs_pose_relative(poses.p + 16 * i, poses.p + 16 * (i - n / 2), e.relative);
            e.relative[3] += 0.05; loops.push_back(e);
        }
        rs_pose_graph_edge bad; bad.from = n + 3; bad.to = 0; for (int k = 0; k < 16; k++) bad.relative[k] = (k % 5 == 0); loops.push_back(bad);
        Exact<rs_pose_graph_edge> L(loops);
        Exact<rs_ba_iteration> trace(4);             // deliberately shorter than the iteration count
        rs_ba_summary sum;
        int tc = 0;
        const double grav[3] = {0.0, -9.8, 0.0};
        for (int fd = 0; fd < 2; fd++)
            if (rs_pose_graph(n, poses.p, L.p, (int)loops.size(), fd, grav, nullptr, out.p, rot.p, &sum, trace.p, 4, &tc) != RS_OK) { printf("pose graph failed\n"); return 1; }
    }
    printf("asan host checks passed\n");
    return 0;
}
