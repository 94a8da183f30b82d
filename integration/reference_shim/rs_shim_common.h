// rs_shim_common.h — helpers shared by the drop-in translation units of this directory.
//
// These files include the reference's own headers plus Eigen / OpenCV, none of which exists in this repository's build
// image: they are the binding a maintainer of GregVS/Racing-SLAM adds to that tree.  Here they are
// syntax-checked against the reference's real headers with stand-in Eigen / OpenCV declarations
// (tests/test_shim_syntax.py), and the same marshalling is compiled, run and checked on plain types in
// racing-slam_amd/host/slam_host.cpp.  The only dependency besides the reference tree is include/rsgpu.h.
#pragma once
#include <Eigen/Dense>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "rsgpu.h"

namespace rs_shim {

inline rs_context* context()
{
    static rs_context* ctx = [] {
        rs_context* c = nullptr;
        if (rs_context_create(0, &c) != RS_OK) { std::printf("rsgpu: no MI355X visible, aborting (no CPU fallback)\n"); std::abort(); }
        return c;
    }();
    return ctx;
}

// no error codes in the reference: log to stdout like it does, caller returns {} / false
inline bool ok(int rc, const char* what)
{
    if (rc == RS_OK) return true;
    std::printf("%s failed: %s\n", what, rs_last_error(context()));
    return false;
}

// One interface call = one staging group (include/rsgpu.h, "staging pool"): grow-only device + pinned arenas,
// asynchronous copies on the context stream, ONE synchronisation per call.  No hipMalloc / hipFree per array.
struct Stage {
    Stage() { ok(rs_stage_begin(context()), "rs_stage_begin"); }
    void sync() { ok(rs_stage_sync(context()), "rs_stage_sync"); }
};

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    explicit DevBuf(size_t count) : n(count) { ok(rs_stage_alloc(context(), sizeof(T) * (count ? count : 1), (void**)&p), "rs_stage_alloc"); }
    explicit DevBuf(const std::vector<T>& h) : n(h.size()) { ok(rs_stage_upload(context(), h.data(), sizeof(T) * h.size(), (void**)&p), "rs_stage_upload"); }
    DevBuf(const T* h, size_t count) : n(count) { ok(rs_stage_upload(context(), h, sizeof(T) * count, (void**)&p), "rs_stage_upload"); }
    // registers an asynchronous read-back of the first `count` entries; filled when Stage::sync() returns
    std::vector<T> fetch(size_t count) const
    {
        std::vector<T> h(count);
        if (count) ok(rs_stage_download(context(), p, sizeof(T) * count, h.data()), "rs_stage_download");
        return h;
    }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};

// Eigen is column-major, the C-ABI row-major: transpose 16 floats.
inline void pose_to_row_major(const Eigen::Matrix4f& T, float out[16])
{
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) out[4 * i + j] = T(i, j);
}
inline Eigen::Matrix4f pose_from_row_major(const float in[16])
{
    Eigen::Matrix4f T;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) T(i, j) = in[4 * i + j];
    return T;
}
inline void append_row_major(const Eigen::Matrix4f& T, std::vector<float>& table)
{
    table.resize(table.size() + 16);
    pose_to_row_major(T, table.data() + table.size() - 16);
}
inline void intrinsics(const Eigen::Matrix3f& M, float K[4])
{
    K[0] = M(0, 0); K[1] = M(1, 1); K[2] = M(0, 2); K[3] = M(1, 2);
}

}  // namespace rs_shim
