// rs_shim_common.h — helpers shared by the four drop-in translation units.
// NOT COMPILED IN THIS REPO: these files include the reference's own headers plus Eigen / OpenCV,
// none of which exists in the build image.  They are the binding a maintainer of
// GregVS/Racing-SLAM adds; the same marshalling is compiled and tested here on plain types in
// racing-slam_amd/host/slam_host.cpp.
#pragma once
#include <hip/hip_runtime_api.h>

#include <Eigen/Dense>
#include <cstdio>
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

template <typename T>
struct DevBuf {
    T* p = nullptr;
    explicit DevBuf(size_t n) { (void)hipMalloc((void**)&p, sizeof(T) * (n ? n : 1)); }
    explicit DevBuf(const std::vector<T>& h) : DevBuf(h.size()) { if (!h.empty()) (void)hipMemcpy(p, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice); }
    ~DevBuf() { (void)hipFree(p); }
    std::vector<T> download(size_t n) const { std::vector<T> h(n); if (n) (void)hipMemcpy(h.data(), p, sizeof(T) * n, hipMemcpyDeviceToHost); return h; }
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

// no error codes in the reference: log to stdout like it does, caller returns {} / false
inline bool ok(int rc, const char* what)
{
    if (rc == RS_OK) return true;
    std::printf("%s failed: %s\n", what, rs_last_error(context()));
    return false;
}

}  // namespace rs_shim
