// Drop-in replacement of the reference's src/LocalWindow.cpp (keeps src/LocalWindow.h).
// NOT COMPILED IN THIS REPO; see rs_shim_common.h.  Pure pointer-set logic: host only.
#include "LocalWindow.h"

#include <unordered_map>

#include "Frame.h"
#include "MapPoint.h"
#include "rsgpu.h"

namespace slam::optimization {

std::vector<FrameConfig> build_local_window(const std::vector<std::shared_ptr<KeyFrame>>& key_frames, Frame& new_frame,
                                            size_t window_size, bool fix_oldest)
{
    const int n = (int)key_frames.size();
    int new_index = -1;
    std::unordered_map<const Frame*, int> fid;
    for (int i = 0; i < n; i++) { fid[key_frames[i].get()] = i; if (key_frames[i].get() == &new_frame) new_index = i; }
    std::unordered_map<const MapPoint*, int> pid;
    std::vector<const MapPoint*> pts;
    std::vector<int32_t> frame_ptr(n + 2, 0), frame_pt;
    auto add = [&](const Frame& f, int slot) {
        for (const auto& m : f.map_matches()) {
            auto it = pid.find(&m.point);
            if (it == pid.end()) { it = pid.emplace(&m.point, (int)pts.size()).first; pts.push_back(&m.point); }
            frame_pt.push_back(it->second);
        }
        frame_ptr[slot + 1] = (int32_t)frame_pt.size();
    };
    for (int i = 0; i < n; i++) add(*key_frames[i], i);
    if (new_index < 0) add(new_frame, n); else frame_ptr[n + 1] = frame_ptr[n];
    std::vector<int32_t> pt_ptr(pts.size() + 1, 0), pt_obs;
    for (size_t p = 0; p < pts.size(); p++) {
        for (const auto& [kf, idx] : pts[p]->observations()) {
            auto it = fid.find(kf);
            if (it != fid.end()) pt_obs.push_back(it->second);
        }
        pt_ptr[p + 1] = (int32_t)pt_obs.size();
    }
    std::vector<int32_t> of(n + 1);
    std::vector<uint8_t> oo(n + 1);
    int32_t cnt = 0;
    rs_build_local_window(n, new_index, (int)window_size, fix_oldest, frame_ptr.data(), frame_pt.data(), pt_ptr.data(), pt_obs.data(),
                          of.data(), oo.data(), &cnt);
    std::vector<FrameConfig> out;
    for (int i = 0; i < cnt; i++) out.push_back({oo[i] != 0, of[i] == n ? &new_frame : static_cast<Frame*>(key_frames[of[i]].get())});
    return out;
}

}  // namespace slam::optimization
