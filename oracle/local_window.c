/*
 * oracle/local_window.c — CPU restatement of optimization::build_local_window
 * (reference src/LocalWindow.cpp:10-52).  TEST INFRASTRUCTURE ONLY.
 * PARITY UNPINNED (see rs_oracle.h).  Pure set logic, no arithmetic.
 *
 * Frames are indices: 0..n_key_frames-1 are Mapper's key frames in order;
 * index n_key_frames stands for a new frame that is not a key frame yet
 * (new_frame == -1).
 */
#include <stdlib.h>
#include <string.h>

#include "rs_oracle.h"

int orc_build_local_window(int n_key_frames, int new_frame, int window_size, int fix_oldest,
                           const int32_t* frame_ptr, const int32_t* frame_pt,
                           const int32_t* pt_ptr, const int32_t* pt_obs, int32_t* out_frame,
                           uint8_t* out_optimize, int32_t* out_count)
{
    if (n_key_frames < 0 || window_size < 0 || new_frame >= n_key_frames) return 1;
    const int n = n_key_frames;
    const int new_id = new_frame >= 0 ? new_frame : n;
    /* :15 */
    const int first_optimized = n > window_size ? n - window_size : 2;
    uint8_t* window = (uint8_t*)calloc((size_t)n + 1, 1);
    uint8_t* anchors = (uint8_t*)calloc((size_t)n + 1, 1);
    window[new_id] = 1;                                         /* :16 */
    for (int i = first_optimized; i < n; i++) window[i] = 1;    /* :17-19 */
    /* :21-30 anchors = observers of window-matched points that are outside the window */
    for (int f = 0; f <= n; f++) {
        if (!window[f]) continue;
        for (int a = frame_ptr[f]; a < frame_ptr[f + 1]; a++) {
            int p = frame_pt[a];
            for (int o = pt_ptr[p]; o < pt_ptr[p + 1]; o++) {
                int obs = pt_obs[o];
                if (!window[obs]) anchors[obs] = 1;
            }
        }
    }
    int count = 0, included = 0;
    for (int i = 0; i < n; i++) {                               /* :35-47 */
        int fixed = i < 2;
        if (fix_oldest && i == first_optimized) fixed = 1;
        if (window[i]) {
            out_frame[count] = i; out_optimize[count] = (uint8_t)!fixed; count++;
            included = included || (i == new_id);
        } else if (fixed || anchors[i]) {
            out_frame[count] = i; out_optimize[count] = 0; count++;
        }
    }
    if (!included) { out_frame[count] = new_id; out_optimize[count] = 1; count++; }   /* :48-50 */
    *out_count = count;
    free(window);
    free(anchors);
    return 0;
}
