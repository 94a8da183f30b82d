// reproj_match.hip — K2/K3: reprojection-gated map-point <-> keypoint matching.
//
// Replaces MapMatcher::match / match_map / match_key_frame / match_for_fuse
// (reference src/MapMatcher.cpp:45-98,107-127,165-175) together with the
// per-point helpers it calls: Camera::project / is_in_image (src/Camera.cpp:25-37),
// MapPoint::avg_viewing_normal / observed_distance_range (src/MapPoint.cpp:24-45)
// and KDTree2D::radius_search (src/KDTree.cpp:45-82).
//
// K2: eight lanes per map point when the frame's KD-tree fits in LDS (k2_reproj_match_grouped below, the default),
// one lane per point otherwise (k2_reproj_match).  Projection and the three f32 gates, then the
// KD-tree radius search walked with an explicit stack kept in LDS, visiting
// nodes in exactly the reference's recursion order (node, near child, far
// child) so that "first candidate wins ties" is preserved.  Every accepted
// candidate is compared against all observations of the point with 8 xor +
// 8 v_bcnt.  The winner per point goes into a packed 64-bit atomicMin on the
// keypoint's slot (dist << 32 | map order), which reproduces the reference's
// sequential strict-'<' proposal table: min distance, earliest point on ties.
// K3: one workgroup decodes the table and emits accepted_matches() in
// ascending keypoint order.
// Built with -ffp-contract=off so the f32 gates execute the oracle's operations.
#include "common.h"

#define K2_THREADS 128
#define K2_STACK 40
#define K2_PRE 8         // observations of a map point whose centre / descriptor row are preloaded
#define K2_MAXC 16       // in-radius candidates queued per map point before their descriptors are compared (phase B)
#define K2_MAX_LDS_NODES 6144    // 16 B per node: the whole KD-tree of a frame (2000 keypoints = 32 KB) sits in LDS

struct K2Frame {
    float T[16];
    float fx, fy, cx, cy;
    int width, height, n_keypoints, kd_root;
    const float2* kp;
    const uint4* desc;
    const uint8_t* kp_matched;
    const int32_t* kd_node_kp;
    const int32_t* kd_left;
    const int32_t* kd_right;
    const float4* packed;      // optional: {x, y, left, right}[n] then int32 keypoint[n] (rs_kdtree_pack), or null
};

struct K2Map {
    int n_points;
    int point_base;          // map order of this view's point 0 (a shard of a larger map: rs_reproj_match_sharded); 0 otherwise
    const float* pos;
    const uint8_t* eligible;
    const int32_t* obs_ptr;
    const int32_t* obs_kf;
    const int32_t* obs_desc;
    const float* kf_centers;
    const uint4* pool;
};

__device__ __forceinline__ float dot3f(const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

__device__ __forceinline__ void normalize3f(float* v)
{
    const float n = dot3f(v, v);
    if (n > 0.0f) {
        const float s = sqrtf(n);
        v[0] = v[0] / s; v[1] = v[1] / s; v[2] = v[2] / s;
    }
}

__device__ __forceinline__ int hamming256(const uint4 a0, const uint4 a1, const uint4 b0, const uint4 b1)
{
    int d = __builtin_popcount(a0.x ^ b0.x);
    d += __builtin_popcount(a0.y ^ b0.y);
    d += __builtin_popcount(a0.z ^ b0.z);
    d += __builtin_popcount(a0.w ^ b0.w);
    d += __builtin_popcount(a1.x ^ b1.x);
    d += __builtin_popcount(a1.y ^ b1.y);
    d += __builtin_popcount(a1.z ^ b1.z);
    d += __builtin_popcount(a1.w ^ b1.w);
    return d;
}

// Ordered acceptance of the proposal table (accepted_matches, src/MapMatcher.cpp:34-43) by ONE workgroup
// of any size; resets the table to all-ones for the next call.
__device__ __forceinline__ void k3_accept_body(unsigned long long* __restrict__ prop, int n, int max_distance,
                                               int32_t* __restrict__ prop_point, int32_t* __restrict__ prop_dist,
                                               int32_t* __restrict__ match_kp, int32_t* __restrict__ match_point,
                                               int32_t* __restrict__ match_count)
{
    // every thread owns a contiguous chunk of keypoints: count, one workgroup scan, ordered write.
    // Loads go out in batches of 8 (clamped addresses, no branches) so that their latencies overlap.
    const int T = blockDim.x, chunk = (n + T - 1) / T;
    const int lo = min((int)threadIdx.x * chunk, n), hi = min(lo + chunk, n);
    int cnt = 0;
    for (int i0 = lo; i0 < hi; i0 += 8) {
        unsigned long long v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = __builtin_nontemporal_load(&prop[min(i0 + u, hi - 1)]);
#pragma unroll
        for (int u = 0; u < 8; u++) cnt += (i0 + u < hi && v[u] != ~0ull) ? 1 : 0;
    }
    int total;
    int off = rs_block_exclusive_scan(cnt, &total);
    for (int i0 = lo; i0 < hi; i0 += 8) {
        unsigned long long v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = __builtin_nontemporal_load(&prop[min(i0 + u, hi - 1)]);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = i0 + u;
            if (i >= hi) break;
            int pt = -1, dist = max_distance;
            if (v[u] != ~0ull) {
                pt = (int)(unsigned)(v[u] & 0xFFFFFFFFull);
                dist = (int)(v[u] >> 32);
                match_kp[off] = i;                                       // accepted_matches :34-43
                match_point[off] = pt;
                off++;
                prop[i] = ~0ull;
            }
            prop_point[i] = pt;
            prop_dist[i] = dist;
        }
    }
    if (threadIdx.x == 0) *match_count = total;
}

// LDS node of the frame's KD-tree: everything a visit needs in ONE ds_read_b128.
//   x, y     keypoint of the node
//   lr       left | right << 16, 0xFFFF = no child (the tree has at most K2_MAX_LDS_NODES < 65535 nodes)
//   kp       keypoint index | (already matched && !replace) << 31   (:65, :81 become a sign test)
struct K2Node { float x, y; unsigned lr; int kp; };

__global__ __launch_bounds__(K2_THREADS) void k2_reproj_match(K2Frame f, K2Map m, int replace, int max_distance, int tree_in_lds,
                                                             int32_t* __restrict__ point_kp,
                                                             int32_t* __restrict__ point_dist,
                                                             unsigned long long* __restrict__ prop)
{
    // dynamic LDS: [K2_STACK][K2_THREADS] traversal stacks, [K2_MAXC][K2_THREADS] candidate queue and the candidates'
    // running minima, then the tree
    extern __shared__ __attribute__((aligned(16))) int k2_lds[];
    int (*stack)[K2_THREADS] = (int (*)[K2_THREADS])k2_lds;
    int (*queue)[K2_THREADS] = (int (*)[K2_THREADS])(k2_lds + K2_STACK * K2_THREADS);
    int (*qmin)[K2_THREADS] = (int (*)[K2_THREADS])(k2_lds + (K2_STACK + K2_MAXC) * K2_THREADS);
    K2Node* tree = (K2Node*)(k2_lds + (K2_STACK + 2 * K2_MAXC) * K2_THREADS);
    // this lane's map point: every load that needs only p goes out before the tree is staged
    const int p = blockIdx.x * K2_THREADS + threadIdx.x;
    const bool have = p < m.n_points;
    const int pc = have ? p : 0;
    const bool elig = have && m.n_points > 0 && m.eligible[pc] != 0;
    const int o0 = m.n_points > 0 ? m.obs_ptr[pc] : 0, o1 = m.n_points > 0 ? m.obs_ptr[pc + 1] : 0;
    const int nobs = elig ? o1 - o0 : 0;
    float X[3] = {0.f, 0.f, 0.f};
    if (m.n_points > 0) { X[0] = m.pos[3 * (size_t)pc]; X[1] = m.pos[3 * (size_t)pc + 1]; X[2] = m.pos[3 * (size_t)pc + 2]; }
    // the first K2_PRE observations of the point: keyframe centre and descriptor row, loaded as two
    // batches (not as nobs dependent pairs inside the loops below); further observations come in batches of K2_PRE too
    int rowv[K2_PRE];
    float Cv[K2_PRE][3];
    {
        int kfv[K2_PRE];
#pragma unroll
        for (int j = 0; j < K2_PRE; j++) {
            kfv[j] = 0; rowv[j] = 0;
            if (j < nobs) { kfv[j] = m.obs_kf[o0 + j]; rowv[j] = m.obs_desc[o0 + j]; }
        }
#pragma unroll
        for (int j = 0; j < K2_PRE; j++) {
            Cv[j][0] = Cv[j][1] = Cv[j][2] = 0.f;
            if (j < nobs) {
                const float* C = m.kf_centers + 3 * (size_t)kfv[j];
                Cv[j][0] = C[0]; Cv[j][1] = C[1]; Cv[j][2] = C[2];
            }
        }
    }
    if (tree_in_lds) {
        // eight nodes per thread and pass, all loads of a level issued together
        for (int base = threadIdx.x; base < f.n_keypoints; base += 8 * K2_THREADS) {
            int kpi[8], l[8], r[8];
            float x[8], y[8];
            bool taken[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int i = base + q * K2_THREADS;
                kpi[q] = 0; l[q] = -1; r[q] = -1; x[q] = 0.f; y[q] = 0.f;
                if (i < f.n_keypoints) {
                    if (f.packed) {        // packed once per frame (rs_kdtree_pack): straight copies instead of dependent gathers
                        const float4 nd = f.packed[i];
                        kpi[q] = ((const int*)(f.packed + f.n_keypoints))[i];
                        x[q] = nd.x; y[q] = nd.y; l[q] = __float_as_int(nd.z); r[q] = __float_as_int(nd.w);
                    } else {
                        kpi[q] = f.kd_node_kp[i]; l[q] = f.kd_left[i]; r[q] = f.kd_right[i];
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const bool in = base + q * K2_THREADS < f.n_keypoints;
                taken[q] = in && !replace && f.kp_matched[kpi[q]] != 0;
                if (in && !f.packed) { const float2 k = f.kp[kpi[q]]; x[q] = k.x; y[q] = k.y; }
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int i = base + q * K2_THREADS;
                if (i < f.n_keypoints) {
                    K2Node nd;
                    nd.x = x[q]; nd.y = y[q];
                    nd.lr = (unsigned)(l[q] & 0xFFFF) | ((unsigned)(r[q] & 0xFFFF) << 16);
                    nd.kp = kpi[q] | (taken[q] ? (int)0x80000000 : 0);
                    tree[i] = nd;
                }
            }
        }
        __syncthreads();
    }
    int out_kp = -1, out_d = max_distance;
    if (have) do {
        if (!elig) break;
        const float* T = f.T;
        // Camera::project (src/Camera.cpp:25-32): K * pose.block<3,4> first, then * homogeneous
        float uvw[3];
        {
            float KP[12];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                KP[0 * 4 + j] = (f.fx * T[0 * 4 + j] + 0.0f * T[1 * 4 + j]) + f.cx * T[2 * 4 + j];
                KP[1 * 4 + j] = (0.0f * T[0 * 4 + j] + f.fy * T[1 * 4 + j]) + f.cy * T[2 * 4 + j];
                KP[2 * 4 + j] = (0.0f * T[0 * 4 + j] + 0.0f * T[1 * 4 + j]) + 1.0f * T[2 * 4 + j];
            }
#pragma unroll
            for (int i = 0; i < 3; i++)
                uvw[i] = (KP[4 * i] * X[0] + KP[4 * i + 1] * X[1]) + (KP[4 * i + 2] * X[2] + KP[4 * i + 3] * 1.0f);
        }
        float u, v;
        if (uvw[2] < 0.0f) { u = -1.0f; v = -1.0f; }
        else { u = uvw[0] / uvw[2]; v = uvw[1] / uvw[2]; }
        if (!(u >= 0.0f && u < (float)f.width && v >= 0.0f && v < (float)f.height)) break;   // :58

        // Frame::camera_center = -R^T t (src/Frame.cpp:39-42)
        float center[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const float a[3] = {-T[0 * 4 + i], -T[1 * 4 + i], -T[2 * 4 + i]};
            const float t[3] = {T[3], T[7], T[11]};
            center[i] = dot3f(a, t);
        }
        const float ray[3] = {X[0] - center[0], X[1] - center[1], X[2] - center[2]};
        float normal[3] = {0.0f, 0.0f, 0.0f};
        float nearest = 3.402823466e+38f, furthest = 0.0f;
        auto add_obs = [&](const float Cx, const float Cy, const float Cz) {   // src/MapPoint.cpp:24-45
            float d[3] = {X[0] - Cx, X[1] - Cy, X[2] - Cz};
            const float dist = sqrtf(dot3f(d, d));
            nearest = dist < nearest ? dist : nearest;
            furthest = furthest < dist ? dist : furthest;
            normalize3f(d);
            normalize3f(d);
            normal[0] += d[0]; normal[1] += d[1]; normal[2] += d[2];
        };
#pragma unroll
        for (int j = 0; j < K2_PRE; j++)
            if (j < nobs) add_obs(Cv[j][0], Cv[j][1], Cv[j][2]);
        for (int ob = K2_PRE; ob < nobs; ob += K2_PRE) {                  // long-lived points: K2_PRE centres per round trip,
            int kfv[K2_PRE];                                              // accumulated in observation order
            float Cx[K2_PRE][3];
#pragma unroll
            for (int j = 0; j < K2_PRE; j++) kfv[j] = (ob + j < nobs) ? m.obs_kf[o0 + ob + j] : 0;
#pragma unroll
            for (int j = 0; j < K2_PRE; j++) {
                const float* C = m.kf_centers + 3 * (size_t)kfv[j];
                Cx[j][0] = Cx[j][1] = Cx[j][2] = 0.f;
                if (ob + j < nobs) { Cx[j][0] = C[0]; Cx[j][1] = C[1]; Cx[j][2] = C[2]; }
            }
#pragma unroll
            for (int j = 0; j < K2_PRE; j++)
                if (ob + j < nobs) add_obs(Cx[j][0], Cx[j][1], Cx[j][2]);
        }
        normalize3f(normal);
        float rn[3] = {ray[0], ray[1], ray[2]};
        normalize3f(rn);
        if (dot3f(normal, rn) < 0.5f) break;                             // :62-66
        const float distance = sqrtf(dot3f(ray, ray));
        if (distance < nearest / 2.0f || distance > furthest * 1.25f) break;   // :69-73

        // KDTree2D::radius_search, r = 20 px (src/MapMatcher.cpp:75, src/KDTree.cpp:45-82)
        const float r2 = 20.0f * 20.0f;
        int best_kp = 0, best_d = max_distance;
        // One candidate keypoint against every descriptor of the point (:84-91).  The reference keeps a running best with
        // a strict '<' over (candidate, observation) pairs; the winner's keypoint is the FIRST candidate (in visiting
        // order) whose minimum over the observations is the overall minimum, so per candidate only that minimum matters
        // and the observations may be taken in any order.
        auto rows_min = [&](const uint4 a0, const uint4 a1, const uint4* b0, const uint4* b1, int nb) {
            int d = 0x7fffffff;
#pragma unroll
            for (int j = 0; j < K2_PRE; j++)
                if (j < nb) d = min(d, hamming256(a0, a1, b0[j], b1[j]));
            return d;
        };
        // Phase A: the traversal touches LDS only and QUEUES the candidates in visiting order.  (Comparing descriptors
        // inside this loop costs one global-memory round trip per iteration in which ANY lane of the wave has a
        // candidate — nearly every one; queued, the round trips are one per four candidates of the busiest lane.)
        // The near child is visited next without going through the stack; only far children are pushed.
        // Candidates beyond K2_MAXC are compared on the spot into a second running best, merged below: they come
        // later in visiting order than every queued one, so the queue wins ties.
        int nc = 0, over_kp = 0, over_d = max_distance;
        int sp = 0;
        int cur = f.kd_root, odd = 0;
        if (tree_in_lds) {
            // Both children of a node are read while the node is examined (their indices are in the node), so the LDS
            // latency of the next visit overlaps this visit's arithmetic; only a node popped from the stack pays a
            // read of its own.
            K2Node nd;
            if (cur >= 0) nd = tree[cur];
            while (cur >= 0) {
                int l = (int)(nd.lr & 0xFFFFu), r = (int)(nd.lr >> 16);
                l = l == 0xFFFF ? -1 : l; r = r == 0xFFFF ? -1 : r;
                const K2Node ndl = tree[l >= 0 ? l : 0], ndr = tree[r >= 0 ? r : 0];
                const float dx = nd.x - u, dy = nd.y - v;
                const float d2 = dx * dx + dy * dy;
                if (d2 <= r2 && nd.kp >= 0) {                               // in range and open (:65, :81)
                    const int kp = nd.kp;
                    if (nc < K2_MAXC) queue[nc++][threadIdx.x] = kp;
                    else {
                        const uint4 a0 = f.desc[2 * (size_t)kp], a1 = f.desc[2 * (size_t)kp + 1];
                        int d = 0x7fffffff;
                        for (int o = o0; o < o1; o++) {
                            const size_t row = (size_t)m.obs_desc[o];
                            d = min(d, hamming256(a0, a1, m.pool[2 * row], m.pool[2 * row + 1]));
                        }
                        if (d < over_d) { over_d = d; over_kp = kp; }
                    }
                }
                const float delta = odd ? dy : dx;
                const bool left_near = delta > 0;
                const int near_child = left_near ? l : r;
                const int far_child = left_near ? r : l;
                odd ^= 1;                                                 // depth parity of the children
                if (delta * delta <= r2 && far_child >= 0 && sp < K2_STACK) stack[sp++][threadIdx.x] = far_child | (odd << 30);
                if (near_child >= 0) { cur = near_child; nd = left_near ? ndl : ndr; }
                else if (sp > 0) { const int e = stack[--sp][threadIdx.x]; cur = e & 0x3FFFFFFF; odd = (e >> 30) & 1; nd = tree[cur]; }
                else cur = -1;
            }
        } else
        while (cur >= 0) {                                                // trees too large for LDS: the same walk on global memory
            const int kp = f.kd_node_kp[cur];
            const float2 q = f.kp[kp];
            const int l = f.kd_left[cur], r = f.kd_right[cur];
            const float dx = q.x - u, dy = q.y - v;
            const float d2 = dx * dx + dy * dy;
            if (d2 <= r2 && (replace || !f.kp_matched[kp])) {               // :65, :81
                if (nc < K2_MAXC) queue[nc++][threadIdx.x] = kp;
                else {
                    const uint4 a0 = f.desc[2 * (size_t)kp], a1 = f.desc[2 * (size_t)kp + 1];
                    int d = 0x7fffffff;
                    for (int o = o0; o < o1; o++) {
                        const size_t row = (size_t)m.obs_desc[o];
                        d = min(d, hamming256(a0, a1, m.pool[2 * row], m.pool[2 * row + 1]));
                    }
                    if (d < over_d) { over_d = d; over_kp = kp; }
                }
            }
            const float delta = odd ? dy : dx;
            const int near_child = (delta > 0) ? l : r;
            const int far_child = (delta > 0) ? r : l;
            odd ^= 1;
            if (delta * delta <= r2 && far_child >= 0 && sp < K2_STACK) stack[sp++][threadIdx.x] = far_child | (odd << 30);
            if (near_child >= 0) cur = near_child;
            else if (sp > 0) { const int e = stack[--sp][threadIdx.x]; cur = e & 0x3FFFFFFF; odd = (e >> 30) & 1; }
            else cur = -1;
        }
        // Phase B: K2_PRE descriptor rows of the point per pass (the first pass's row numbers are already here), the
        // queued candidates four at a time (their row loads go out together); each candidate's minimum over the
        // observations accumulates in LDS across passes
        if (nc > 0) {
            for (int ob = 0; ob < nobs; ob += K2_PRE) {
                const int nb = min(K2_PRE, nobs - ob);
                int rows[K2_PRE];
#pragma unroll
                for (int j = 0; j < K2_PRE; j++) rows[j] = ob == 0 ? rowv[j] : (j < nb ? m.obs_desc[o0 + ob + j] : 0);
                uint4 b0[K2_PRE], b1[K2_PRE];
#pragma unroll
                for (int j = 0; j < K2_PRE; j++) {
                    b0[j] = make_uint4(0, 0, 0, 0); b1[j] = b0[j];
                    if (j < nb) { b0[j] = m.pool[2 * (size_t)rows[j]]; b1[j] = m.pool[2 * (size_t)rows[j] + 1]; }
                }
                for (int c = 0; c < nc; c += 4) {
                    int kq[4];
                    uint4 a0[4], a1[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        kq[q] = (c + q < nc) ? queue[c + q][threadIdx.x] : -1;
                        a0[q] = make_uint4(0, 0, 0, 0); a1[q] = a0[q];
                        if (kq[q] >= 0) { a0[q] = f.desc[2 * (size_t)kq[q]]; a1[q] = f.desc[2 * (size_t)kq[q] + 1]; }
                    }
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        if (kq[q] >= 0) {
                            const int d = rows_min(a0[q], a1[q], b0, b1, nb);
                            qmin[c + q][threadIdx.x] = ob == 0 ? d : min(d, qmin[c + q][threadIdx.x]);
                        }
                }
            }
            if (nobs > 0)
                for (int c = 0; c < nc; c++) {
                    const int d = qmin[c][threadIdx.x];
                    if (d < best_d) { best_d = d; best_kp = queue[c][threadIdx.x]; }     // :88-91, visiting order
                }
        }
        if (over_d < best_d) { best_d = over_d; best_kp = over_kp; }
        if (best_d < max_distance) {
            out_kp = best_kp;
            out_d = best_d;
            // :95-97 sequential strict-'<' over map order == atomicMin of (dist, map order)
            atomicMin(&prop[best_kp], ((unsigned long long)(unsigned)best_d << 32) | (unsigned)(m.point_base + p));
        }
    } while (0);
    if (p < m.n_points) {
        point_kp[p] = out_kp;
        point_dist[p] = out_d;
    }
}

// K2g: the same search with EIGHT lanes per map point (trees that fit in LDS).  With one lane per point a 64-wide
// wave walks 64 trees in lock step and then compares every candidate against the point's descriptors one after the
// other; here a wave holds 8 points, lane 0 of a group walks the tree and queues the candidates, and the group's lanes
// each own ONE observation of the point: its keyframe centre (viewing-normal gate) and its descriptor row (loaded
// before the tree is staged), so a candidate costs one 256-bit compare per lane and a 3-step DPP minimum.
// Observation order of the normal's f32 sum, visiting order of the candidates and the strict '<' rules are those of
// k2_reproj_match above — the outputs are bit-identical (tests/test_gpu_parity.py runs both).
#define K2G 8

#ifndef RS_STAMPS
#define RS_STAMPS 0
#endif
#if RS_STAMPS
// diagnostic build (RS_STAMPS=1 python build.py): phase ends of the first wave of the middle workgroup, 10 ns ticks
__device__ unsigned long long k2_stamps[8];
#define K2_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) k2_stamps[i] = wall_clock64(); } while (0)
#define K2_STAMP0() K2_STAMP(0)
extern "C" int rs_k2_stamps(unsigned long long* h_out, int reset)
{
    if (hipMemcpyFromSymbol(h_out, HIP_SYMBOL(k2_stamps), sizeof(unsigned long long) * 8) != hipSuccess) return RS_ERR_HIP;
    if (reset) {
        unsigned long long z[8] = {~0ull, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(k2_stamps), z, sizeof z) != hipSuccess) return RS_ERR_HIP;
    }
    return RS_OK;
}
#else
#define K2_STAMP(i) do { } while (0)
#define K2_STAMP0() do { } while (0)
#endif

__device__ __forceinline__ int group8_min(int v)
{
    v = min(v, __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true));     // quad_perm [1,0,3,2]
    v = min(v, __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true));     // quad_perm [2,3,0,1]
    v = min(v, __builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, true));    // row_half_mirror: the other quad of the eight
#if K2G == 16
    v = min(v, __builtin_amdgcn_mov_dpp(v, 0x140, 0xf, 0xf, true));    // row_mirror: the other eight of the sixteen
#endif
    return v;
}

template <int K2G_THREADS_T>
__global__ __launch_bounds__(K2G_THREADS_T) void k2_reproj_match_grouped(K2Frame f, K2Map m, int replace, int max_distance,
                                                                       int32_t* __restrict__ point_kp,
                                                                       int32_t* __restrict__ point_dist,
                                                                       unsigned long long* __restrict__ prop)
{
    // dynamic LDS: [K2_STACK][K2G_PTS] traversal stacks, [K2_MAXC][K2G_PTS] candidate queue and minima, then the tree
    constexpr int K2G_PTS = K2G_THREADS_T / K2G;
    extern __shared__ __attribute__((aligned(16))) int k2_lds[];
    int (*stack)[K2G_PTS] = (int (*)[K2G_PTS])k2_lds;
    int (*queue)[K2G_PTS] = (int (*)[K2G_PTS])(k2_lds + K2_STACK * K2G_PTS);
    int (*qmin)[K2G_PTS] = (int (*)[K2G_PTS])(k2_lds + (K2_STACK + K2_MAXC) * K2G_PTS);
    K2Node* tree = (K2Node*)(k2_lds + (K2_STACK + 2 * K2_MAXC) * K2G_PTS);
    K2_STAMP0();
    const int g = threadIdx.x / K2G, sub = threadIdx.x % K2G;
    const int p = blockIdx.x * K2G_PTS + g;
    const bool have = p < m.n_points;
    const int pc = have ? p : 0;
    const bool elig = have && m.eligible[pc] != 0;
    const int o0 = m.obs_ptr[pc], o1 = m.obs_ptr[pc + 1];
    const int nobs = elig ? o1 - o0 : 0;
    const float X[3] = {m.pos[3 * (size_t)pc], m.pos[3 * (size_t)pc + 1], m.pos[3 * (size_t)pc + 2]};
    // this lane's observations of the first sixteen (sub and sub + 8): keyframe centre and descriptor row, in flight
    // while the tree is staged; later ones are fetched inside the passes below
    float C0[3] = {0.f, 0.f, 0.f}, C1[3] = {0.f, 0.f, 0.f};
    uint4 b0 = make_uint4(0, 0, 0, 0), b1 = b0, b2 = b0, b3 = b0;
    {
        const bool h0 = sub < nobs, h1 = K2G + sub < nobs;
        const int kf0 = h0 ? m.obs_kf[o0 + sub] : 0, kf1 = h1 ? m.obs_kf[o0 + K2G + sub] : 0;
        const size_t row0 = h0 ? (size_t)m.obs_desc[o0 + sub] : 0, row1 = h1 ? (size_t)m.obs_desc[o0 + K2G + sub] : 0;
        if (h0) {
            const float* C = m.kf_centers + 3 * (size_t)kf0;
            C0[0] = C[0]; C0[1] = C[1]; C0[2] = C[2];
            b0 = m.pool[2 * row0]; b1 = m.pool[2 * row0 + 1];
        }
        if (h1) {
            const float* C = m.kf_centers + 3 * (size_t)kf1;
            C1[0] = C[0]; C1[1] = C[1]; C1[2] = C[2];
            b2 = m.pool[2 * row1]; b3 = m.pool[2 * row1 + 1];
        }
    }
    for (int base = threadIdx.x; base < f.n_keypoints; base += 8 * K2G_THREADS_T) {
        int kpi[8], l[8], r[8];
        float x[8], y[8];
        bool taken[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int i = base + q * K2G_THREADS_T;
            kpi[q] = 0; l[q] = -1; r[q] = -1; x[q] = 0.f; y[q] = 0.f;
            if (i < f.n_keypoints) {
                if (f.packed) {
                    const float4 nd = f.packed[i];
                    kpi[q] = ((const int*)(f.packed + f.n_keypoints))[i];
                    x[q] = nd.x; y[q] = nd.y; l[q] = __float_as_int(nd.z); r[q] = __float_as_int(nd.w);
                } else {
                    kpi[q] = f.kd_node_kp[i]; l[q] = f.kd_left[i]; r[q] = f.kd_right[i];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const bool in = base + q * K2G_THREADS_T < f.n_keypoints;
            taken[q] = in && !replace && f.kp_matched[kpi[q]] != 0;
            if (in && !f.packed) { const float2 k = f.kp[kpi[q]]; x[q] = k.x; y[q] = k.y; }
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int i = base + q * K2G_THREADS_T;
            if (i < f.n_keypoints) {
                K2Node nd;
                nd.x = x[q]; nd.y = y[q];
                nd.lr = (unsigned)(l[q] & 0xFFFF) | ((unsigned)(r[q] & 0xFFFF) << 16);
                nd.kp = kpi[q] | (taken[q] ? (int)0x80000000 : 0);
                tree[i] = nd;
            }
        }
    }
    K2_STAMP(1);
    __syncthreads();
    K2_STAMP(2);
    // everything below is uniform within a group of eight lanes except where `sub` appears
    int out_kp = -1, out_d = max_distance;
    if (have) do {
        if (!elig) break;
        const float* T = f.T;
        float uvw[3];                                                    // Camera::project (src/Camera.cpp:25-32)
        {
            float KP[12];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                KP[0 * 4 + j] = (f.fx * T[0 * 4 + j] + 0.0f * T[1 * 4 + j]) + f.cx * T[2 * 4 + j];
                KP[1 * 4 + j] = (0.0f * T[0 * 4 + j] + f.fy * T[1 * 4 + j]) + f.cy * T[2 * 4 + j];
                KP[2 * 4 + j] = (0.0f * T[0 * 4 + j] + 0.0f * T[1 * 4 + j]) + 1.0f * T[2 * 4 + j];
            }
#pragma unroll
            for (int i = 0; i < 3; i++)
                uvw[i] = (KP[4 * i] * X[0] + KP[4 * i + 1] * X[1]) + (KP[4 * i + 2] * X[2] + KP[4 * i + 3] * 1.0f);
        }
        float u, v;
        if (uvw[2] < 0.0f) { u = -1.0f; v = -1.0f; }
        else { u = uvw[0] / uvw[2]; v = uvw[1] / uvw[2]; }
        if (!(u >= 0.0f && u < (float)f.width && v >= 0.0f && v < (float)f.height)) break;   // :58

        float center[3];                                                 // Frame::camera_center (src/Frame.cpp:39-42)
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const float a[3] = {-T[0 * 4 + i], -T[1 * 4 + i], -T[2 * 4 + i]};
            const float t[3] = {T[3], T[7], T[11]};
            center[i] = dot3f(a, t);
        }
        const float ray[3] = {X[0] - center[0], X[1] - center[1], X[2] - center[2]};
        float normal[3] = {0.0f, 0.0f, 0.0f};
        float nearest = 3.402823466e+38f, furthest = 0.0f;
        for (int ob = 0; ob < nobs; ob += K2G) {                          // src/MapPoint.cpp:24-45, eight observations per pass
            float Cx[3] = {ob == 0 ? C0[0] : C1[0], ob == 0 ? C0[1] : C1[1], ob == 0 ? C0[2] : C1[2]};
            if (ob > K2G && ob + sub < nobs) {
                const float* C = m.kf_centers + 3 * (size_t)m.obs_kf[o0 + ob + sub];
                Cx[0] = C[0]; Cx[1] = C[1]; Cx[2] = C[2];
            }
            float d[3] = {X[0] - Cx[0], X[1] - Cx[1], X[2] - Cx[2]};
            const float dist = sqrtf(dot3f(d, d));
            normalize3f(d);
            normalize3f(d);
#pragma unroll
            for (int j = 0; j < K2G; j++) {                               // summed in observation order by every lane
                const float dj0 = __shfl(d[0], j, K2G), dj1 = __shfl(d[1], j, K2G), dj2 = __shfl(d[2], j, K2G);
                const float distj = __shfl(dist, j, K2G);
                if (ob + j < nobs) {
                    nearest = distj < nearest ? distj : nearest;
                    furthest = furthest < distj ? distj : furthest;
                    normal[0] += dj0; normal[1] += dj1; normal[2] += dj2;
                }
            }
        }
        normalize3f(normal);
        float rn[3] = {ray[0], ray[1], ray[2]};
        normalize3f(rn);
        if (dot3f(normal, rn) < 0.5f) break;                             // :62-66
        const float distance = sqrtf(dot3f(ray, ray));
        if (distance < nearest / 2.0f || distance > furthest * 1.25f) break;   // :69-73

        K2_STAMP(3);
        // KDTree2D::radius_search (src/MapMatcher.cpp:75, src/KDTree.cpp:45-82) by lane 0 of the group
        const float r2 = 20.0f * 20.0f;
        int nc = 0, over_kp = 0, over_d = max_distance;
        if (sub == 0) {
            int sp = 0;
            int cur = f.kd_root, odd = 0;
            K2Node nd;
            if (cur >= 0) nd = tree[cur];
            while (cur >= 0) {
                int l = (int)(nd.lr & 0xFFFFu), r = (int)(nd.lr >> 16);
                l = l == 0xFFFF ? -1 : l; r = r == 0xFFFF ? -1 : r;
                const K2Node ndl = tree[l >= 0 ? l : 0], ndr = tree[r >= 0 ? r : 0];
                const float dx = nd.x - u, dy = nd.y - v;
                const float d2 = dx * dx + dy * dy;
                if (d2 <= r2 && nd.kp >= 0) {                               // in range and open (:65, :81)
                    const int kp = nd.kp;
                    if (nc < K2_MAXC) queue[nc++][g] = kp;
                    else {                                                  // crowded neighbourhoods: on the spot, after every queued one
                        const uint4 a0 = f.desc[2 * (size_t)kp], a1 = f.desc[2 * (size_t)kp + 1];
                        int d = 0x7fffffff;
                        for (int o = o0; o < o1; o++) {
                            const size_t row = (size_t)m.obs_desc[o];
                            d = min(d, hamming256(a0, a1, m.pool[2 * row], m.pool[2 * row + 1]));
                        }
                        if (d < over_d) { over_d = d; over_kp = kp; }
                    }
                }
                const float delta = odd ? dy : dx;
                const bool left_near = delta > 0;
                const int near_child = left_near ? l : r;
                const int far_child = left_near ? r : l;
                odd ^= 1;
                if (delta * delta <= r2 && far_child >= 0 && sp < K2_STACK) stack[sp++][g] = far_child | (odd << 30);
                if (near_child >= 0) { cur = near_child; nd = left_near ? ndl : ndr; }
                else if (sp > 0) { const int e = stack[--sp][g]; cur = e & 0x3FFFFFFF; odd = (e >> 30) & 1; nd = tree[cur]; }
                else cur = -1;
            }
        }
        K2_STAMP(4);
        nc = __shfl(nc, 0, K2G);
        over_d = __shfl(over_d, 0, K2G);
        over_kp = __shfl(over_kp, 0, K2G);
        // the queued candidates (visiting order), four per round trip; each lane compares its observation's row
        int best_kp = 0, best_d = max_distance;
        if (nc > 0 && nobs > 0) {
            for (int ob = 0; ob < nobs; ob += K2G) {
                uint4 r0 = ob == 0 ? b0 : b2, r1 = ob == 0 ? b1 : b3;
                const bool mine = ob + sub < nobs;
                if (ob > K2G && mine) {
                    const size_t row = (size_t)m.obs_desc[o0 + ob + sub];
                    r0 = m.pool[2 * row]; r1 = m.pool[2 * row + 1];
                }
                for (int c = 0; c < nc; c += 4) {
                    int kq[4];
                    uint4 a0[4], a1[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        kq[q] = queue[min(c + q, nc - 1)][g];
                        a0[q] = f.desc[2 * (size_t)kq[q]]; a1[q] = f.desc[2 * (size_t)kq[q] + 1];
                    }
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int d = group8_min(mine ? hamming256(a0[q], a1[q], r0, r1) : 0x7fffffff);
                        if (c + q < nc) {
                            if (ob == 0) { if (d < best_d) { best_d = d; best_kp = kq[q]; } }   // :88-91, visiting order
                            if (nobs > K2G) qmin[c + q][g] = ob == 0 ? d : min(d, qmin[c + q][g]);
                        }
                    }
                }
            }
            if (nobs > K2G) {                                             // more than eight observations: minima over all passes
                best_kp = 0; best_d = max_distance;
                for (int c = 0; c < nc; c++) {
                    const int d = qmin[c][g];
                    if (d < best_d) { best_d = d; best_kp = queue[c][g]; }
                }
            }
        }
        K2_STAMP(5);
        if (over_d < best_d) { best_d = over_d; best_kp = over_kp; }
        if (best_d < max_distance) {
            out_kp = best_kp;
            out_d = best_d;
            // :95-97 sequential strict-'<' over map order == atomicMin of (dist, map order)
            if (sub == 0) atomicMin(&prop[best_kp], ((unsigned long long)(unsigned)best_d << 32) | (unsigned)(m.point_base + p));
        }
    } while (0);
    if (have && sub == 0) {
        point_kp[p] = out_kp;
        point_dist[p] = out_d;
    }
    K2_STAMP(6);
}

// K3: one workgroup decodes the proposal table, compacts the accepted matches in order and resets the table
// (tried as a tail of K2's last workgroup: no cheaper than this launch)
__global__ __launch_bounds__(1024) void k3_accept(unsigned long long* __restrict__ prop, int n,
                                                  int max_distance, int32_t* __restrict__ prop_point,
                                                  int32_t* __restrict__ prop_dist, int32_t* __restrict__ match_kp,
                                                  int32_t* __restrict__ match_point, int32_t* __restrict__ match_count)
{
    k3_accept_body(prop, n, max_distance, prop_point, prop_dist, match_kp, match_point, match_count);
}

// the map sharded over ranks (point_base = map order of this shard's first point).  Every rank runs
// K2 on its points; the per-keypoint proposal table — packed (distance << 32 | GLOBAL map order), built with atomicMin —
// is then MIN-all-reduced over the communicator (one ncclAllReduce(min, u64) of 8 N bytes; the in-process group: an
// on-device minimum), so that every rank's K3 accepts the same winners: exactly the unsharded result, ties included
// (src/MapMatcher.cpp:95-97: first in map order wins).  d_prop_point / d_match_point hold GLOBAL map indices;
// d_point_kp / d_point_dist are this shard's.
static int reproj_match_impl(rs_context* ctx, const rs_frame_view* fr, const rs_map_view* mp, int point_base, bool reduce, int replace,
                             int max_distance, int32_t* d_point_kp, int32_t* d_point_dist,
                             int32_t* d_prop_point, int32_t* d_prop_dist, int32_t* d_match_kp,
                             int32_t* d_match_point, int32_t* d_match_count)
{
    if (!ctx || !fr || !mp) return RS_ERR_INVALID;
    const int N = fr->n_keypoints, P = mp->n_points;
    if (N < 0 || P < 0 || point_base < 0) return rs_fail(ctx, RS_ERR_INVALID, "negative size");
    if (!d_match_count) return rs_fail(ctx, RS_ERR_INVALID, "null match_count");
    if (N >= (1 << 30)) return rs_fail(ctx, RS_ERR_UNSUPPORTED, "too many keypoints");
    RS_HIP(ctx, hipSetDevice(ctx->device));
    if (N == 0) {
        RS_HIP(ctx, hipMemsetAsync(d_match_count, 0, sizeof(int32_t), ctx->stream));
        if (P > 0 && d_point_kp) RS_HIP(ctx, hipMemsetAsync(d_point_kp, 0xFF, sizeof(int32_t) * (size_t)P, ctx->stream));
        return RS_OK;
    }
    if (!d_prop_point || !d_prop_dist || !d_match_kp || !d_match_point)
        return rs_fail(ctx, RS_ERR_INVALID, "null output");
    if (P > 0 && (!d_point_kp || !d_point_dist || !mp->d_positions || !mp->d_eligible || !mp->d_obs_ptr))
        return rs_fail(ctx, RS_ERR_INVALID, "null map pointer");
    if ((size_t)N > ctx->prop_cap) {     // grow-only; all-ones once, every call leaves it all-ones again
        RS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->prop) RS_HIP(ctx, hipFree(ctx->prop));
        ctx->prop = nullptr; ctx->prop_cap = 0;
        const size_t cap = ((size_t)N + 4095) & ~(size_t)4095;
        if (hipMalloc(&ctx->prop, sizeof(unsigned long long) * cap) != hipSuccess)
            return rs_fail(ctx, RS_ERR_NOMEM, "proposal table of %zu entries", cap);
        RS_HIP(ctx, hipMemsetAsync(ctx->prop, 0xFF, sizeof(unsigned long long) * cap, ctx->stream));
        ctx->prop_cap = cap;
    }
    unsigned long long* prop = ctx->prop;
    if (P > 0) {
        K2Frame f;
        memcpy(f.T, fr->pose, sizeof f.T);
        f.fx = fr->fx; f.fy = fr->fy; f.cx = fr->cx; f.cy = fr->cy;
        f.width = fr->width; f.height = fr->height; f.n_keypoints = N; f.kd_root = fr->kd_root;
        f.kp = (const float2*)fr->d_keypoints; f.desc = (const uint4*)fr->d_descriptors;
        f.kp_matched = fr->d_kp_matched; f.kd_node_kp = fr->d_kd_node_kp; f.kd_left = fr->d_kd_left;
        f.kd_right = fr->d_kd_right;
        f.packed = (const float4*)fr->d_kd_packed;
        if (((uintptr_t)f.packed) & 15) return rs_fail(ctx, RS_ERR_INVALID, "d_kd_packed must be 16-byte aligned");
        K2Map m;
        m.n_points = P; m.point_base = point_base; m.pos = mp->d_positions; m.eligible = mp->d_eligible; m.obs_ptr = mp->d_obs_ptr;
        m.obs_kf = mp->d_obs_kf; m.obs_desc = mp->d_obs_desc; m.kf_centers = mp->d_kf_centers;
        m.pool = (const uint4*)mp->d_desc_pool;
        const int tree_in_lds = N <= K2_MAX_LDS_NODES ? 1 : 0;
        rs_prof_scope ps(ctx, "K2_reproj_match");
        if (tree_in_lds && ctx->k2_mode == 0) {                           // eight lanes per map point
            const int th = 512, pts = th / K2G;                           // 64 map points per workgroup (256 / 1024 threads: slower)
            const size_t lds = sizeof(int) * (K2_STACK + 2 * K2_MAXC) * pts + sizeof(K2Node) * (size_t)N;
            auto kern = k2_reproj_match_grouped<512>;
            if (lds > 48 * 1024)
                RS_HIP(ctx, rs_lds_attr((const void*)kern, lds));
            hipLaunchKernelGGL(kern, dim3((P + pts - 1) / pts), dim3(th), lds,
                               ctx->stream, f, m, replace, max_distance, d_point_kp, d_point_dist, prop);
        } else {
            const size_t lds = sizeof(int) * (K2_STACK + 2 * K2_MAXC) * K2_THREADS + (tree_in_lds ? sizeof(K2Node) * (size_t)N : 0);
            if (lds > 48 * 1024)
                RS_HIP(ctx, rs_lds_attr((const void*)k2_reproj_match, lds));
            hipLaunchKernelGGL(k2_reproj_match, dim3((P + K2_THREADS - 1) / K2_THREADS), dim3(K2_THREADS), lds,
                               ctx->stream, f, m, replace, max_distance, tree_in_lds, d_point_kp, d_point_dist, prop);
        }
    }
    if (reduce && rs_comm_active(ctx)) {
        rs_prof_scope ps(ctx, "C3_allreduce_min_proposals");
        const int rc = rs_allreduce_min_u64(ctx, prop, (size_t)N);
        if (rc) return rc;
    }
    {
        rs_prof_scope ps(ctx, "K3_accept");
        hipLaunchKernelGGL(k3_accept, dim3(1), dim3(1024), 0, ctx->stream, prop, N, max_distance, d_prop_point,
                           d_prop_dist, d_match_kp, d_match_point, d_match_count);
    }
    RS_HIP(ctx, hipGetLastError());
    return RS_OK;
}

extern "C" int rs_reproj_match(rs_context* ctx, const rs_frame_view* fr, const rs_map_view* mp, int replace,
                               int max_distance, int32_t* d_point_kp, int32_t* d_point_dist,
                               int32_t* d_prop_point, int32_t* d_prop_dist, int32_t* d_match_kp,
                               int32_t* d_match_point, int32_t* d_match_count)
{
    return reproj_match_impl(ctx, fr, mp, 0, false, replace, max_distance, d_point_kp, d_point_dist, d_prop_point, d_prop_dist, d_match_kp,
                             d_match_point, d_match_count);
}

extern "C" int rs_reproj_match_sharded(rs_context* ctx, const rs_frame_view* fr, const rs_map_view* mp_shard, int point_base, int replace,
                                       int max_distance, int32_t* d_point_kp, int32_t* d_point_dist,
                                       int32_t* d_prop_point, int32_t* d_prop_dist, int32_t* d_match_kp,
                                       int32_t* d_match_point, int32_t* d_match_count)
{
    return reproj_match_impl(ctx, fr, mp_shard, point_base, true, replace, max_distance, d_point_kp, d_point_dist, d_prop_point, d_prop_dist,
                             d_match_kp, d_match_point, d_match_count);
}

// rs_kdtree_pack: the frame's KD-tree as K2 wants it in LDS — {x, y, left, right} per node, then the node's keypoint
// index — so that the workgroups of the (two) match calls of a frame copy it instead of gathering it.
__global__ __launch_bounds__(256) void k2_pack_tree(int n, const float2* __restrict__ kp, const int32_t* __restrict__ node_kp,
                                                    const int32_t* __restrict__ left, const int32_t* __restrict__ right,
                                                    float4* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int kpi = node_kp[i];
    const float2 q = kp[kpi];
    out[i] = make_float4(q.x, q.y, __int_as_float(left[i]), __int_as_float(right[i]));
    ((int*)(out + n))[i] = kpi;
}

extern "C" int rs_kdtree_pack(rs_context* ctx, const rs_frame_view* fr, void* d_packed)
{
    if (!ctx || !fr) return RS_ERR_INVALID;
    const int n = fr->n_keypoints;
    if (n < 0) return rs_fail(ctx, RS_ERR_INVALID, "negative size");
    if (n == 0) return RS_OK;
    if (!d_packed || !fr->d_keypoints || !fr->d_kd_node_kp || !fr->d_kd_left || !fr->d_kd_right) return rs_fail(ctx, RS_ERR_INVALID, "null pointer");
    if (((uintptr_t)d_packed) & 15) return rs_fail(ctx, RS_ERR_INVALID, "d_packed must be 16-byte aligned");
    RS_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k2_pack_tree, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, (const float2*)fr->d_keypoints,
                       fr->d_kd_node_kp, fr->d_kd_left, fr->d_kd_right, (float4*)d_packed);
    RS_HIP(ctx, hipGetLastError());
    return RS_OK;
}
