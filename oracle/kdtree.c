/*
 * oracle/kdtree.c — CPU restatement of KDTree2D (reference src/KDTree.cpp:8-82).
 * TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see rs_oracle.h).
 *
 * Build (src/KDTree.cpp:19-43): axis = depth % 2, mid = (start+end)/2,
 * std::nth_element on the coordinate, node = element at mid, children built
 * from [start,mid) and [mid+1,end).  nth_element leaves the arrangement of
 * equal coordinates unspecified; here ties are ordered by keypoint index so
 * the tree is a pure function of the input.  Without coordinate ties the tree
 * equals the reference's for any nth_element implementation (the element at
 * mid and the two sets are determined by the values).
 *
 * The flattened form: node id = its position `mid` in the index permutation.
 */
#define _GNU_SOURCE
#include <stdlib.h>

#include "rs_oracle.h"

typedef struct {
    const float* pts;
    int axis;
} kd_cmp_ctx;

static int kd_cmp(const void* a, const void* b, void* c)
{
    const kd_cmp_ctx* ctx = (const kd_cmp_ctx*)c;
    int ia = *(const int32_t*)a, ib = *(const int32_t*)b;
    float va = ctx->pts[2 * ia + ctx->axis], vb = ctx->pts[2 * ib + ctx->axis];
    if (va < vb) return -1;
    if (va > vb) return 1;
    return (ia > ib) - (ia < ib);
}

static int kd_build(const float* pts, int32_t* perm, int depth, int start, int end,
                    int32_t* node_left, int32_t* node_right)
{
    if (start >= end) return -1;
    kd_cmp_ctx ctx = {pts, depth % 2};
    int mid = (start + end) / 2;
    qsort_r(perm + start, (size_t)(end - start), sizeof(int32_t), kd_cmp, &ctx);
    node_left[mid] = kd_build(pts, perm, depth + 1, start, mid, node_left, node_right);
    node_right[mid] = kd_build(pts, perm, depth + 1, mid + 1, end, node_left, node_right);
    return mid;
}

int orc_kdtree_build(const float* keypoints, int n, int32_t* node_kp, int32_t* node_left,
                     int32_t* node_right, int32_t* root)
{
    if (n < 0) return 1;
    for (int i = 0; i < n; i++) node_kp[i] = i;
    *root = kd_build(keypoints, node_kp, 0, 0, n, node_left, node_right);
    return 0;
}

typedef struct {
    const float* pts;
    const int32_t* node_kp;
    const int32_t* left;
    const int32_t* right;
    float x, y, r2;
    int32_t* out;
    int cap, count;
} kd_search;

/* src/KDTree.cpp:52-82 */
static void kd_radius(kd_search* s, int node, int depth)
{
    if (node < 0) return;
    int kp = s->node_kp[node];
    float dx = s->pts[2 * kp] - s->x;
    float dy = s->pts[2 * kp + 1] - s->y;
    float d2 = dx * dx + dy * dy;
    if (d2 <= s->r2) {                       /* inclusive, :65 */
        if (s->count < s->cap) s->out[s->count] = kp;
        s->count++;
    }
    float delta = (depth % 2 == 0) ? dx : dy;
    int near_child = (delta > 0) ? s->left[node] : s->right[node];   /* :73-74 */
    int far_child = (delta > 0) ? s->right[node] : s->left[node];
    kd_radius(s, near_child, depth + 1);
    if (delta * delta <= s->r2) {            /* :79 */
        kd_radius(s, far_child, depth + 1);
    }
}

int orc_kdtree_radius(const float* keypoints, const int32_t* node_kp, const int32_t* node_left,
                      const int32_t* node_right, int root, float x, float y, float radius,
                      int32_t* out, int cap)
{
    kd_search s = {keypoints, node_kp, node_left, node_right, x, y, radius * radius, out, cap, 0};
    kd_radius(&s, root, 0);
    return s.count;
}
