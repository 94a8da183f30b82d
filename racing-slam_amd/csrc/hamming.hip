// hamming.hip — K1: brute-force 2-NN under 256-bit Hamming distance, and the
// match_descriptors filters + ordered compaction.
//
// Replaces cv::BFMatcher(NORM_HAMMING).knnMatch(k=2) + the Lowe / max-distance
// filters of MapMatcher::match_descriptors (reference src/MapMatcher.cpp:129-163).
//
// Mapping to gfx950: one lane owns one query descriptor (8 VGPRs); train rows
// are wave-uniform, so they stream through the SCALAR cache (s_load_dwordx8)
// and the inner loop is pure VALU: 8 v_xor + 8 v_bcnt_u32_b32 (accumulating
// form) + key pack + a 2-deep sorted insert.  The pair space is cut into
// (query block of 64) x (train split) so that ~2k waves cover the 1024 SIMDs;
// the 4 waves of a workgroup take 4 consecutive train sub-ranges and merge
// through LDS.  The packed key (dist << 20 | train index) orders candidates by
// distance first and index second, which is exactly OpenCV's tie rule.
#include "common.h"

#define K1_WAVES 4
#define K1_KEY_NONE 0xFFFFFFFFu
#define K1_IDX_BITS 20
#define K1_IDX_MASK ((1u << K1_IDX_BITS) - 1u)

__device__ __forceinline__ void top2_insert(uint32_t& k0, uint32_t& k1, uint32_t key)
{
    k1 = min(k1, max(k0, key));
    k0 = min(k0, key);
}

__global__ __launch_bounds__(64 * K1_WAVES) void k1_hamming_knn2(
    const uint4* __restrict__ query, int nq, const uint4* __restrict__ train, int nt,
    int rows_per_wave, int nsplit, int nqb, int xcd_groups, uint2* __restrict__ top)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // 1-D grid -> (query block, train split, pair).  Blocks are dealt round-robin over the 8 XCDs (blocks id and id + 8
    // share one, each XCD with its own L2): with xcd_groups set (batch % 8 == 0) all blocks of a pair carry the same
    // id % 8, so a pair's descriptor rows are fetched into ONE L2 instead of all eight (speed only, any placement is
    // correct).  Otherwise the plain order.
    const int per_pair = nqb * nsplit;
    int b, rem;
    if (xcd_groups) {
        const int j = (int)blockIdx.x >> 3;
        b = (j / per_pair) * 8 + ((int)blockIdx.x & 7);
        rem = j % per_pair;
    } else {
        b = (int)blockIdx.x / per_pair;
        rem = (int)blockIdx.x % per_pair;
    }
    const int split = rem / nqb, qblock = rem % nqb;
    const int qi = qblock * 64 + lane;
    const uint4* q = query + (size_t)b * nq * 2;
    const uint4* t = train + (size_t)b * nt * 2;

    uint4 qa = make_uint4(0, 0, 0, 0), qb = make_uint4(0, 0, 0, 0);
    if (qi < nq) { qa = q[2 * qi]; qb = q[2 * qi + 1]; }

    const int t0 = (split * K1_WAVES + wave) * rows_per_wave;
    const int t1 = min(t0 + rows_per_wave, nt);
    uint32_t k0 = K1_KEY_NONE, k1 = K1_KEY_NONE;
#pragma unroll 4
    for (int j = t0; j < t1; ++j) {
        const uint4 ta = t[2 * j], tb = t[2 * j + 1];   // wave-uniform address -> scalar loads
        uint32_t d = __builtin_popcount(qa.x ^ ta.x);
        d += __builtin_popcount(qa.y ^ ta.y);
        d += __builtin_popcount(qa.z ^ ta.z);
        d += __builtin_popcount(qa.w ^ ta.w);
        d += __builtin_popcount(qb.x ^ tb.x);
        d += __builtin_popcount(qb.y ^ tb.y);
        d += __builtin_popcount(qb.z ^ tb.z);
        d += __builtin_popcount(qb.w ^ tb.w);
        top2_insert(k0, k1, (d << K1_IDX_BITS) | (uint32_t)j);
    }

    __shared__ uint32_t sk[K1_WAVES][2][64];
    sk[wave][0][lane] = k0;
    sk[wave][1][lane] = k1;
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int w = 1; w < K1_WAVES; ++w) {
            top2_insert(k0, k1, sk[w][0][lane]);
            top2_insert(k0, k1, sk[w][1][lane]);
        }
        // Merge into the per-query table {best, second} (all-ones before the launch).  Keys are unique (the train index
        // sits in the low bits), so the table's top-2 of the union of all splits' top-2 is: best = min over the k0s;
        // second = min over every k1 and every LOSER of a best-update — whichever of (old, new) a min displaced or
        // failed to displace is a union element other than the final best, and the true second is among them.
        if (qi < nq) {
            uint32_t* t = (uint32_t*)&top[(size_t)b * nq + qi];
            if (nsplit == 1) { t[0] = k0; t[1] = k1; }
            else {
                if (k0 != K1_KEY_NONE) {
                    const uint32_t old = atomicMin(&t[0], k0);
                    const uint32_t loser = max(old, k0);
                    if (loser != K1_KEY_NONE) atomicMin(&t[1], loser);
                }
                if (k1 != K1_KEY_NONE) atomicMin(&t[1], k1);
            }
        }
    }
}

// One workgroup per batch item: read the merged {best, second} keys (and put the table back to all-ones), decode, apply
// the filters of src/MapMatcher.cpp:150-161 and emit the accepted matches in ascending query order.  Every thread owns a
// CONTIGUOUS chunk of queries (<= K1B_CH per pass); the ordered compaction is one workgroup scan per pass: one pass for
// up to 4096 queries (2000 at the metric's size: two queries per thread).
#define K1B_CH 4
__global__ __launch_bounds__(1024) void k1_merge_filter(
    uint2* __restrict__ top, int nq, int nt, int max_distance, int do_filter,
    int32_t* __restrict__ idx0, int32_t* __restrict__ dist0, int32_t* __restrict__ idx1,
    int32_t* __restrict__ dist1, int32_t* __restrict__ match_query, int32_t* __restrict__ match_train,
    int32_t* __restrict__ match_count)
{
    const int b = blockIdx.x;
    int running = 0;                                         // matches emitted by earlier passes (same in every thread)
    for (int base = 0; base < nq; base += 1024 * K1B_CH) {
        const int span = min(nq - base, 1024 * K1B_CH);
        const int ch = (span + 1023) / 1024;                 // queries per thread in this pass (1 .. K1B_CH)
        const int q0 = base + (int)threadIdx.x * ch;
        uint32_t k0[K1B_CH], k1[K1B_CH];
#pragma unroll
        for (int u = 0; u < K1B_CH; u++) {
            const int qi = min(q0 + u, nq - 1);              // clamped: no branches around the loads
            const uint2 v = top[(size_t)b * nq + qi];
            k0[u] = v.x; k1[u] = v.y;
        }
#pragma unroll
        for (int u = 0; u < K1B_CH; u++)
            if (u < ch && q0 + u < base + span) top[(size_t)b * nq + q0 + u] = make_uint2(K1_KEY_NONE, K1_KEY_NONE);
        int cnt = 0;
        bool ok[K1B_CH];
#pragma unroll
        for (int u = 0; u < K1B_CH; u++) {
            const int qi = q0 + u;
            const bool live = u < ch && qi < base + span;
            const int i0 = (int)(k0[u] & K1_IDX_MASK), d0 = (int)(k0[u] >> K1_IDX_BITS);
            const bool has1 = k1[u] != K1_KEY_NONE;
            const int i1 = has1 ? (int)(k1[u] & K1_IDX_MASK) : -1, d1 = has1 ? (int)(k1[u] >> K1_IDX_BITS) : -1;
            if (live) {
                const size_t o = (size_t)b * nq + qi;
                if (idx0) idx0[o] = i0;
                if (dist0) dist0[o] = d0;
                if (idx1) idx1[o] = i1;
                if (dist1) dist1[o] = d1;
            }
            ok[u] = do_filter && live && d0 <= max_distance;                      // :152
            if (ok[u] && nt >= 2 && 4 * d0 > 3 * d1) ok[u] = false;               // :156, 0.75 = 3/4 exactly
            cnt += ok[u] ? 1 : 0;
        }
        if (!do_filter) continue;
        int total;
        int off = running + rs_block_exclusive_scan(cnt, &total);
#pragma unroll
        for (int u = 0; u < K1B_CH; u++)
            if (ok[u]) {
                match_query[(size_t)b * nq + off] = q0 + u;
                match_train[(size_t)b * nq + off] = (int)(k0[u] & K1_IDX_MASK);
                off++;
            }
        running += total;
    }
    if (do_filter && threadIdx.x == 0) match_count[b] = running;
}

static int knn2_launch(rs_context* ctx, const uint8_t* d_query, int nq, const uint8_t* d_train, int nt,
                       int batch, int max_distance, int do_filter, int32_t* mq, int32_t* mt, int32_t* mc,
                       int32_t* i0, int32_t* d0, int32_t* i1, int32_t* d1)
{
    if (!ctx) return RS_ERR_INVALID;
    if (nq < 0 || nt < 0 || batch < 0) return rs_fail(ctx, RS_ERR_INVALID, "negative size");
    if (nt >= (1 << K1_IDX_BITS)) return rs_fail(ctx, RS_ERR_UNSUPPORTED, "nt must be < 2^20");
    if (batch == 0) return RS_OK;
    if (nq == 0 || nt == 0) {   // empty guard, src/MapMatcher.cpp:139-141
        if (do_filter && mc) RS_HIP(ctx, hipMemsetAsync(mc, 0, sizeof(int32_t) * (size_t)batch, ctx->stream));
        return RS_OK;
    }
    if (!d_query || !d_train) return rs_fail(ctx, RS_ERR_INVALID, "null descriptor pointer");
    if (((uintptr_t)d_query | (uintptr_t)d_train) & 15) return rs_fail(ctx, RS_ERR_INVALID, "descriptors must be 16-byte aligned");
    if (do_filter && (!mq || !mt || !mc)) return rs_fail(ctx, RS_ERR_INVALID, "null match output");
    RS_HIP(ctx, hipSetDevice(ctx->device));

    const int nqb = (nq + 63) / 64;
    // aim at ~2048 waves in flight (2 per SIMD), at least 8 train rows per wave
    int nsplit = (2048 + K1_WAVES * nqb * batch - 1) / (K1_WAVES * nqb * batch);
    const int max_split = (nt + 8 * K1_WAVES - 1) / (8 * K1_WAVES);
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit < 1) nsplit = 1;
    const int rows_per_wave = (nt + nsplit * K1_WAVES - 1) / (nsplit * K1_WAVES);
    if ((size_t)nqb * nsplit * batch > 0x7fffffffu) return rs_fail(ctx, RS_ERR_UNSUPPORTED, "too many query blocks for one launch");
    const size_t need = (size_t)batch * nq;
    if (need > ctx->k1_top_cap) {        // grow-only; all-ones once, every call leaves it all-ones again
        RS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->k1_top) RS_HIP(ctx, hipFree(ctx->k1_top));
        ctx->k1_top = nullptr; ctx->k1_top_cap = 0;
        const size_t cap = (need + 4095) & ~(size_t)4095;
        if (hipMalloc(&ctx->k1_top, sizeof(uint2) * cap) != hipSuccess) return rs_fail(ctx, RS_ERR_NOMEM, "top-2 table of %zu entries", cap);
        RS_HIP(ctx, hipMemsetAsync(ctx->k1_top, 0xFF, sizeof(uint2) * cap, ctx->stream));
        ctx->k1_top_cap = cap;
    }
    void* ws = ctx->k1_top;
    {
        rs_prof_scope ps(ctx, "K1_hamming_knn2");
        hipLaunchKernelGGL(k1_hamming_knn2, dim3((unsigned)((size_t)nqb * nsplit * batch)), dim3(64 * K1_WAVES), 0, ctx->stream,
                           (const uint4*)d_query, nq, (const uint4*)d_train, nt, rows_per_wave, nsplit, nqb,
                           (batch % 8 == 0) ? 1 : 0, (uint2*)ws);
    }
    {
        rs_prof_scope ps(ctx, "K1b_merge_filter");
        hipLaunchKernelGGL(k1_merge_filter, dim3(batch), dim3(1024), 0, ctx->stream, (uint2*)ws, nq, nt,
                           max_distance, do_filter, i0, d0, i1, d1, mq, mt, mc);
    }
    RS_HIP(ctx, hipGetLastError());
    return RS_OK;
}

extern "C" int rs_hamming_knn2(rs_context* ctx, const uint8_t* d_query, int nq, const uint8_t* d_train, int nt,
                               int batch, int32_t* d_idx0, int32_t* d_dist0, int32_t* d_idx1, int32_t* d_dist1)
{
    return knn2_launch(ctx, d_query, nq, d_train, nt, batch, 0, 0, nullptr, nullptr, nullptr, d_idx0, d_dist0,
                       d_idx1, d_dist1);
}

extern "C" int rs_match_descriptors(rs_context* ctx, const uint8_t* d_query, int nq, const uint8_t* d_train,
                                    int nt, int batch, int max_distance, int32_t* d_match_query,
                                    int32_t* d_match_train, int32_t* d_match_count, int32_t* d_idx0,
                                    int32_t* d_dist0, int32_t* d_idx1, int32_t* d_dist1)
{
    return knn2_launch(ctx, d_query, nq, d_train, nt, batch, max_distance, 1, d_match_query, d_match_train,
                       d_match_count, d_idx0, d_dist0, d_idx1, d_dist1);
}
