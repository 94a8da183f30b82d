// ba_schur.hip — K5: linearisation + point-block Schur complement on the f64
// matrix cores (v_mfma_f64_16x16x4_f64), and the one-off landmark grouping
// that makes it possible.
//
// Replaces, per LM iteration, the residual/Jacobian evaluation
// (ReprojectionError, reference src/Optimization.cpp:21-72), the Huber
// corrector and Ceres' SchurEliminator inside ceres::Solve (:360).
//
// Idea.  S = U + Lambda - sum_p W_p V_p^-1 W_p^T is a sum of rank-3 updates.
// With V_p = L_p L_p^T and Y_p = W_p L_p^-T (6k x 3 for a landmark seen by k free
// cameras) the sum is a SYRK:  S -= Y Y^T  with Y = [Y_1 Y_2 ...].  Landmarks are
// sorted once per solve by (first, last) free-camera slot, so 16 consecutive
// landmarks ("an item") touch a small union of cameras (<= 10 for a local
// window).  One WAVE owns one item: it writes the 16 Y_p into a COMPACT row
// space (6 rows per camera of the union, + 1 row carrying L_p^-1 g_p) in LDS,
// then runs a dense 64 x 48 (x 64) SYRK on the matrix cores — K = 48 columns,
// no padding along K — and scatter-adds the small dense result into the global
// S / rhs once per item.  The extra row makes the reduced right-hand side
// sum_p W_p V_p^-1 g_p fall out of the same MFMAs.  Jacobians live only in
// registers; nothing per-observation is written to HBM.
//
// Item classes: union <= 10 cameras -> 4x4 tiles (NT = 4, 16 landmarks per
// SYRK); <= 21 cameras -> 8x8 tiles (NT = 8, two half-items of 8 landmarks to
// stay inside the same 30 KB LDS tile); larger unions fall back to per-landmark
// f64 atomics (correct for any covisibility, slow).
#include "ba_common.h"
#include "ba_init_body.h"
#include "ba_schur_body.h"

// ------------------------------------------------------------ setup kernels
// Counting sort of the landmarks by (first, last) free-camera slot.  Histogram and cursors are
// kept in GRP_REP replicas (workgroup w uses replica w % GRP_REP) and the histogram is first
// accumulated in LDS: device-scope atomics on one 64-B line serialise (~12 ns each), and the
// ~170 live buckets of a local window share a dozen lines.
#define GRP_REP 8
#define GRP_LDS_BINS 4096

// 1024 threads, 128 landmarks per workgroup, EIGHT lanes per landmark: the observations of a landmark go out together (one
// lane each, then the next eight) instead of as a chain of dependent loads in one thread (obs_cam -> slot, six deep: the
// kernel was 9 us for a 10 k-landmark window and 240 us for a batch of 32 of them), and the lanes combine mask, first and last
// slot with three xor-shuffles.  The histogram replica of a landmark is the one the scatter kernel will draw its position
// from: (landmark / 256) mod GRP_REP.
#define GRP_COUNT_LM 128
// from_mask: the reduced-system slot of a camera comes from the free-camera mask (windows of at most 64 cameras) instead of
// b.slot — in ba_init_count the table is being written by the same launch.
static __device__ __forceinline__ void ba_group_count_body(const BaDims& d, const BaBufs& b, const BaGroup& g,
                                                           const unsigned long long free_mask = 0ull, const int from_mask = 0)
{
    __shared__ int lh[GRP_LDS_BINS];
    __shared__ int s_span;                 // the workgroup's largest span: ONE atomic on the window's word per workgroup (atomic
                                           // instructions on one address serialise at ~10 ns each whatever their lane count)
    const int nb = g.n_buckets + 1;
    const bool use_lds = nb <= GRP_LDS_BINS;
    if (use_lds) for (int i = threadIdx.x; i < nb; i += blockDim.x) lh[i] = 0;
    if (threadIdx.x == 0) s_span = 0;
    __syncthreads();
    const int p = blockIdx.x * GRP_COUNT_LM + (int)(threadIdx.x >> 3), sub = threadIdx.x & 7;
    int* hist = g.hist + (size_t)((blockIdx.x * GRP_COUNT_LM >> 8) & (GRP_REP - 1)) * nb;
    {
        uint64_t m0 = 0, m1 = 0;
        int first = 1 << 30, last = -1;
        if (p < d.P) {
            const int o1 = b.obs_ptr[p + 1];
            for (int o = b.obs_ptr[p] + sub; o < o1; o += 8) {
                const int c = b.obs_cam[o];
                const int s = from_mask ? ((free_mask >> c & 1ull) ? __popcll(free_mask & ((1ull << c) - 1ull)) : -1) : b.slot[c];
                g.obs_cs[o] = c | ((s + 1) << 16);
                if (s < 0) continue;
                if (s < 64) m0 |= 1ull << s; else m1 |= 1ull << (s - 64);
                first = min(first, s);
                last = max(last, s);
            }
        }
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) {
            m0 |= __shfl_xor(m0, off, 64);
            m1 |= __shfl_xor(m1, off, 64);
            first = min(first, __shfl_xor(first, off, 64));
            last = max(last, __shfl_xor(last, off, 64));
        }
        if (p < d.P && sub == 0) {
            g.mask[2 * (size_t)p] = m0;
            g.mask[2 * (size_t)p + 1] = m1;
            const int bk = last < 0 ? d.Cf * d.Cf : first * d.Cf + last;
            g.bucket[p] = bk;
            if (last > first) atomicMax(&s_span, last - first);
            if (use_lds) atomicAdd(&lh[bk], 1); else atomicAdd(&hist[bk], 1);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_span > 0) atomicMax(g.maxspan, s_span);
    if (use_lds)
        for (int i = threadIdx.x; i < nb; i += blockDim.x) { const int v = lh[i]; if (v) atomicAdd(&hist[i], v); }
}

static __device__ __forceinline__ void ba_group_scan_body(const BaGroup& g)
{
    // exclusive scan over (bucket-major, replica-minor) of hist[rep][bucket] into cursor[rep][bucket].  Each wave owns a
    // contiguous range of buckets and walks it 64 buckets per trip, lane = bucket, so every load and store is a row of 64
    // consecutive ints per replica: pass 1 the wave's total, one barrier for the prefix over the waves, pass 2 (the
    // histogram again, L2 hits) a wave scan per trip with a running carry.  (The scan order is the transpose of the storage
    // order: walked entry by entry it was uncoalesced 4-byte traffic from one compute unit — 89 us for the 77 k entries
    // of a 98-camera window.)
    __shared__ int wtot[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (int)blockDim.x >> 6;
    const int nb = g.n_buckets + 1;
    const int seg = ((nb + nw - 1) / nw + 63) & ~63;              // buckets per wave, whole trips
    const int b0 = wave * seg, b1 = min(b0 + seg, nb);
    int tsum = 0;
    for (int bk = b0 + lane; bk < b1; bk += 64) {
#pragma unroll
        for (int rep = 0; rep < GRP_REP; rep++) tsum += g.hist[(size_t)rep * nb + bk];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tsum += __shfl_xor(tsum, off, 64);
    if (lane == 0) wtot[wave] = tsum;
    __syncthreads();
    int carry = 0;
    for (int w = 0; w < wave; w++) carry += wtot[w];
    for (int t0 = b0; t0 < b1; t0 += 64) {
        const int bk = t0 + lane;
        int v[GRP_REP], tot = 0;
#pragma unroll
        for (int rep = 0; rep < GRP_REP; rep++) { v[rep] = bk < b1 ? g.hist[(size_t)rep * nb + bk] : 0; tot += v[rep]; }
        int x = tot;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(x, off, 64);
            if (lane >= off) x += t;
        }
        int run = carry + x - tot;                                 // first position of bucket bk, replica 0
        if (bk < b1) {
#pragma unroll
            for (int rep = 0; rep < GRP_REP; rep++) { g.cursor[(size_t)rep * nb + bk] = run; run += v[rep]; }
        }
        carry += __shfl(x, 63, 64);
    }
}


// the camera mask of landmark p, OR-ed into the mask of the item its sorted position falls into (g.item_mask zeroed before):
// ~it_l atomic instructions per item word, all items side by side — instead of a launch of its own (ba_group_items)
static __device__ __forceinline__ void ba_group_or_item(const BaGroup& g, int p, int pos)
{
    const unsigned long long m0 = g.mask[2 * (size_t)p], m1 = g.mask[2 * (size_t)p + 1];
    unsigned long long* im = (unsigned long long*)g.item_mask + 2 * (size_t)(pos / g.it_l);
    if (m0) atomicOr(im, m0);
    if (m1) atomicOr(im + 1, m1);
}

template <bool OR_ITEMS>
static __device__ __forceinline__ void ba_group_scatter_body(const BaDims& d, const BaBufs& b, const BaGroup& g)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= d.P) return;
    const int nb = g.n_buckets + 1;
    const int pos = atomicAdd(&g.cursor[(size_t)(blockIdx.x & (GRP_REP - 1)) * nb + g.bucket[p]], 1);
    g.sorted[pos] = p;
    const int o0 = b.obs_ptr[p];
    g.lm[pos] = make_int4(p, o0, b.obs_ptr[p + 1] - o0, 0);
    if (OR_ITEMS) ba_group_or_item(g, p, pos);
}

// one wave per item: lanes = the item's 64 landmarks, 128-bit OR across the wave
static __device__ __forceinline__ void ba_group_items_body(const BaDims& d, const BaGroup& g)
{
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= g.n_items) return;
    const int q = t * g.it_l + (threadIdx.x & 63);
    uint64_t m0 = 0, m1 = 0;
    if ((int)(threadIdx.x & 63) < g.it_l && q < d.P) {
        const int p = g.sorted[q];
        m0 = g.mask[2 * (size_t)p];
        m1 = g.mask[2 * (size_t)p + 1];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        m0 |= (uint64_t)__shfl_xor((unsigned long long)m0, off, 64);
        m1 |= (uint64_t)__shfl_xor((unsigned long long)m1, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        g.item_mask[2 * (size_t)t] = m0;
        g.item_mask[2 * (size_t)t + 1] = m1;
    }
}

__global__ __launch_bounds__(64 * SCH_WAVES) void ba_schur_mfma(BaDims d, BaBufs b, BaOpt opt, BaGroup g, int it)
{
    __shared__ BaState st_sh;
    ba_schur_body<true, false>(d, b, opt, g, it, (int)blockIdx.x, &st_sh);
}

// windows of more than SCH_MAXC_LDS cameras (cfg 5: 100 key frames): camera blocks from global memory
__global__ __launch_bounds__(64 * SCH_WAVES) void ba_schur_mfma_big(BaDims d, BaBufs b, BaOpt opt, BaGroup g, int it)
{
    __shared__ BaState st_sh;
    ba_schur_body<false, false>(d, b, opt, g, it, (int)blockIdx.x, &st_sh);
}


// single-window and batched (blockIdx.z = window, arguments from the device array) entry points of the grouping kernels
__global__ __launch_bounds__(1024) void ba_group_count(BaDims d, BaBufs b, BaGroup g) { ba_group_count_body(d, b, g); }
__global__ __launch_bounds__(1024) void ba_group_scan(BaGroup g) { ba_group_scan_body(g); }
__global__ __launch_bounds__(256) void ba_group_scatter(BaDims d, BaBufs b, BaGroup g) { ba_group_scatter_body<false>(d, b, g); }
__global__ __launch_bounds__(256) void ba_group_scatter_items(BaDims d, BaBufs b, BaGroup g) { ba_group_scatter_body<true>(d, b, g); }
// K0 + the grouping's count in ONE launch (single solves): the init work is spread over the count's workgroups; the
// histogram they add into was left at zero by the previous solve's finalize kernel (or by a memset: ba.hip), the item masks
// are zeroed here for the scatter launch that follows
__global__ __launch_bounds__(1024) void ba_init_count(BaDims d, BaBufs b, BaOpt opt, BaGroup g, const double* __restrict__ cams_in,
                                                      const double* __restrict__ pts_in, unsigned long long free_mask, int from_mask,
                                                      uint8_t* __restrict__ cam_free)
{
    ba_init_body(d, b, opt, cams_in, pts_in, free_mask, from_mask, cam_free, (int32_t*)g.item_mask, 4 * g.n_items);
    if ((int)(blockIdx.x * GRP_COUNT_LM) < d.P) ba_group_count_body(d, b, g, free_mask, from_mask);
}

// Scatter with the scan inside (local windows: at most GRP_SCAN_LDS histogram entries): every workgroup scans the
// (bucket-major, replica-minor) histogram for itself in LDS — 2.6 k entries for 18 free cameras, one chunk per thread and
// one workgroup scan — and takes positions as base + atomicAdd on a cursor array that ba_init left at zero.  Replaces the
// one-workgroup scan launch between count and scatter (a launch gap + 4.8 us for 10 KB of work).
#define GRP_SCAN_LDS 4096
template <bool OR_ITEMS>
static __device__ __forceinline__ void ba_group_scatter_scan_body(const BaDims& d, const BaBufs& b, const BaGroup& g)
{
    __shared__ int base[GRP_SCAN_LDS];
    const int nb = g.n_buckets + 1, total = nb * GRP_REP;
    const int per = (total + 255) / 256;                  // <= 16
    const int i0 = (int)threadIdx.x * per;
    int v[GRP_SCAN_LDS / 256];
    int sum = 0;
#pragma unroll
    for (int u = 0; u < GRP_SCAN_LDS / 256; u++) {
        const int i = i0 + u;
        v[u] = (u < per && i < total) ? g.hist[(size_t)(i % GRP_REP) * nb + i / GRP_REP] : 0;
        sum += v[u];
    }
    int tot;
    int run = rs_block_exclusive_scan(sum, &tot);
#pragma unroll
    for (int u = 0; u < GRP_SCAN_LDS / 256; u++) {
        const int i = i0 + u;
        if (u < per && i < total) base[i] = run;
        run += v[u];
    }
    __syncthreads();
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= d.P) return;
    const int rep = (int)(blockIdx.x & (GRP_REP - 1)), bk = g.bucket[p];
    const int pos = base[bk * GRP_REP + rep] + atomicAdd(&g.cursor[(size_t)rep * nb + bk], 1);
    g.sorted[pos] = p;
    const int o0 = b.obs_ptr[p];
    g.lm[pos] = make_int4(p, o0, b.obs_ptr[p + 1] - o0, 0);
    if (OR_ITEMS) ba_group_or_item(g, p, pos);
}
__global__ __launch_bounds__(256) void ba_group_scatter_scan(BaDims d, BaBufs b, BaGroup g) { ba_group_scatter_scan_body<false>(d, b, g); }
__global__ __launch_bounds__(256) void ba_group_scatter_scan_items(BaDims d, BaBufs b, BaGroup g) { ba_group_scatter_scan_body<true>(d, b, g); }
__global__ __launch_bounds__(256) void ba_group_items(BaDims d, BaGroup g) { ba_group_items_body(d, g); }
// The round's decision as a launch of its own (one wave per window), for K5 launches whose workgroups do not all run at
// once: there every item re-deriving it (slot lines, twelve wave reductions, the trust-region logic: 2.7 us) is serial work
// per compute unit — 5 item rounds at cfg 5, 20 in a batch of 32 windows.
__global__ __launch_bounds__(64) void ba_decide_round(BaBufs b, BaOpt opt, int it)
{
    __shared__ BaState sh;
    b.decided = 0;
    (void)ba_round_state(b, opt, it, &sh, true);
}
__global__ __launch_bounds__(64) void ba_decide_round_batch(const BaWin* w, BaOpt opt, int it)
{
    __shared__ BaState sh;
    BaBufs b = ba_win_round(w[blockIdx.x], it, false);
    b.decided = 0;
    (void)ba_round_state(b, opt, it, &sh, true);
}
__global__ __launch_bounds__(1024) void ba_group_count_batch(const BaWin* w) { const BaWin& x = w[blockIdx.z]; if ((int)(blockIdx.x * GRP_COUNT_LM) < x.d.P) ba_group_count_body(x.d, x.b, x.g); }
__global__ __launch_bounds__(1024) void ba_group_scan_batch(const BaWin* w) { ba_group_scan_body(w[blockIdx.z].g); }
__global__ __launch_bounds__(256) void ba_group_scatter_batch(const BaWin* w) { const BaWin& x = w[blockIdx.z]; ba_group_scatter_body<false>(x.d, x.b, x.g); }
__global__ __launch_bounds__(256) void ba_group_items_batch(const BaWin* w) { const BaWin& x = w[blockIdx.z]; ba_group_items_body(x.d, x.g); }
__global__ __launch_bounds__(64 * SCH_WAVES) void ba_schur_mfma_batch(const BaWin* w, BaOpt opt, int it)
{
    const BaWin& x = w[blockIdx.z];
    if ((int)blockIdx.x >= x.g.n_items) return;
    BaBufs b = ba_win_round(x, it, false);
    b.decided = 1;                       // ba_decide_round_batch ran in front of this launch
    __shared__ BaState st_sh;
    ba_schur_body<true, false>(x.d, b, opt, x.g, it, (int)blockIdx.x, &st_sh);
}

// ------------------------------------------------------------------ host glue
size_t ba_group_bytes(int P, int Cf, int M)
{
    const size_t nb = ((size_t)Cf * Cf + 2) * GRP_REP;
    const size_t ni = ((size_t)P + 31) / 32 + 1;                 // (the smallest item: 32 landmarks, throughput mode)
    return 256 * 10 + sizeof(int32_t) * (2 * (size_t)P + 2 * nb + 16 + (size_t)M) + sizeof(int4) * (size_t)P +
           sizeof(uint64_t) * (2 * (size_t)P + 2 * ni);
}

// The whole grouping in ONE workgroup (a launch costs ~4 us on this GPU; a local window has ~10^4
// landmarks): histogram and cursors in LDS, landmarks strided over the 1024 threads.
#define GRP_SMALL_P 2048
__global__ __launch_bounds__(1024) void ba_group_small(BaDims d, BaBufs b, BaGroup g)
{
    __shared__ int lh[GRP_LDS_BINS];
    __shared__ int wsum[16];
    __shared__ int carry;
    const int nb = g.n_buckets + 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < nb; i += 1024) lh[i] = 0;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int p = threadIdx.x; p < d.P; p += 1024) {
        uint64_t m0 = 0, m1 = 0;
        int first = 1 << 30, last = -1;
        for (int o = b.obs_ptr[p]; o < b.obs_ptr[p + 1]; o++) {
            const int c = b.obs_cam[o];
            const int s = b.slot[c];
            g.obs_cs[o] = c | ((s + 1) << 16);
            if (s < 0) continue;
            if (s < 64) m0 |= 1ull << s; else m1 |= 1ull << (s - 64);
            first = min(first, s);
            last = max(last, s);
        }
        g.mask[2 * (size_t)p] = m0;
        g.mask[2 * (size_t)p + 1] = m1;
        const int bk = last < 0 ? d.Cf * d.Cf : first * d.Cf + last;
        g.bucket[p] = bk;
        if (last > first) atomicMax(g.maxspan, last - first);
        atomicAdd(&lh[bk], 1);
    }
    __syncthreads();
    // exclusive scan of lh in place
    for (int base = 0; base < nb; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < nb ? lh[i] : 0;
        int x = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(x, off, 64);
            if (lane >= off) x += t;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int pre = carry;
        for (int w = 0; w < wave; w++) pre += wsum[w];
        if (i < nb) lh[i] = pre + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = pre + x;
        __syncthreads();
    }
    for (int p = threadIdx.x; p < d.P; p += 1024) {
        const int pos = atomicAdd(&lh[g.bucket[p]], 1);
        g.sorted[pos] = p;
        const int o0 = b.obs_ptr[p];
        g.lm[pos] = make_int4(p, o0, b.obs_ptr[p + 1] - o0, 0);
    }
    __syncthreads();
    for (int t = wave; t < g.n_items; t += 16) {
        const int q = t * g.it_l + lane;
        uint64_t m0 = 0, m1 = 0;
        if (lane < g.it_l && q < d.P) {
            const int p = g.sorted[q];
            m0 = g.mask[2 * (size_t)p];
            m1 = g.mask[2 * (size_t)p + 1];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            m0 |= (uint64_t)__shfl_xor((unsigned long long)m0, off, 64);
            m1 |= (uint64_t)__shfl_xor((unsigned long long)m1, off, 64);
        }
        if (lane == 0) {
            g.item_mask[2 * (size_t)t] = m0;
            g.item_mask[2 * (size_t)t + 1] = m1;
        }
    }
}

void ba_group_zero_range(const BaGroup& g, int32_t** ptr, int* count)
{
    *ptr = g.hist;          // the histogram and, behind it, the cursors (ba_group_scatter_scan counts from zero)
    *count = (int)(g.maxspan - g.hist) + 1;      // histogram, cursors and the span word behind them
}

static size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
// landmarks per item of a single solve: 40 while that keeps the items within ONE shift of 256 workgroups, 64 beyond.  (48 / 56
// are there as "ba_item_landmarks": 3 us per launch faster at 11 - 14 k landmarks when the items' camera unions are small, 10 us
// slower when they are not — the 8 x 8-tile class deals its 36 tiles to the workgroup's waves, and 6 waves carry 6 each.)
static int ba_default_item(int P) { return P <= 256 * IT_L_SMALL ? IT_L_SMALL : IT_L; }

void ba_group_carve(char* base, int P, int Cf, int M, BaGroup* g)
{
    const size_t nb = ((size_t)Cf * Cf + 2) * GRP_REP;
    const size_t ni_max = ((size_t)P + 31) / 32 + 1;
    g->it_l = ba_default_item(P);
    const size_t ni = P > 0 ? ((size_t)P + g->it_l - 1) / g->it_l : 1;     // an empty landmark shard keeps one (empty) item:
    size_t off = 0;                                                            // its workgroup runs the round's decision
    g->sorted = (int32_t*)(base + off); off += al256(sizeof(int32_t) * P);
    g->bucket = (int32_t*)(base + off); off += al256(sizeof(int32_t) * P);
    g->hist = (int32_t*)(base + off); off += al256(sizeof(int32_t) * nb);
    g->cursor = (int32_t*)(base + off); off += al256(sizeof(int32_t) * (nb + 16));
    g->maxspan = g->cursor + nb;
    g->mask = (uint64_t*)(base + off); off += al256(sizeof(uint64_t) * 2 * P);
    g->item_mask = (uint64_t*)(base + off); off += al256(sizeof(uint64_t) * 2 * ni_max);
    g->lm = (int4*)(base + off); off += al256(sizeof(int4) * P);
    g->obs_cs = (int32_t*)(base + off); off += al256(sizeof(int32_t) * M);
    g->n_items = (int)ni;
    g->n_buckets = Cf * Cf;           // + 1 bucket for landmarks without a free camera
}

int ba_launch_grouping(rs_context* ctx, const BaDims& d, const BaBufs& b, const BaGroup& g)
{
    hipStream_t s = ctx->stream;
    rs_prof_scope ps(ctx, "K5s_group_landmarks");
    if (d.P <= GRP_SMALL_P && g.n_buckets + 1 <= GRP_LDS_BINS) {
        hipLaunchKernelGGL(ba_group_small, dim3(1), dim3(1024), 0, s, d, b, g);
        return RS_OK;
    }
    // g.hist was zeroed by ba_init (ba_group_zero_range)
    const int pb = (d.P + 255) / 256;
    hipLaunchKernelGGL(ba_group_count, dim3((d.P + GRP_COUNT_LM - 1) / GRP_COUNT_LM), dim3(1024), 0, s, d, b, g);
    if ((g.n_buckets + 1) * GRP_REP <= GRP_SCAN_LDS) {
        hipLaunchKernelGGL(ba_group_scatter_scan, dim3(pb), dim3(256), 0, s, d, b, g);
    } else {
        hipLaunchKernelGGL(ba_group_scan, dim3(1), dim3(1024), 0, s, g);
        hipLaunchKernelGGL(ba_group_scatter, dim3(pb), dim3(256), 0, s, d, b, g);
    }
    hipLaunchKernelGGL(ba_group_items, dim3((g.n_items + 3) / 4), dim3(256), 0, s, d, g);
    return RS_OK;
}

// K0 + grouping as two launches (ba_init_count; scatter with the item masks) instead of four: windows the one-workgroup
// grouping does not cover.  The caller guarantees that the histogram / cursor / span words (ba_group_zero_range) are zero.
bool ba_setup_fusable(const BaDims& d, const BaGroup& g)
{
    return !(d.P <= GRP_SMALL_P && g.n_buckets + 1 <= GRP_LDS_BINS) && d.P > 0;
}
void ba_launch_setup_fused(rs_context* ctx, const BaDims& d, const BaBufs& b, const BaOpt& opt, const BaGroup& g, const double* cams_in,
                           const double* pts_in, unsigned long long free_mask, int from_mask, uint8_t* cam_free)
{
    hipStream_t s = ctx->stream;
    {
        rs_prof_scope ps(ctx, "K0_ba_init_count");
        const int cb = (d.P + GRP_COUNT_LM - 1) / GRP_COUNT_LM;
        hipLaunchKernelGGL(ba_init_count, dim3(cb < 16 ? 16 : cb), dim3(1024), 0, s, d, b, opt, g, cams_in, pts_in, free_mask, from_mask, cam_free);
    }
    rs_prof_scope ps(ctx, "K5s_group_landmarks");
    const int pb = (d.P + 255) / 256;
    if ((g.n_buckets + 1) * GRP_REP <= GRP_SCAN_LDS) {
        hipLaunchKernelGGL(ba_group_scatter_scan_items, dim3(pb), dim3(256), 0, s, d, b, g);
    } else {
        hipLaunchKernelGGL(ba_group_scan, dim3(1), dim3(1024), 0, s, g);
        hipLaunchKernelGGL(ba_group_scatter_items, dim3(pb), dim3(256), 0, s, d, b, g);
    }
}

size_t ba_schur_lds_bytes(int C, int Cf, int it_l, int ns)
{
    (void)Cf;
    const size_t prep = C <= SCH_MAXC_LDS ? (size_t)C * BA_PREP_LDS : 0;
    // behind the slot table: Mt [it_l][6] (set-by-set path) or Gt [ns - 1][it_l][6] (all sets in one pass) — sized by the
    // item and the solve, not by their maxima: throughput mode runs TWO workgroups of 32-landmark items per compute unit
    // and has 160 KB for both
    const size_t gt = 6 * (size_t)(ns > 2 ? ns - 1 : 1) * (size_t)it_l;
    return sizeof(double) * ((size_t)sch_tile_doubles(it_l) + (size_t)SCH_UCAP * 42 + prep + gt) + sizeof(int) * 32;
}

int ba_prepare_schur(int C, int Cf)
{
    const void* fn = C <= SCH_MAXC_LDS ? (const void*)ba_schur_mfma : (const void*)ba_schur_mfma_big;
    return (int)rs_lds_attr(fn, ba_schur_lds_bytes(C, Cf));
}

void ba_launch_schur(hipStream_t s, const BaDims& d, const BaBufs& b, const BaOpt& opt, const BaGroup& g, int it)
{
    if (d.C <= SCH_MAXC_LDS)
        hipLaunchKernelGGL(ba_schur_mfma, dim3(g.n_items), dim3(8 * g.it_l), ba_schur_lds_bytes(d.C, d.Cf, g.it_l, b.ns), s, d, b, opt, g, it);
    else
        hipLaunchKernelGGL(ba_schur_mfma_big, dim3(g.n_items), dim3(8 * g.it_l), ba_schur_lds_bytes(d.C, d.Cf, g.it_l, b.ns), s, d, b, opt, g, it);
}

void ba_launch_decide(hipStream_t s, const BaBufs& b, const BaOpt& opt, int it)
{
    hipLaunchKernelGGL(ba_decide_round, dim3(1), dim3(64), 0, s, b, opt, it);
}
void ba_launch_decide_batch(hipStream_t s, const BaWin* d_wins, int B, const BaOpt& opt, int it)
{
    hipLaunchKernelGGL(ba_decide_round_batch, dim3(B), dim3(64), 0, s, d_wins, opt, it);
}

// ---- batched launches (one per kernel for B windows)
void ba_launch_grouping_batch(hipStream_t s, const BaWin* d_wins, int B, int max_P, int max_items)
{
    const int pb = (max_P + 255) / 256;
    hipLaunchKernelGGL(ba_group_count_batch, dim3((max_P + GRP_COUNT_LM - 1) / GRP_COUNT_LM, 1, B), dim3(1024), 0, s, d_wins);
    hipLaunchKernelGGL(ba_group_scan_batch, dim3(1, 1, B), dim3(1024), 0, s, d_wins);
    hipLaunchKernelGGL(ba_group_scatter_batch, dim3(pb, 1, B), dim3(256), 0, s, d_wins);
    hipLaunchKernelGGL(ba_group_items_batch, dim3((max_items + 3) / 4, 1, B), dim3(256), 0, s, d_wins);
}

int ba_prepare_schur_batch(size_t lds)
{
    return (int)rs_lds_attr((const void*)ba_schur_mfma_batch, lds);
}

void ba_launch_schur_batch(hipStream_t s, const BaWin* d_wins, int B, const BaOpt& opt, int it, int max_items, int it_l, size_t lds)
{
    hipLaunchKernelGGL(ba_schur_mfma_batch, dim3(max_items, 1, B), dim3(8 * it_l), lds, s, d_wins, opt, it);
}

void ba_group_set_items(BaGroup* g, int P, bool throughput, int batch_item)
{
    // throughput mode (windows batched in one grid): 32 landmarks per item by default — a 75 KB LDS image, so that TWO
    // workgroups share a compute unit and one's barriers / LDS round trips are covered by the other's arithmetic
    const int tp = (batch_item >= 32 && batch_item <= 64 && batch_item % 8 == 0) ? batch_item : 32;
    g->it_l = throughput ? tp : ba_default_item(P);
    g->n_items = P > 0 ? (P + g->it_l - 1) / g->it_l : 1;
}
