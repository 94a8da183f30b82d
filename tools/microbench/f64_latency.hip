// f64_latency.hip — cycles per instruction of one wave on gfx950: dependent / independent v_fma_f64, v_rcp_f64 chains,
// v_readlane broadcast of a double, LDS write -> read round trip, a 6x6 L D L^T per lane.  One workgroup of one wave
// (and, second column, of two waves on one SIMD: waves 0 and 4 of a 512-thread workgroup, the others parked).
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/f64_latency.hip -o /tmp/f64_latency && /tmp/f64_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define N 256
__device__ __forceinline__ double rl(double v, int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

template <int MODE>
__global__ __launch_bounds__(512) void k(double* out, unsigned long long* cyc, double seed)
{
    __shared__ double sh[1024];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if ((wave & 3) != 0) return;                    // waves 0 and 4 share a SIMD
    double a = seed + lane * 1e-3, b = 1.0000001, c = 1e-9;
    double r[8];
    for (int i = 0; i < 8; i++) r[i] = a + i;
    sh[threadIdx.x] = a;
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t0 = clock64();
    if (MODE == 0) {            // dependent fma chain
#pragma unroll 16
        for (int i = 0; i < N; i++) a = fma(a, b, c);
    } else if (MODE == 1) {     // 8 independent fma chains
#pragma unroll 4
        for (int i = 0; i < N / 8; i++) {
#pragma unroll
            for (int q = 0; q < 8; q++) r[q] = fma(r[q], b, c);
        }
        for (int q = 1; q < 8; q++) a += r[q];
        a += r[0];
    } else if (MODE == 2) {     // dependent rcp chain (raw v_rcp_f64)
#pragma unroll 16
        for (int i = 0; i < N; i++) a = __builtin_amdgcn_rcp(a) + 0.0 * c;
    } else if (MODE == 3) {     // readlane broadcast of a double, dependent
#pragma unroll 16
        for (int i = 0; i < N; i++) a = rl(a, i & 63) + c;
    } else if (MODE == 4) {     // LDS write -> read round trip, dependent
#pragma unroll 8
        for (int i = 0; i < N; i++) {
            sh[threadIdx.x] = a;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            a = sh[threadIdx.x ^ 1] + c;
        }
    } else if (MODE == 5) {     // 6x6 L D L^T per lane, dependent through the first entry (N / 8 of them)
        for (int i = 0; i < N / 8; i++) {
            double L[6][6];
#pragma unroll
            for (int x = 0; x < 6; x++)
#pragma unroll
                for (int y = 0; y <= x; y++) L[x][y] = (x == y ? 10.0 + a : 0.1 * (x + y)) ;
            double acc = 0.0;
#pragma unroll
            for (int cc = 0; cc < 6; cc++) {
                const double piv = L[cc][cc];
                double rd = __builtin_amdgcn_rcp(piv);
                rd = fma(rd, fma(-piv, rd, 1.0), rd);
                double lc[6];
#pragma unroll
                for (int x = cc + 1; x < 6; x++) lc[x] = L[x][cc] * rd;
#pragma unroll
                for (int x = cc + 1; x < 6; x++)
#pragma unroll
                    for (int y = cc + 1; y <= x; y++) L[x][y] -= lc[x] * L[y][cc];
                acc += rd;
            }
            a = acc * 1e-3;
        }
    } else if (MODE == 6) {     // dependent f32 fma chain, for scale
        float fa = (float)a, fb = 1.0000001f, fc = 1e-9f;
#pragma unroll 16
        for (int i = 0; i < N; i++) fa = fmaf(fa, fb, fc);
        a = fa;
    } else if (MODE == 7) {     // ds_read latency: dependent pointer chase in LDS
        int idx = lane;
        ((int*)sh)[threadIdx.x] = (threadIdx.x + 1) & 63;
        __builtin_amdgcn_s_waitcnt(0);
#pragma unroll 8
        for (int i = 0; i < N; i++) idx = ((int*)sh)[idx];
        a = idx;
    }
    const unsigned long long t1 = clock64();
    out[threadIdx.x] = a;
    if (lane == 0) cyc[wave >> 2] = t1 - t0;
}

template <int MODE>
void run(const char* name, int per)
{
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 512 * 8); hipMalloc(&cyc, 16);
    for (int threads : {64, 512}) {
        unsigned long long h[2] = {0, 0};
        for (int rep = 0; rep < 3; rep++) {
            hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, out, cyc, 1.5);
            hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
        }
        printf("%-34s %s: %7.1f cycles per %s\n", name, threads == 64 ? "one wave      " : "two waves/SIMD", (double)h[0] / per, MODE == 5 ? "6x6 factor" : "op");
    }
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<0>("dependent v_fma_f64", N);
    run<1>("8 independent v_fma_f64 chains", N);
    run<2>("dependent v_rcp_f64 (+ add)", N);
    run<3>("dependent readlane f64 (+ add)", N);
    run<4>("LDS write -> read (+ add)", N);
    run<5>("6x6 L D L^T per lane", N / 8);
    run<6>("dependent v_fma_f32", N);
    run<7>("dependent ds_read_b32", N);
    return 0;
}
