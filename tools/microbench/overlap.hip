#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) double d4;
// mode bit0: wave `fw` runs a dependent f64 FMA chain; bit1: wave `mw` runs independent f64 MFMAs;
// same wave when fw == mw (interleaved 1 MFMA : 8 FMA)
__global__ void k_mix(double* out, long long* cyc, unsigned* hw, int n, int fw, int mw, int mode) {
    const int wave = threadIdx.x >> 6;
    unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) hw[wave] = id;
    d4 a0 = {0,0,0,0}, a1 = a0, a2 = a0, a3 = a0;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4, y = 1.0;
    __syncthreads();
    long long t0 = clock64();
    if (fw == mw && wave == fw && mode == 3) {
#pragma unroll 4
        for (int i = 0; i < n; i += 4) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a0, 0, 0, 0);
            y = fma(y, b, 1.0); y = fma(y, b, 1.0); y = fma(y, b, 1.0); y = fma(y, b, 1.0); y = fma(y, b, 1.0); y = fma(y, b, 1.0); y = fma(y, b, 1.0); y = fma(y, b, 1.0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a1, 0, 0, 0);
            y = fma(y, b, 1.0); y = fma(y, b, 1.0); y = fma(y, b, 1.0); y = fma(y, b, 1.0); y = fma(y, b, 1.0); y = fma(y, b, 1.0); y = fma(y, b, 1.0); y = fma(y, b, 1.0);
            a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a2, 0, 0, 0);
            y = fma(y, b, 1.0); y = fma(y, b, 1.0); y = fma(y, b, 1.0); y = fma(y, b, 1.0); y = fma(y, b, 1.0); y = fma(y, b, 1.0); y = fma(y, b, 1.0); y = fma(y, b, 1.0);
            a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a3, 0, 0, 0);
            y = fma(y, b, 1.0); y = fma(y, b, 1.0); y = fma(y, b, 1.0); y = fma(y, b, 1.0); y = fma(y, b, 1.0); y = fma(y, b, 1.0); y = fma(y, b, 1.0); y = fma(y, b, 1.0);
        }
    } else {
        if (wave == fw && (mode & 1)) {
#pragma unroll 32
            for (int i = 0; i < 8 * n; i++) y = fma(y, b, 1.0);
        }
        if (wave == mw && (mode & 2)) {
#pragma unroll 4
            for (int i = 0; i < n; i += 4) {
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a3, 0, 0, 0);
            }
        }
    }
    d4 s = a0 + a1 + a2 + a3;
    out[threadIdx.x] = s[0] + s[1] + s[2] + s[3] + y;
    long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}
int main() {
    double* out; long long* cyc; unsigned* hw; hipMalloc(&out, 8 * 2048); hipMalloc(&cyc, 256); hipMalloc(&hw, 256);
    long long h[32]; unsigned hh[32]; const int n = 2048;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct { int threads, fw, mw, mode; const char* what; } cases[] = {
        {64, 0, 0, 1, "FMA chain alone (8n dependent fma)"},
        {64, 0, 0, 2, "MFMA alone (n independent)"},
        {64, 0, 0, 3, "same wave, 1 MFMA : 8 dependent FMA interleaved"},
        {512, 0, 4, 1, "8 waves, FMA chain on wave 0 only"},
        {512, 0, 4, 3, "8 waves, FMA wave 0 + MFMA wave 4 (same SIMD?)"},
        {512, 0, 1, 3, "8 waves, FMA wave 0 + MFMA wave 1 (other SIMD?)"},
    };
    for (int rep = 0; rep < 2; rep++)
    for (auto& c : cases) {
        hipMemset(cyc, 0, 256);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_mix, dim3(1), dim3(c.threads), 0, 0, out, cyc, hw, n, c.fw, c.mw, c.mode);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, cyc, 256, hipMemcpyDeviceToHost); hipMemcpy(hh, hw, 256, hipMemcpyDeviceToHost);
        printf("%-52s: wave%d %.1f cyc per 8 fma | wave%d %.1f cyc per mfma | kernel %.1f us | simd ids:", c.what, c.fw, (double)h[c.fw] / n, c.mw, (double)h[c.mw] / n, ms * 1e3);
        for (int w = 0; w < c.threads / 64; w++) printf(" %u", (hh[w] >> 4) & 3);
        printf("\n");
    }
    return 0;
}
