#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) double d4;
__global__ void k_dep(double* out, long long* cyc, int n) {
    d4 acc = {0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    long long t0 = clock64();
    for (int i = 0; i < n; i++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    out[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    long long t1 = clock64();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_indep(double* out, long long* cyc, int n) {
    d4 a0 = {0,0,0,0}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    long long t0 = clock64();
    for (int i = 0; i < n; i += 8) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a3, 0, 0, 0);
        a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a4, 0, 0, 0);
        a5 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a5, 0, 0, 0);
        a6 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a6, 0, 0, 0);
        a7 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a7, 0, 0, 0);
    }
    d4 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    long long t1 = clock64();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_fma(double* out, long long* cyc, int n) {   // dependent fp64 FMA chain and 8 independent chains
    double x = threadIdx.x * 1e-3, y0 = 1, y1 = 2, y2 = 3, y3 = 4, y4 = 5, y5 = 6, y6 = 7, y7 = 8;
    long long t0 = clock64();
    for (int i = 0; i < n; i++) y0 = fma(y0, x, 1.0);
    long long t1 = clock64();
    for (int i = 0; i < n; i += 8) { y0 = fma(y0, x, 1.0); y1 = fma(y1, x, 1.0); y2 = fma(y2, x, 1.0); y3 = fma(y3, x, 1.0); y4 = fma(y4, x, 1.0); y5 = fma(y5, x, 1.0); y6 = fma(y6, x, 1.0); y7 = fma(y7, x, 1.0); }
    long long t2 = clock64();
    double r = y0;
    for (int i = 0; i < n; i++) r = __builtin_amdgcn_rcp(r + 1.0);
    long long t3 = clock64();
    for (int i = 0; i < n; i++) r = __builtin_amdgcn_rsq(r + 1.0);
    long long t4 = clock64();
    out[threadIdx.x] = y0 + y1 + y2 + y3 + y4 + y5 + y6 + y7 + r;
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; }
}
__global__ void k_bar(long long* cyc, int n) {
    long long t0 = clock64();
    for (int i = 0; i < n; i++) __syncthreads();
    long long t1 = clock64();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    double* out; long long* cyc; hipMalloc(&out, 8 * 1024); hipMalloc(&cyc, 64);
    long long h[8]; const int n = 4096;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_dep, dim3(1), dim3(64), 0, 0, out, cyc, n); hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
        printf("mfma_f64_16x16x4 dependent   : %.1f cycles each\n", (double)h[0] / n);
        hipLaunchKernelGGL(k_indep, dim3(1), dim3(64), 0, 0, out, cyc, n); hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
        printf("mfma_f64_16x16x4 independent : %.1f cycles each (1 wave)\n", (double)h[0] / n);
        hipLaunchKernelGGL(k_indep, dim3(1), dim3(256), 0, 0, out, cyc, n); hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
        printf("mfma_f64_16x16x4 independent : %.1f cycles each (4 waves on one CU, wave 0)\n", (double)h[0] / n);
        hipLaunchKernelGGL(k_fma, dim3(1), dim3(64), 0, 0, out, cyc, n); hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
        printf("v_fma_f64 dependent %.1f, 8 independent chains %.1f, rcp chain %.1f, rsq chain %.1f cycles/op\n", (double)h[0] / n, (double)h[1] / n, (double)h[2] / n, (double)h[3] / n);
        hipLaunchKernelGGL(k_bar, dim3(1), dim3(256), 0, 0, cyc, n); hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
        printf("__syncthreads (4 waves)      : %.1f cycles each\n", (double)h[0] / n);
        hipLaunchKernelGGL(k_bar, dim3(1), dim3(1024), 0, 0, cyc, n); hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
        printf("__syncthreads (16 waves)     : %.1f cycles each\n", (double)h[0] / n);
    }
    return 0;
}
